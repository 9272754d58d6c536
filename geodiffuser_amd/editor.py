"""Edit orchestration — ``perform_geometric_edit`` (alias ``run_geodiffuser``) and the per-step loop.

Mirror of GeoDiffuser/utils/editor.py: ``text2image_ldm_stable`` :65-423 (the 50-step loop: optimisation pass with
gradient, CFG pass, reference-latent replacement, latent warp), ``run_and_display`` :713-718 and
``perform_geometric_edit`` :428-710 (same signature and defaults).  Dead code of the reference is not reproduced:
the SGD optimizer branch (``use_optimizer`` is never forwarded), the per-step decay of the splat constants
(it mutates an object nobody reads, F3), the ``geometry_stitch*`` editors (their controller classes do not exist).
"""
from __future__ import annotations

import contextlib
import logging
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops, vis_utils, warp_utils
from .attention_processors import (AttentionGeometryEdit, AttentionGeometryRemover, VanillaAttentionProcessor,
                                   register_attention_control_diffusers, set_attn_processor_for_edit)
from .diffusion import _unet_nograd, diffusion_step, encode_text, image2latent, latent2image, load_model
from .generic_torch import binarize_tensor, norm_tensor, reshape_transform_coords, torch_erode
from .image_processing import masked_histogram_matching
from .inversion import NullInversion
from .graphs import GraphedOptPass, release_opt_graph
from .optimization import (_apply_latent_update, _update_latent, adaptive_optimization_step_editing,
                           adaptive_optimization_step_remover)
from .warp_utils import warp_grid_edit

UNCOND_TEXT = ""
DIFFUSION_MODEL = "stabilityai/stable-diffusion-2-1-base"
LOW_RESOURCE = False
NUM_DDIM_STEPS = 50
GUIDANCE_SCALE = 4.0
MAX_NUM_WORDS = 77
IMAGE_SIZE = 512
SKIP_OPTIM_STEPS = 0
SEED = 1234
DEVICE = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
MODE = "bilinear"

LDM_STABLE = None
SCHEDULER = None
TOKENIZER = None
UNET_NAME = None
PROGRESS_BAR = None
TIE_IDENTICAL_ROWS = os.environ.get("GD_TIE_ROWS", "1") == "1"   # first optimisation pass of an edit: see attention_processors.rows_identical
HONOUR_SPLAT_ARGS = os.environ.get("GD_HONOUR_SPLAT_ARGS", "0") == "1"   # see _perform_geometric_edit: the reference ignores them (F3)
SPLATTER = warp_utils.RasterizePointsXYsBlending()      # editor.py:50 — the reference's dead object, kept for attribute compatibility
SKIP_UNCOND_REF = True      # drop the CFG pass's unused `uncond_ref` batch row (identical edit output)
# at a step with an optimisation pass the CFG pass takes the reference row's layer tensors from that pass instead of running the
# reference sample through the UNet a second time with the same inputs (attention_processors.ref_stash; captured passes only)
REF_FROM_OPT = os.environ.get("GD_REF_FROM_OPT", "1") == "1"
REF_FROM_OPT_PASSES = 0     # CFG passes that ran without their reference row so far (both drivers count here)
# ... and one step further (r06): the reference sample of an optimisation step rides in the CFG pass of the step BEFORE it as a fourth,
# vanilla row with its own timestep (its latent is the inversion trajectory's, known ahead; rows do not interact outside the hooked
# layers), so the optimisation pass runs forward + backward on the edit row alone (attention_processors "ONE STEP AHEAD").  Applies where
# REF_FROM_OPT does, from the second optimisation step of an edit on (the first has no CFG pass in front of it) and where the pass in
# front is the plain 3-row form; anything else falls back to the two-row optimisation pass.
# REF_AHEAD: 1 (default) = ALL reference rows of an edit in ONE batched vanilla pass behind the inversion, row m copied into persistent one-row
# tensors before optimisation step m (attention_processors "ALL REFERENCE ROWS"); 2 = the carrying form described above; 0 = off.
REF_AHEAD = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("GD_REF_AHEAD", "1").strip(), 1)
REF_AHEAD_PASSES = 0        # optimisation passes that ran on the edit row alone so far


PASS_TIMES = {} if os.environ.get("GD_PASS_TIMES", "0") == "1" else None      # kind -> [(start event, end event)]


def _timed_passes(cfg_pass, opt_pass, controller):
    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def cfg(lat, ctx, tt, ahead=None):
        by_stash = controller.ref_stash_serial is not None and controller.ref_stash_t == int(tt)
        kind = "cfg 2 rows (reference row from a stash)" if by_stash else ("cfg 4 rows (carries the next reference row)" if ahead is not None else "cfg 3 rows")
        a = ev()
        out = cfg_pass(lat, ctx, tt, ahead)
        PASS_TIMES.setdefault(kind, []).append((a, ev()))
        return out

    class _Opt:
        def grads(self, c, lat, ctx, t, edit_row_only=False):
            a = ev()
            out = opt_pass.grads(c, lat, ctx, t, edit_row_only=edit_row_only)
            PASS_TIMES.setdefault("optimisation pass, edit row only" if edit_row_only else "optimisation pass, 2 rows", []).append((a, ev()))
            return out

    return cfg, _Opt()


def pass_times_report() -> str:
    torch.cuda.synchronize()
    rows = [f"{k}: {len(v)} passes, {sum(a.elapsed_time(b) for a, b in v) / len(v):.2f} ms each" for k, v in sorted(PASS_TIMES.items())]
    PASS_TIMES.clear()
    return "\n".join(rows)


@torch.no_grad()
def _reference_rows_pass(model, controller, steps, timesteps, latents, ddim_latents, ref_text, transform_coordinates):
    """The reference sample of every optimisation step of an edit as ONE batch of vanilla rows (row m: the latent the driver will find in
    batch row 0 at step steps[m] — the initial latent at step 0, the inversion trajectory's entry afterwards, U/editor.py:375-377 —, its
    timestep, the reference's text row), through the hooked UNet with ``collect_ahead = "all"``.  -> ({step: row}, ref_slots entry) or
    None when the pass left no complete set of tensors (a layer off the token-major path: the drivers then fall back)."""
    if not steps:
        return None
    n = len(ddim_latents)
    rows = [latents[0:1] if i == 0 else ddim_latents[n - 1 - i].type_as(latents) for i in steps]
    if any(r.shape != latents[0:1].shape for r in rows):
        return None
    x = torch.cat([r.to(latents.dtype) for r in rows])
    ctx = ref_text.expand(len(steps), -1, -1).contiguous()           # (cond_ref: context row 2 of every step, U/editor.py:157-160)
    set_attn_processor_for_edit(model, coords_base=(0, 1), coords_edit=(1, 2), use_cfg=True, n_batch=len(steps))
    controller.collect_ahead = "all"
    controller.ref_stash = controller.ref_stash_serial = controller.ref_stash_t = None
    step0, layer0 = controller.cur_step, controller.cur_att_layer
    ev0 = None
    if PASS_TIMES is not None:
        ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
    try:
        _unet_nograd(model, controller, x, tuple(int(timesteps[i]) for i in steps), ctx, "refs", transform_coordinates, ctx_src=None, learn=False)
    finally:
        controller.collect_ahead = False
        controller.cur_step, controller.cur_att_layer = step0, layer0          # not a step of the edit: the counters stay where they were
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
            PASS_TIMES.setdefault(f"reference rows of all optimisation steps, one pass of {len(steps)} rows", []).append((ev0, ev1))
    stash = controller.ref_stash
    controller.ref_stash = controller.ref_stash_serial = None
    if stash is None or any(t.shape[0] != len(steps) for e in stash for t in e):
        return None
    return {i: m for m, i in enumerate(steps)}, controller.ref_slots(stash)


def ref_from_opt_supported() -> bool:
    """Both passes hand the hooked layers queries that carry scale * log2(e) from the projection's epilogue (the same 16-bit values)."""
    from . import attention_processors as AP
    return bool(AP.SCALED_Q and AP.SCALED_Q_OPT and AP.TOK_OPT and AP.TOKEN_MAJOR and AP.FUSED_WARP)


def clear_controller_loss(controller):
    """generic.py:41-47."""
    controller.loss = 0.0
    if controller.loss_log_dict is not None:
        controller.initialize_loss_log_dict()


def _log_keys_vals(loss_log_dict):
    keys, vals = [], []
    for att_type in ("self", "cross"):
        for key, v in loss_log_dict[att_type].items():
            keys.append((att_type, key))
            vals.append(v if torch.is_tensor(v) else torch.tensor(float(v)))
    return keys, vals


def _log_from_host(loss_log_dict, keys, host):
    out = {"self": {}, "cross": {}}
    for (a, k), x in zip(keys, host):
        out[a][k] = x
    for k, v in loss_log_dict.items():
        if k not in ("self", "cross"):
            out[k] = v
    return out


def convert_loss_log_to_numpy(loss_log_dict):
    """generic.py:50-60 — one host sync for the whole dict instead of one ``.item()`` per term."""
    keys, vals = _log_keys_vals(loss_log_dict)
    dev = next((v.device for v in vals if v.is_cuda), torch.device("cpu"))
    host = torch.stack([v.detach().float().to(dev).reshape(()) for v in vals]).tolist()
    return _log_from_host(loss_log_dict, keys, host)


_PINNED_LOGS = []          # small ring of pinned host buffers for the asynchronous read of the loss log


def loss_log_to_host_async(loss_log_dict):
    """First half of convert_loss_log_to_numpy for device-resident logs: the terms are gathered and copied to pinned host memory behind the
    work queued so far, an event marks the copy.  -> handle for ``loss_log_finish`` (None: nothing on the device, convert synchronously)."""
    keys, vals = _log_keys_vals(loss_log_dict)
    dev = next((v.device for v in vals if v.is_cuda), None)
    if dev is None:
        return None
    stacked = torch.stack([v.detach().float().to(dev).reshape(()) for v in vals])
    if len(_PINNED_LOGS) < 4:
        _PINNED_LOGS.append(torch.empty(64, dtype=torch.float32).pin_memory())
    buf = _PINNED_LOGS[0]
    _PINNED_LOGS.append(_PINNED_LOGS.pop(0))            # rotate: a buffer is reused four reads later, long after its copy has been read
    host = buf[:stacked.numel()]
    host.copy_(stacked, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return loss_log_dict, keys, host, ev


def loss_log_finish(handle):
    loss_log_dict, keys, host, ev = handle
    ev.synchronize()
    return _log_from_host(loss_log_dict, keys, host.tolist())


LATE_LOSS_SYNC = os.environ.get("GD_LATE_LOSS_SYNC", "1") == "1"


def _after_opt_pass(controller, log_dev, i, skip_optim_steps, edit_type, use_adaptive_optimization, removal_loss_value_in, global_loss_log_dict,
                    handle=None):
    """:284-306 — the logged terms to the host (the one sync of an optimisation step) and the adaptive weight schedule.  ``handle``: the
    read was started earlier (loss_log_to_host_async): only its event is waited for."""
    out_loss_log_dict = loss_log_finish(handle) if handle is not None else convert_loss_log_to_numpy(log_dev)     # :284 (host sync)
    if use_adaptive_optimization:
        if edit_type == "geometry_editor":
            adaptive_optimization_step_editing(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps=NUM_DDIM_STEPS,
                                               removal_loss_value_in=removal_loss_value_in)
        elif edit_type == "geometry_remover":
            adaptive_optimization_step_remover(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps=NUM_DDIM_STEPS,
                                               removal_loss_value_in=removal_loss_value_in)
    global_loss_log_dict[i] = out_loss_log_dict


def init_latent(latent, model, height, width, generator, batch_size):
    """generic_torch.py:266-273."""
    if latent is None:
        latent = torch.randn((1, 4, height // 8, width // 8), generator=generator)
    latents = latent.expand(batch_size, 4, height // 8, width // 8).to(model.device)
    return latent, latents


def _resize_mask(m, s):
    return F.interpolate(m, size=(s, s), mode="bilinear", align_corners=False, antialias=False)


def _coords_dtype(like: torch.Tensor):
    """Dtype the warp coordinates are rounded through where the reference writes ``.type_as(<a model tensor>)`` (U/editor.py:148,389): fp16
    on its GPU path.  A bf16 model must NOT round pixel coordinates through bf16 (8 mantissa bits = 2 pixels at 512^2 — found by the
    full-width 512^2 loop test: the inpaint region lost its only 64^2 pixel); it keeps the reference's fp16 rounding."""
    return torch.float16 if like.dtype == torch.bfloat16 else like.dtype


@torch.no_grad()
def text2image_ldm_stable(model, prompt: List[str], controller, num_inference_steps: int = 20, guidance_scale: Optional[float] = 7.5,
                          generator=None, latent=None, uncond_embeddings=None, start_time=50, return_type="image",
                          transform_coordinates=None, mask_obj=None, optimize_steps=0.2, latent_replace=0.2, lr=0.0,
                          optimize_embeddings=False, optimize_latents=False, ddim_latents=None, ddim_noise=None,
                          edit_type="geometry_editor", fast_start_steps=0.0, num_first_optim_steps=5,
                          use_adaptive_optimization=True, adain_latents_steps=0.95, use_optimizer=False,
                          removal_loss_value_in=-1.5, image_size=None, timings: Optional[Dict[str, float]] = None):
    """editor.py:65-423."""
    global_loss_log_dict = {}
    skip_optim_steps = SKIP_OPTIM_STEPS
    batch_size = len(prompt)
    controller.persistent_tables = os.environ.get("GD_PERSISTENT_TABLES", "1") == "1"      # lets captured CFG-pass graphs be reused by the next edit (one edit at a time per
    #                                          process, like the reference's module-level singletons)
    register_attention_control_diffusers(model, controller, transform_coordinates)
    height = width = image_size or IMAGE_SIZE

    tok = model.tokenizer
    text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    text_embeddings = encode_text(model, text_input.input_ids)
    max_length = text_input.input_ids.shape[-1]
    if uncond_embeddings is None:
        uncond_input = tok([UNCOND_TEXT] * batch_size, padding="max_length", max_length=max_length, return_tensors="pt", truncation=True)
        uncond_embeddings_ = encode_text(model, uncond_input.input_ids)
    else:
        uncond_embeddings_ = None

    latent, latents = init_latent(latent[:1], model, height, width, generator, batch_size)
    model.scheduler.set_timesteps(num_inference_steps)
    for p in model.unet.parameters():
        p.requires_grad = False
    context_save = None
    timesteps = model.scheduler.timesteps[-start_time:]
    T = len(timesteps)

    if transform_coordinates is not None:                                            # :147-149 (512^2 mask warp, once)
        t_coords_m = reshape_transform_coords(transform_coordinates.to(model.device).float(), in_mat_shape=controller.image_mask.shape)
        t_coords_m = t_coords_m.tile(controller.image_mask.shape[0], 1, 1, 1).to(_coords_dtype(text_embeddings))
        controller.mask_new_warped = binarize_tensor(
            warp_grid_edit(controller.image_mask[:, None].to(model.device).float(), t_coords_m)).type_as(text_embeddings)

    is_geo = type(controller).__name__.startswith("AttentionGeometry")
    # parity-preserving saving: with an inversion trajectory the CFG pass runs 3 batch rows, not 4
    skip_ref = ddim_latents is not None and batch_size == 2 and SKIP_UNCOND_REF

    # (not with store_attention_maps: the processors then leave the token-major path the stash is written and read on — ADVICE r05; the
    #  flag is set on the controller before this driver runs, U/editor.py:538)
    ref_from_opt = (REF_FROM_OPT and skip_ref and is_geo and ref_from_opt_supported() and getattr(controller, "supports_token_major", False)
                    and not getattr(controller, "store_attention_maps", False))

    ref_ahead = int(REF_AHEAD) if (REF_AHEAD and ref_from_opt and fast_start_steps == 0.0 and hasattr(controller, "collect_ahead")) else 0

    def is_opt_step(j):
        return j < T and (j < optimize_steps * T) and (j % skip_optim_steps == 0) and (j >= fast_start_steps * T)      # :181

    def cfg_pass(lat, ctx, tt, ahead=None):
        # the reference decorates this driver with @torch.no_grad() (editor.py:64); the graph / token-major fast paths depend on it
        assert not torch.is_grad_enabled(), "text2image_ldm_stable: the CFG pass must run without autograd"
        if ref_from_opt and controller.ref_stash_serial is not None and controller.ref_stash_t == int(tt):
            # a step with an optimisation pass: the reference row's layer tensors are already there (attention_processors.ref_stash)
            global REF_FROM_OPT_PASSES
            REF_FROM_OPT_PASSES += 1
            set_attn_processor_for_edit(model, coords_base=(1, 1), coords_edit=(1, 2), use_cfg=True, n_batch=2)
            controller.use_ref_stash, controller._ref_pos = True, 0
            try:
                return diffusion_step(model, controller, lat, ctx, tt, guidance_scale, transform_coords=transform_coordinates,
                                      skip_uncond_ref=True, ref_from_stash=True)
            finally:
                controller.use_ref_stash, controller.ref_stash_t = False, None
        if skip_ref and ahead is not None:
            # the next step optimises: its reference sample rides along as row 0 and leaves its layer tensors (REF_AHEAD)
            set_attn_processor_for_edit(model, coords_base=(2, 3), coords_edit=(3, 4), use_cfg=True, n_batch=4)
            controller.collect_ahead = True
            controller.ref_stash_serial = controller.ref_stash_t = None
            try:
                out = diffusion_step(model, controller, lat, ctx, tt, guidance_scale, transform_coords=transform_coordinates,
                                     skip_uncond_ref=True, ref_ahead=ahead)
            finally:
                controller.collect_ahead = False
            controller.ref_stash_t = int(ahead[1]) if controller.ref_stash_serial is not None else None
            return out
        if skip_ref:
            set_attn_processor_for_edit(model, coords_base=(1, 2), coords_edit=(2, 3), use_cfg=True, n_batch=3)
            return diffusion_step(model, controller, lat, ctx, tt, guidance_scale, transform_coords=transform_coordinates,
                                  skip_uncond_ref=True)
        set_attn_processor_for_edit(model, coords_base=(2, 3), coords_edit=(3, 4), use_cfg=True)           # :343,366
        return diffusion_step(model, controller, lat, ctx, tt, guidance_scale, transform_coords=transform_coordinates)

    opt_pass = GraphedOptPass(model, transform_coordinates, guidance_scale)
    # REF_AHEAD = 1: the reference rows of ALL optimisation steps in one no-grad pass, now (their latents are the inversion trajectory's,
    # their text row never changes: the embedding update touches the edit row only, U/optimization.py:240-249)
    ref_rows = None          # step -> row of the batched pass
    if ref_ahead == 1:
        ref_rows = _reference_rows_pass(model, controller, [i for i in range(T) if is_opt_step(i)], timesteps, latents, ddim_latents,
                                        text_embeddings[0:1], transform_coordinates)
    if PASS_TIMES is not None:          # development aid (GD_PASS_TIMES=1): device time per kind of pass, HIP events on the launch stream
        cfg_pass, opt_pass = _timed_passes(cfg_pass, opt_pass, controller)
    first_optim_complete = False
    # :157-160 concatenates the same two tensors at every step; built ONCE here, so the context of a step is the same tensor OBJECT
    # until an optimisation pass replaces it — which is what the captured passes' text-row K / V cache keys on (graphs.GraphedUNet)
    context_fixed = torch.cat([uncond_embeddings_, text_embeddings]) if uncond_embeddings_ is not None else None
    for i, t in enumerate(timesteps):
        if uncond_embeddings_ is None:
            context = torch.cat([uncond_embeddings[i].expand(*text_embeddings.shape), text_embeddings])
        else:
            context = context_fixed

        if not is_geo:
            latents = diffusion_step(model, controller, latents, context, t, guidance_scale, transform_coords=transform_coordinates)
            continue

        clear_controller_loss(controller)
        # the reference row of the NEXT step, if that step optimises (REF_AHEAD): the trajectory entry the end of this step puts in row 0
        ahead = None
        if ref_ahead == 2 and is_opt_step(i + 1) and len(ddim_latents) - 2 - i >= 0:
            ahead = (ddim_latents[len(ddim_latents) - 2 - i].type_as(latents.detach()), timesteps[i + 1])
        if (i < optimize_steps * T) and (i % skip_optim_steps == 0) and (i >= fast_start_steps * T):      # :181
            l_eff = lr * (50 - i) * skip_optim_steps * (50 / (NUM_DDIM_STEPS + 1e-8))                      # :207
            set_attn_processor_for_edit(model, coords_base=(0, 1), coords_edit=(1, 2), use_cfg=False)    # :213
            # this step's reference row already went through the UNet (the batched pass / the CFG pass of the step before): the edit row alone
            if ref_rows is not None and i in ref_rows[0]:
                ent = ref_rows[1]
                ent["table"].copy(ref_rows[0][i])                # row m of every kept tensor -> the persistent one-row tensors, one launch
                controller.ref_stash, controller.ref_stash_serial, controller.ref_stash_t = ent["slots"], ent["serial"], int(t)
            edit_row_only = bool(ref_ahead and controller.ref_stash_serial is not None and controller.ref_stash_t == int(t)
                                 and latents.shape[0] == 2)
            controller.collect_ref = ref_from_opt and not edit_row_only
            n0 = ops.sumsq(latents[-1].detach().float().contiguous())                                      # orig_norm^2 (:219)
            ctx_cur = context if context_save is None else context_save
            lat_cur = latents
            # first pass of a REMOVAL edit: reference and edit rows are the same sample and stay the same through the whole UNet (the
            # remover returns replace_out, which then equals the vanilla output; the editor's blend moves the object in the edit row, so
            # its rows differ from the second hooked layer on).  One host comparison per edit -> see rows_identical
            controller.rows_identical = bool(TIE_IDENTICAL_ROWS and i == 0 and is_geo and batch_size == 2 and latents.shape[0] == 2
                                             and type(controller).__name__ == "AttentionGeometryRemover"
                                             and torch.equal(latents[0], latents[1]) and torch.equal(ctx_cur[2], ctx_cur[3]))
            edit_row_only = edit_row_only and not controller.rows_identical
            # :183-187 — the first optimised step after a fast start runs num_first_optim_steps iterations and keeps the inputs of
            # the lowest-loss one (:236-239); every other step runs one iteration and keeps its updated latents (:252-254)
            if (not first_optim_complete) and fast_start_steps > 0.0:
                num_optim_steps = max(int(num_first_optim_steps), 1)
                first_optim_complete = True
            else:
                num_optim_steps = 1
            best_loss, latents_new, context_new = 1e8, None, None
            for _opt_iter in range(num_optim_steps):
                # :218-273 — forward with losses + autograd back to latent / embedding (one hipGraph, reused across edits)
                if edit_row_only and num_optim_steps == 1:
                    global REF_AHEAD_PASSES
                    REF_AHEAD_PASSES += 1
                    controller.use_ahead, controller._ref_pos = True, 0
                    try:
                        g_lat, g_ctx, latents_in, context_in = opt_pass.grads(controller, lat_cur, ctx_cur, t, edit_row_only=True)
                    finally:
                        controller.use_ahead = False
                else:
                    g_lat, g_ctx, latents_in, context_in = opt_pass.grads(controller, lat_cur, ctx_cur, t)
                if num_optim_steps > 1:
                    loss_val = float(controller.loss)                                                      # :236 (host sync)
                    if loss_val < best_loss:
                        best_loss = loss_val
                        latents_new, context_new = latents_in.detach().clone(), context_in.detach().clone()
                lat_upd, ctx_upd = _apply_latent_update(latents_in, g_lat, context_in, g_ctx, l_eff, controller.mask_new_warped[:1])
                if num_optim_steps == 1:
                    latents_new, context_new = lat_upd, ctx_upd
                lat_cur, ctx_cur = lat_upd.detach(), ctx_upd.detach()
                # The loss log is read on the host (:284, one sync) for the adaptive schedule, which only edits the loss WEIGHTS — used by the
                # next optimisation pass, not by the CFG pass that follows (no losses there).  With one optimisation iteration per step
                # (every driver of the reference) the copy to (pinned) host memory is queued HERE, behind the optimisation pass, and
                # waited for only after that CFG pass has been queued too: the device works through the CFG pass while the host runs the
                # schedule and prepares the next step, instead of idling while the host does (LATE_LOSS_SYNC).
                log_dev = controller.loss_log_dict
                late = LATE_LOSS_SYNC and num_optim_steps == 1
                log_handle = loss_log_to_host_async(log_dev) if late else None
                if not late:
                    _after_opt_pass(controller, log_dev, i, skip_optim_steps, edit_type, use_adaptive_optimization, removal_loss_value_in, global_loss_log_dict)
                clear_controller_loss(controller)
                controller.cur_step -= 1                                                                   # :307
                controller.rows_identical = False
            if optimize_latents:                                                                           # :312-316
                latents = latents_new.detach()
                last = latents[-1].float().contiguous()
                latents = torch.cat([latents[:-1], ops.norm_rescale(last, n0, ops.sumsq(last))[None].to(latents.dtype)], 0)
            if context_new is not None and optimize_embeddings:                                            # :319-322
                context = context_new.detach()
                context_save = context
            latents = cfg_pass(latents, context, t, ahead)                                                   # :343-351
            if late:
                _after_opt_pass(controller, log_dev, i, skip_optim_steps, edit_type, use_adaptive_optimization, removal_loss_value_in, global_loss_log_dict,
                                handle=log_handle)
        elif i < fast_start_steps * T:
            pass
        else:
            if context_save is not None:
                context = context_save
            latents = cfg_pass(latents, context, t, ahead)                                                   # :366-368

        if ddim_latents is not None:                                                                       # :375-377
            i_n = len(ddim_latents) - 2 - i
            latents = torch.cat([ddim_latents[i_n].type_as(latents.detach()), latents[-1:].detach()], 0)

        if type(controller).__name__ != "AttentionGeometryRemover":                                        # :382-399 latent warp
            if (i < T * latent_replace and mask_obj is not None) or (i < T * fast_start_steps):
                s = latents.shape[-1]
                t_coords = reshape_transform_coords(transform_coordinates.to(model.device).float(), in_mat_shape=latents[1:].shape)
                t_coords = t_coords.to(_coords_dtype(latents))
                i_mask = (_resize_mask(controller.mask_new_warped[:1].detach().float(), s) > 0.5) * 1.0
                i_mask = i_mask.type_as(latents)
                warped = warp_grid_edit(latents[-2:-1].detach().clone(), t_coords)
                base = latents[:1] if i < T * fast_start_steps else latents[-1:]
                latents = torch.cat([latents[:-1], base * (1 - i_mask) + i_mask * warped.type_as(latents)], 0)

    release_opt_graph(controller)
    if return_type == "image":
        image = latent2image(model.vae, latents)
    else:
        image = latents
    return image, latent, global_loss_log_dict


@torch.no_grad()
def post_process(image, image_mask, edited, transform_coordinates, mask_new_warped, edit_type, as_numpy: bool = True):
    """U/editor.py:660-693 — the histogram post-process of the decoded edit, entirely on the device.
    geometry_editor: the input image warped by the edit's transform is pasted over the input wherever the moved object lands, the pixels
    the object left are masked out, and the decoded edit's per-channel histograms are matched to that composite inside the remaining
    region; geometry_remover: matched to the input image outside the object mask.
    image [H,W,3] uint8 (numpy or tensor), image_mask [H,W] in {0,1}, edited [H,W,3] uint8 tensor on the device, transform_coordinates
    [1,H,W,3], mask_new_warped [>=1,1,H,W] -> the reference's float64 image ``lut[edited]`` (numpy unless ``as_numpy=False``).
    The integer arithmetic is the reference's: masks are {0,1}, so mask*uint8 sums are exact and the ``astype('uint8')`` truncations act
    on the same values."""
    dev = edited.device
    # (device tensors are taken as they are: the drivers upload both during the pre-pass, an upload here would wait for the decode)
    img = image.to(dev) if torch.is_tensor(image) else torch.as_tensor(np.ascontiguousarray(image)).to(dev)          # [H,W,3] uint8
    m_im = torch.as_tensor(np.asarray(image_mask) if not torch.is_tensor(image_mask) else image_mask).to(dev).float()
    if edit_type == "geometry_editor":
        img_t = (img[None].permute(0, 3, 1, 2) / 255.0).float()
        warped = warp_grid_edit(img_t, transform_coordinates.to(dev).float())                   # fp16 [1,3,H,W] (warp_utils.py:176)
        moved = (warped[0].permute(1, 2, 0).float() * 255.0).to(torch.uint8)                    # the warped image as uint8 (truncated)
        m_edit = mask_new_warped[0, 0].detach().float()
        keep = 1.0 - ((m_edit + m_im) > 0.5).float()                                            # neither the object's old nor its new place
        target = (keep[..., None] * img.float() + m_edit[..., None] * moved.float()).to(torch.uint8)
        region = ((m_edit + keep) > 0.5).float()
        out = masked_histogram_matching(edited, target, region, region)
    else:
        out = masked_histogram_matching(edited, img, 1.0 - m_im)
    return out.cpu().numpy() if as_numpy else out


def run_and_display(ldm_stable, prompts, controller, latent=None, run_baseline=False, generator=None, uncond_embeddings=None,
                    verbose=True, transform_coordinates=None, mask_obj=None, optimize_steps=0.0, latent_replace=0.0, lr=0.0,
                    optimize_embeddings=False, optimize_latents=True, ddim_latents=None, ddim_noise=None, edit_type="geometry_editor",
                    fast_start_steps=0.0, num_first_optim_steps=1, use_adaptive_optimization=True, removal_loss_value_in=-1.5,
                    return_type="image", image_size=None):
    """editor.py:713-718 (``verbose`` display is a notebook helper and is ignored)."""
    return text2image_ldm_stable(ldm_stable, prompts, controller, latent=latent, num_inference_steps=NUM_DDIM_STEPS,
                                 guidance_scale=GUIDANCE_SCALE, generator=generator, uncond_embeddings=uncond_embeddings,
                                 transform_coordinates=transform_coordinates, mask_obj=mask_obj, optimize_steps=optimize_steps,
                                 latent_replace=latent_replace, lr=lr, optimize_embeddings=optimize_embeddings,
                                 optimize_latents=optimize_latents, ddim_latents=ddim_latents, ddim_noise=ddim_noise,
                                 edit_type=edit_type, fast_start_steps=fast_start_steps, num_first_optim_steps=num_first_optim_steps,
                                 use_adaptive_optimization=use_adaptive_optimization, removal_loss_value_in=removal_loss_value_in,
                                 return_type=return_type, image_size=image_size)


_SIDE_STREAMS = {}


@contextlib.contextmanager
def side_stream():
    """Run the body on a second HIP stream of DEVICE; the current stream waits for it on exit.  Host synchronisations inside the body
    (``nonzero`` / ``.item()`` / downloads) then wait for the body's own kernels only, not for work already queued on the current
    stream.  Tensors made inside are used on the current stream afterwards and are released only after the edit's final download (a
    full synchronisation), so the caching allocator cannot hand their blocks out while the current stream still reads them."""
    dev = torch.device(DEVICE)
    if dev.type != "cuda":
        yield
        return
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(dev)
    if side is None:
        side = _SIDE_STREAMS[dev] = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        yield
    main.wait_stream(side)


def _drain_ahead(ahead):
    """The caller is already unwinding with an exception of its own: wait for the work started beside it (nothing may keep launching
    behind the caller's back) and keep ITS exception out of the way — logged, not raised (`finally: ahead.result()` replaced
    the inversion's exception with the pre-pass's when both failed)."""
    try:
        ahead.result()
    except Exception as e:  # noqa: BLE001
        import sys
        print(f"[geodiffuser_amd] the pre-pass beside a failing inversion failed as well: {e!r}", file=sys.stderr, flush=True)


PREPASS_THREAD = os.environ.get("GD_PREPASS_THREAD", "1") == "1"
_WORKER = None


class _Now:
    def __init__(self, fn):
        with torch.no_grad(), side_stream():
            self.v = fn()

    def result(self):
        return self.v


class _Ahead:
    """``fn()`` on the module's worker thread, on the side stream (see ``start_ahead``)."""

    def __init__(self, fn, dev):
        global _WORKER
        from concurrent.futures import ThreadPoolExecutor
        if _WORKER is None:
            _WORKER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="gd-prepass")   # one long-lived thread: library handles stay warm
        self.dev = dev.index if dev.index is not None else torch.cuda.current_device()
        self.side = _SIDE_STREAMS.get(dev)
        if self.side is None:
            self.side = _SIDE_STREAMS[dev] = torch.cuda.Stream(dev)
        self.side.wait_stream(torch.cuda.current_stream(dev))        # uploads the caller has already queued
        self.fut = _WORKER.submit(self._run, fn)
        from . import graphs
        graphs.IN_FLIGHT.append(self.fut)                            # a graph capture waits for it (graphs.wait_in_flight)
        self.fut.add_done_callback(lambda f: graphs.IN_FLIGHT.remove(f) if f in graphs.IN_FLIGHT else None)

    def _run(self, fn):
        torch.cuda.set_device(self.dev)                              # current device, grad mode and current stream are per thread
        with torch.no_grad(), torch.cuda.stream(self.side):
            return fn()

    def result(self):
        v = self.fut.result()                                        # re-raises what fn raised
        torch.cuda.current_stream(self.dev).wait_stream(self.side)
        return v


def start_ahead(fn):
    """Start ``fn()`` — host work with device round trips that does not depend on what the caller queues next — and return a handle;
    ``handle.result()`` gives its value once the caller needs it.  On a GPU the body runs on a worker thread with the side stream
    current, so it overlaps BOTH the device work the caller queues meanwhile and the caller's own host time (a graph launch holds the
    host for milliseconds and releases the GIL); ``result()`` makes the caller's stream wait for the side stream.  The lifetime rule of
    ``side_stream`` applies to the tensors ``fn`` makes.  GD_PREPASS_THREAD=0 (or a CPU device): ``fn`` runs here, now."""
    dev = torch.device(DEVICE)
    if dev.type != "cuda" or not PREPASS_THREAD:
        return _Now(fn)
    return _Ahead(fn, dev)


def perform_geometric_edit(image, depth, image_mask, transform_in, prompt="", ldm_stable_model=None, tokenizer_model=None,
                           scheduler_in=None, cross_replace_steps={"default_": 0.95}, self_replace_steps=0.95, optimize_steps=0.6,
                           lr=0.03, latent_replace=0.6, optimize_embeddings=True, optimize_latents=True, obj_edit_step=1.0,
                           perform_inversion=True, guidance_scale=7.5, skip_optim_steps=1, num_ddim_steps=50, splatting_radius=1.3,
                           edit_type="geometry_editor", image_stitch=None, progress=None, fast_start_steps=0.0,
                           num_first_optim_steps=1, loss_weights_dict=None, return_loss_log_dict=False, splatting_tau=1.0,
                           splatting_points_per_pixel=15, use_adaptive_optimization=True, return_attention_maps=False, unet_path="",
                           use_optimizer=True, removal_loss_value_in=-1.5, return_latents=False):
    """editor.py:428-710.  Returns ``images`` (2 uint8 [H,W,3]: reference reconstruction, edit)
    ``[, loss_log_dict][, attention_store]``; ``return_latents=True`` (extension for parity tests) appends the final
    latents [2,4,h,w]."""
    # the reference switches autograd off globally and never restores it (editor.py:467); the previous mode is restored here
    prev_grad = torch.is_grad_enabled()
    torch.set_grad_enabled(False)
    try:
        return _perform_geometric_edit(**{k: v for k, v in locals().items() if k != "prev_grad"})
    finally:
        torch.set_grad_enabled(prev_grad)


def _perform_geometric_edit(image, depth, image_mask, transform_in, prompt, ldm_stable_model, tokenizer_model, scheduler_in,
                            cross_replace_steps, self_replace_steps, optimize_steps, lr, latent_replace, optimize_embeddings,
                            optimize_latents, obj_edit_step, perform_inversion, guidance_scale, skip_optim_steps, num_ddim_steps,
                            splatting_radius, edit_type, image_stitch, progress, fast_start_steps, num_first_optim_steps,
                            loss_weights_dict, return_loss_log_dict, splatting_tau, splatting_points_per_pixel,
                            use_adaptive_optimization, return_attention_maps, unet_path, use_optimizer, removal_loss_value_in,
                            return_latents):
    global SEED, TOKENIZER, LDM_STABLE, SCHEDULER, PROGRESS_BAR, GUIDANCE_SCALE, SKIP_OPTIM_STEPS, NUM_DDIM_STEPS, UNET_NAME
    torch.manual_seed(SEED)
    torch.cuda.manual_seed_all(SEED)
    max_opt = max(self_replace_steps, cross_replace_steps["default_"])
    if optimize_steps > max_opt:
        optimize_steps = max_opt
    # editor.py:487-490 writes these onto editor.SPLATTER — an object of its own that nothing reads (the warps run on
    # warp_utils.SPLATTER, which keeps radius 1.3 / tau 1.0 / K 15 unless a warp_grid_edit call passed explicit values).  A drop-in does
    # the same: the arguments are IGNORED by default; HONOUR_SPLAT_ARGS (GD_HONOUR_SPLAT_ARGS=1) applies them to the live splatter.
    SPLATTER.radius, SPLATTER.tau, SPLATTER.points_per_pixel = splatting_radius, splatting_tau, splatting_points_per_pixel
    warp_utils.SPLATTER.clear_cache()
    if HONOUR_SPLAT_ARGS:
        warp_utils.SPLATTER.radius = splatting_radius
        warp_utils.SPLATTER.tau = splatting_tau
        warp_utils.SPLATTER.points_per_pixel = splatting_points_per_pixel
    PROGRESS_BAR = progress
    GUIDANCE_SCALE, SKIP_OPTIM_STEPS, NUM_DDIM_STEPS = guidance_scale, skip_optim_steps, num_ddim_steps
    if edit_type not in ("geometry_editor", "geometry_remover"):
        if edit_type in ("geometry_stitch", "geometry_stitch_single"):
            raise NameError("AttentionGeometryStitch is not defined (it is not defined in the reference either, editor.py:618-621)")
        raise NotImplementedError(edit_type)

    image = np.asarray(image)
    image_mask = torch.as_tensor(np.asarray(image_mask)).float()
    H = image.shape[0]
    ldm_stable, tokenizer, scheduler = LDM_STABLE, TOKENIZER, SCHEDULER
    if scheduler_in is not None and (unet_path == "" or unet_path == UNET_NAME):
        ldm_stable, tokenizer, scheduler = ldm_stable_model, tokenizer_model, scheduler_in
    elif ldm_stable is None or tokenizer is None or scheduler is None or (unet_path != "" and unet_path != UNET_NAME):
        ldm_stable, tokenizer, scheduler = load_model(diffusion_model=DIFFUSION_MODEL, unet_path=unet_path, device=DEVICE)
        UNET_NAME = unet_path or DIFFUSION_MODEL
    LDM_STABLE, TOKENIZER, SCHEDULER = ldm_stable, tokenizer, scheduler

    prompts = [prompt, prompt]
    cls = AttentionGeometryEdit if edit_type == "geometry_editor" else AttentionGeometryRemover

    def prepass():
        # the geometry pre-pass (editor.py:497-503) and the controller do not depend on the inversion: they run beside it
        t_coords_depth, _, amodal = vis_utils.get_transform_coordinates(image, depth, image_mask.numpy(), transform_in=transform_in,
                                                                        focal_length=550 * H / 512.0 if H != 512 else 550,
                                                                        return_mesh=True, device=str(DEVICE), as_torch=True, preview=False)
        controller = cls(prompts, NUM_DDIM_STEPS, cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps,
                         equalizer=None, local_blend=None, controller=None, image_mask=image_mask.numpy(), empty_scale=0.0, use_all=False,
                         obj_edit_step=obj_edit_step, tokenizer=tokenizer, device=DEVICE, mode=MODE)
        controller.amodal_mask = torch_erode(amodal.float())                                              # :633
        # (image and mask on the device for the post-process)
        return t_coords_depth[None].detach(), controller, torch.from_numpy(np.ascontiguousarray(image)).to(DEVICE), image_mask.to(DEVICE)

    ahead = start_ahead(prepass)
    null_inversion = NullInversion(ldm_stable, num_ddim_steps=NUM_DDIM_STEPS, uncond_text=UNCOND_TEXT, device=DEVICE,
                                   progress_bar=PROGRESS_BAR, guidance_scale=GUIDANCE_SCALE)
    try:
        (_, _), x_t, uncond_embeddings, ddim_latents, ddim_noise = null_inversion.invert(image, prompt, offsets=(0, 0, 0, 0), verbose=False,
                                                                                          perform_inversion=perform_inversion, image_2=None)
    except BaseException:
        _drain_ahead(ahead)                 # the inversion's exception is the one to report; the pre-pass's, if any, is logged
        raise
    transform_coordinates, controller, image_dev, mask_dev = ahead.result()
    if return_attention_maps:
        controller.store_attention_maps = True
    if loss_weights_dict is not None:                                                                      # :636-638
        controller.loss_weight_dict = loss_weights_dict
        controller.default_loss_weights = loss_weights_dict

    m_i_1 = image_mask[None, None]
    out, _, global_loss_log_dict = run_and_display(
        ldm_stable, prompts, controller, run_baseline=False, latent=x_t, uncond_embeddings=uncond_embeddings,
        transform_coordinates=transform_coordinates, mask_obj=m_i_1, optimize_steps=optimize_steps, latent_replace=latent_replace, lr=lr,
        optimize_embeddings=optimize_embeddings, optimize_latents=optimize_latents, ddim_latents=ddim_latents, verbose=False,
        ddim_noise=ddim_noise, edit_type=edit_type, fast_start_steps=fast_start_steps, num_first_optim_steps=num_first_optim_steps,
        use_adaptive_optimization=use_adaptive_optimization, removal_loss_value_in=removal_loss_value_in,
        return_type="latents", image_size=H)
    final_latents = out if return_latents else None
    decoded = latent2image(ldm_stable.vae, out, as_tensor=True)                    # [2, H, W, 3] uint8, still on the device
    # :660-693 on the device: one download of the finished images instead of image -> host -> device -> host
    edited = post_process(image_dev, mask_dev, decoded[-1], transform_coordinates, controller.mask_new_warped, edit_type)
    images = [decoded[0].cpu().numpy(), edited]
    ldm_stable.unet.set_attn_processor(VanillaAttentionProcessor())                                        # :698
    ret = [images]
    if return_loss_log_dict:
        ret.append(global_loss_log_dict)
    if return_attention_maps:
        ret.append(controller.attention_store)
    if return_latents:
        ret.append(final_latents)
    return ret[0] if len(ret) == 1 else tuple(ret)


# The reference has no function of this name in its own code (the only ``run_geodiffuser.py`` is a baseline-comparator
# runner); the north star asks for this entry point, so it is an alias.
run_geodiffuser = perform_geometric_edit
