"""ORACLE (test infrastructure, never shipped / never on the product path): CPU restatement of the reference's per-step edit loop.

Restates ``text2image_ldm_stable`` (GeoDiffuser/utils/editor.py:65-423) — optimisation pass with gradient -> ``_update_latent`` ->
adaptive schedule -> classifier-free-guidance pass -> reference-latent replacement -> latent warp — and the attention-processor
protocol (``EditProcessor.__call__``, GeoDiffuser/utils/attention_processors.py:155-228) on top of the controller oracles of
``ref_cpu.py``, in the reference's formulation (materialised maps, per-call rasterisation, unfused losses, autograd), fp32, any
``nn.Module`` UNet that exposes ``attn_processors`` / ``set_attn_processor``.

PINNED: ``tests/test_oracle_golden.py::test_oracle_loop_*`` hold it to the fixtures G18 / G19 / G20, which are the REFERENCE's own
driver run on the same seeded model and inputs (oracle/gen_golden.py).  Used by bench.py's ``cpu_baseline`` for the BASELINE
configs[0] end-to-end case timed on the host cores.
"""
from __future__ import annotations

import numpy as np
import torch

import ref_cpu as O


class OracleVanillaProcessor:
    """VanillaAttentionProcessor (U/attention_processors.py:69-139): plain attention, fp32 softmax over materialised scores."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        out = torch.bmm(O.compute_attention(q, k, attn.scale), v)
        return attn.to_out[1](attn.to_out[0](attn.batch_to_head_dim(out)))


class OracleEditProcessor:
    """EditProcessor (U/attention_processors.py:141-228): q / k / v projections -> head_to_batch_dim -> controller ->
    batch_to_head_dim -> output projection (the SD2.1 attention modules have no group / spatial norm, no residual, rescale 1)."""

    def __init__(self, transform_coords, controller, place_in_unet):
        self.transform_coords, self.controller, self.place_in_unet = transform_coords, controller, place_in_unet

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        is_cross = encoder_hidden_states is not None
        ctx = encoder_hidden_states if is_cross else hidden_states
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        out = self.controller(q, k, v, is_cross, self.place_in_unet, transform_coords=self.transform_coords, scale=attn.scale)
        return attn.to_out[1](attn.to_out[0](attn.batch_to_head_dim(out)))


def register(unet, controller, transform_coords):
    """register_attention_control_diffusers (U/attention_processors.py:26-53)."""
    procs, n = {}, 0
    for name in unet.attn_processors.keys():
        place = "mid" if name.startswith("mid_block") else ("up" if name.startswith("up_blocks") else ("down" if name.startswith("down_blocks") else None))
        if place is None:
            continue
        n += 1
        procs[name] = OracleEditProcessor(transform_coords, controller, place)
    unet.set_attn_processor(procs)
    controller.num_att_layers = n


def _set(controller, coords_base, coords_edit, use_cfg):
    controller.coords_base, controller.coords_edit, controller.use_cfg = coords_base, coords_edit, use_cfg      # U/attention_processors.py:56-67


def _clear(controller):
    controller.loss = 0.0                                                                                         # U/generic.py:41-47
    controller.initialize_loss_log_dict()


def _log_to_host(d):
    return {kind: {k: float(v) for k, v in d[kind].items()} for kind in ("self", "cross")} | {"num_layers": d["num_layers"]}


def text2image_loop(unet, text_embeddings, uncond_embeddings, controller, x_T, ddim_latents, transform_coordinates, mask_obj, *, num_steps,
                    guidance_scale, skip_optim_steps, optimize_steps, latent_replace, lr, edit_type="geometry_editor",
                    removal_loss_value_in=-1.5, progress=None, timings=None, max_steps=None, prediction_type="epsilon"):
    """U/editor.py:65-423 with optimize_embeddings = optimize_latents = True, fast_start_steps = 0, use_adaptive_optimization = True,
    an inversion trajectory given (the configuration of every reference driver).  -> (latents [2,4,h,w], {step: loss log}).
    ``timings`` (dict) accumulates wall seconds / counts of the optimisation and CFG passes; ``max_steps`` stops early (bench sample).
    ``prediction_type="v_prediction"`` (no reference path, parity unpinned: ref_cpu.prev_step_v): the UNet output is v (SD2.1-768)."""
    import time as _time
    remover = edit_type == "geometry_remover"
    ac = O.alphas_cumprod()
    timesteps = O.ddim_timesteps(num_steps)
    register(unet, controller, transform_coordinates)
    latents = x_T[:1].expand(2, *x_T.shape[1:]).clone()                                                           # init_latent
    for p in unet.parameters():
        p.requires_grad = False
    # :147-149 — the 512^2 (image-size) warp of the object mask, once
    S_img = controller.image_mask.shape[-1]
    t_m = O.reshape_transform_coords(transform_coordinates, S_img).tile(controller.image_mask.shape[0], 1, 1, 1)
    controller.mask_new_warped = O.binarize_tensor(O.warp_grid_edit(controller.image_mask[:, None].float(), t_m))
    context_save = None
    logs = {}
    T = len(timesteps)
    for i, t in enumerate(timesteps):
        if max_steps is not None and i >= max_steps:
            break
        t = int(t)
        context = torch.cat([uncond_embeddings, text_embeddings])
        _clear(controller)
        t_start = _time.perf_counter()
        if (i < optimize_steps * T) and (i % skip_optim_steps == 0):                                              # :181
            l_eff = lr * (50 - i) * skip_optim_steps * (50 / (num_steps + 1e-8))                                  # :207
            _set(controller, (0, 1), (1, 2), False)                                                               # :213
            lat_in = latents.detach().float().requires_grad_(True)                                                # :218
            orig_norm = float(O.norm_tensor(lat_in[-1:].detach()))
            ctx_in = (context if context_save is None else context_save).detach().float().requires_grad_(True)
            with torch.enable_grad():
                unet(lat_in, t, encoder_hidden_states=ctx_in[2:])                                                 # diffusion_step(use_cfg=False), :253
                g_lat, g_ctx = torch.autograd.grad(controller.loss, [lat_in, ctx_in])
            lat_new, ctx_new = O.update_latent(lat_in, g_lat, l_eff, controller.mask_new_warped[:1], ctx_in, g_ctx)
            log = _log_to_host(controller.loss_log_dict)                                                          # :284
            O.adaptive_step(controller, i, skip_optim_steps, log["self"]["removal"], num_steps, removal_loss_value_in, remover)
            logs[i] = log
            _clear(controller)
            controller.cur_step -= 1                                                                              # :307
            latents = lat_new.detach()
            latents = torch.cat([latents[:-1], latents[-1:] * orig_norm / float(O.norm_tensor(latents[-1:]))], 0)  # :312-316
            context = ctx_new.detach()                                                                            # :319-322
            context_save = context
            if timings is not None:
                timings["opt_s"] = timings.get("opt_s", 0.0) + _time.perf_counter() - t_start
                timings["opt_n"] = timings.get("opt_n", 0) + 1
        elif context_save is not None:
            context = context_save
        t_start = _time.perf_counter()
        # classifier-free-guidance pass (:343-368, diffusion.py:39-59)
        _set(controller, (2, 3), (3, 4), True)
        with torch.no_grad():
            eps = unet(torch.cat([latents] * 2), t, encoder_hidden_states=context)["sample"]
            eu, ec = eps.chunk(2)
            step_fn = O.prev_step_v if prediction_type == "v_prediction" else O.prev_step
            latents = step_fn(O.cfg_combine(eu, ec, guidance_scale), t, latents, ac, num_steps)
        if timings is not None:
            timings["cfg_s"] = timings.get("cfg_s", 0.0) + _time.perf_counter() - t_start
            timings["cfg_n"] = timings.get("cfg_n", 0) + 1
        if ddim_latents is not None:                                                                              # :375-377
            latents = torch.cat([ddim_latents[len(ddim_latents) - 2 - i].type_as(latents), latents[-1:].detach()], 0)
        if not remover and i < T * latent_replace and mask_obj is not None:                                       # :382-399 latent warp
            s = latents.shape[-1]
            tc = O.reshape_transform_coords(transform_coordinates, s)
            i_mask = (O.resize_bilinear(controller.mask_new_warped[:1].float(), s) > 0.5) * 1.0
            warped = O.warp_grid_edit(latents[-2:-1].detach().clone(), tc)
            latents = torch.cat([latents[:-1], latents[-1:] * (1 - i_mask) + i_mask * warped.type_as(latents)], 0)
        if progress is not None:
            progress(i)
    return latents.detach(), logs


def make_controller(kind, mask, cfg, amodal_input=None):
    """The controllers of the loop fixtures (same constructor arguments and loss weights as oracle/gen_golden.py: run_reference_loop)."""
    if kind == "geometry_editor":
        c = O.GeometryEditOracle(mask, cfg["steps"], cfg["self_replace"], cfg["obj_edit_step"])
        c.amodal_mask = O.torch_erode(torch.from_numpy(amodal_input))
        lw = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30.0, "amodal": 80.5},
              "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15.0, "amodal": 3.5}}
    else:
        c = O.GeometryRemoverOracle(mask, cfg["steps"], 0.9, 1.0)
        lw = {"self": {"sim": 55, "removal": 4.6, "smoothness": 30.0}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15.0}}
    c.default_loss_weights = lw
    c.initialize_default_loss_weights()
    return c
