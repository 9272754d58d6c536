"""Attention processors and the geometry-edit controllers — the drop-in boundary of the hot path.

Mirror of the reference's GeoDiffuser/utils/attention_processors.py: same class names (the driver dispatches on
``type(controller).__name__``), constructor signatures, public state (``loss``, ``loss_log_dict``,
``loss_weight_dict``, ``default_loss_weights``, ``mask_new_warped``, ``amodal_mask``, ``image_mask``, ``cur_step``,
``num_att_layers``, ``coords_base``, ``coords_edit``, ``use_cfg``, ``store_attention_maps``) and the diffusers
attention-processor protocol.  The arithmetic of one hooked layer call —

    vanilla rows, q_warp = blend(q_base, splat(q_base)), edit_out, replace_out, the five losses, the blend —

runs as a handful of HIP launches inside one ``torch.autograd.Function`` (``_EditLayer``), so the UNet's own autograd
graph stays in PyTorch while no ``[f, N, N]`` map, correlation tensor or per-loss temporary is created.

Exactness notes (all derived from the reference's own code):
  * the reference rows never receive gradient: everything taken from them is detached before it enters a loss
    (attention_sharing.py:242, attention_processors.py:428,433,549,555-557,250,309), and their vanilla output feeds
    only their own sample, so d loss / d(reference rows) is exactly zero in the reference too;
  * the point rasterisation is cached per resolution (the reference recomputes an identical result on every call,
    SURVEY.md F3), the 4-NN table of the amodal loss per edit (it depends on the mask only).
"""
from __future__ import annotations

import abc
import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from . import ops
from . import warp_utils
from ._lib import GD_TOKEN_MAJOR
from .attention_sharing import attention_tok, AttentionStore, attention, compute_attention, get_base_edit_qkv, pad_head_dim
from .generic_torch import (CoordinateDistances, binarize_tensor, reshape_attention_mask,
                            reshape_transform_coords, torch_dilate)

try:  # pragma: no cover - diffusers is optional (absent in the build image)
    from diffusers.models.attention_processor import USE_PEFT_BACKEND
except Exception:  # noqa: BLE001
    USE_PEFT_BACKEND = False

DISTANCE_CLASS = CoordinateDistances()


# ---------------------------------------------------------------------------------------------------------
# registration (attention_processors.py:26-67)
# ---------------------------------------------------------------------------------------------------------
def register_attention_control_diffusers(model, controller, transform_coords=None):
    attn_procs = {}
    cross_att_count = 0
    for name in model.unet.attn_processors.keys():
        if name.startswith("mid_block"):
            place_in_unet = "mid"
        elif name.startswith("up_blocks"):
            place_in_unet = "up"
        elif name.startswith("down_blocks"):
            place_in_unet = "down"
        else:
            continue
        cross_att_count += 1
        attn_procs[name] = EditProcessor(transform_coords, controller, place_in_unet)
    model.unet.set_attn_processor(attn_procs)
    controller.num_att_layers = cross_att_count


def set_attn_processor_for_edit(model, perform_edit=True, coords_base=(2, 3), coords_edit=(3, 4), use_cfg=True, n_batch=None):
    """attention_processors.py:56-67.  ``n_batch`` (extension): number of batch entries of the next UNet pass when it differs
    from the reference's 2*len(prompts) / len(prompts) — used by the driver to drop the CFG pass's unused ``uncond_ref`` row."""
    for proc in model.unet.attn_processors.values():
        proc.perform_edit = perform_edit
        proc.controller.coords_base = coords_base
        proc.controller.coords_edit = coords_edit
        proc.controller.use_cfg = use_cfg
        proc.controller.n_batch = n_batch


def _project_qkv(attn, hidden_states, encoder_hidden_states, temb, scale, token_major=False, scaled_q_head_major=False, batched_ok=True):
    """Shared front half of both processors (attention_processors.py:85-120 / :165-203).  ``scaled_q_head_major``: the caller's
    head-major consumer (_EditLayer: the optimisation pass) takes pre-scaled queries too."""
    args = () if USE_PEFT_BACKEND else (scale,)
    if getattr(attn, "spatial_norm", None) is not None:
        hidden_states = attn.spatial_norm(hidden_states, temb)
    input_ndim = hidden_states.ndim
    shape4 = None
    if input_ndim == 4:
        shape4 = hidden_states.shape
        b, c, hh, ww = shape4
        hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
    if getattr(attn, "group_norm", None) is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    lin_args = args if getattr(attn, "linear_takes_scale", False) else ()
    q_scaled = False
    if token_major and batched_ok and BATCHED_QKV and SCALED_Q and not lin_args and getattr(attn, "norm_cross", None) in (None, False):
        fused = _batched_qkv(attn, hidden_states, encoder_hidden_states)
        if fused is not None:
            return fused[:4] + (shape4,) + fused[5:]
    if ((token_major and SCALED_Q) or (scaled_q_head_major and SCALED_Q_OPT)) and not lin_args and type(attn.to_q) is torch.nn.Linear \
            and attn.to_q.bias is None:
        # q' = 16-bit(scale*log2(e) * (x W^T)): the softmax scale and the base change applied as the GEMM's alpha in its fp32
        # epilogue, BEFORE the one rounding to 16 bits (no extra rounding, unlike scaling a rounded q).  The attention kernels
        # then compute p = exp2(q'.k - mu) with one vector instruction per probability (gd_attn_seg_t::q_scaled).  addmm is
        # differentiable, so the optimisation pass takes the same route: its backward kernels then see q' with scale = ln 2
        # (s = ln2 * q'.k), i.e. the probabilities they recompute are EXACTLY the forward's — with unscaled queries the forward's
        # and the backward's scores differ by the rounding of scale*log2(e) inside the kernel.
        w = attn.to_q.weight
        x2 = hidden_states.reshape(-1, hidden_states.shape[-1])
        query = torch.addmm(w[:, 0], x2, w.t(), beta=0.0, alpha=float(attn.scale) * LOG2E).view(*hidden_states.shape[:-1], w.shape[0])
        q_scaled = True
    else:
        query = attn.to_q(hidden_states, *lin_args)
    is_cross = True
    if encoder_hidden_states is None:
        encoder_hidden_states = hidden_states
        is_cross = False
    elif getattr(attn, "norm_cross", None):
        encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
    key = attn.to_k(encoder_hidden_states, *lin_args)
    value = attn.to_v(encoder_hidden_states, *lin_args)
    if token_major:
        # to_q/to_k/to_v output [B, N, heads*D] is consumed in place by the attention kernel's token-major mode: the
        # reference's head_to_batch_dim / batch_to_head_dim permutes (4 copies per layer) disappear
        return query.contiguous(), key.contiguous(), value.contiguous(), is_cross, shape4, lin_args, q_scaled
    query = attn.head_to_batch_dim(query).contiguous()
    key = attn.head_to_batch_dim(key).contiguous()
    value = attn.head_to_batch_dim(value).contiguous()
    return query, key, value, is_cross, shape4, lin_args, q_scaled


def _stacked_weights(attn, names, alpha0):
    """[len(names), out, in] stack of the projections' weights (cached on the module, refreshed when a weight changes); the first one is
    multiplied by ``alpha0`` in fp32 before the ONE rounding to the storage dtype."""
    mods = [getattr(attn, n) for n in names]
    if any(type(m) is not torch.nn.Linear or m.bias is not None for m in mods) or len({m.weight.shape for m in mods}) != 1:
        return None
    ver = tuple(m.weight._version for m in mods) + (mods[0].weight.dtype, mods[0].weight.device, alpha0)
    c = attn.__dict__.get("_gd_stack_" + "".join(names))
    if c is None or c[0] != ver:
        with torch.no_grad():
            ws = [m.weight.detach() for m in mods]
            if alpha0 != 1.0:
                ws[0] = (ws[0].float() * alpha0).to(ws[0].dtype)
            c = attn.__dict__["_gd_stack_" + "".join(names)] = (ver, torch.stack(ws, 0).transpose(1, 2).contiguous())     # [n, in, out]
    return c[1]


def _batched_qkv(attn, hidden_states, encoder_hidden_states):
    """No-grad token-major passes: the three (self-attention) or two (cross-attention k, v) bias-free projections of one input as ONE
    batched GEMM ``[n, M, in] x [n, in, out]`` (the input broadcast over the batch with stride 0: no copy), so their outputs are separate
    contiguous [B, N, C] tensors the attention kernels take as they are.  The query's ``scale*log2(e)`` is folded into its weight in fp32
    before the weight's one rounding (``q_scaled``).  Returns None when the module's projections do not qualify."""
    alpha = float(attn.scale) * LOG2E
    x2 = hidden_states.reshape(-1, hidden_states.shape[-1])
    if encoder_hidden_states is None:
        if x2.shape[0] * x2.shape[1] > 3_000_000:     # measured (tools/bench_qkv.py): at M = 12288, C = 320 three GEMMs beat the batched one
            return None
        w3 = _stacked_weights(attn, ("to_q", "to_k", "to_v"), alpha)
        if w3 is None or w3.shape[1] != x2.shape[1]:
            return None
        qkv = torch.bmm(x2.unsqueeze(0).expand(3, -1, -1), w3)
        shp = (*hidden_states.shape[:-1], w3.shape[2])
        return qkv[0].view(shp), qkv[1].view(shp), qkv[2].view(shp), False, None, (), True
    wq = _stacked_weights(attn, ("to_q",), alpha)
    w2 = _stacked_weights(attn, ("to_k", "to_v"), 1.0)
    if wq is None or w2 is None or wq.shape[1] != x2.shape[1]:
        return None
    e2 = encoder_hidden_states.reshape(-1, encoder_hidden_states.shape[-1])
    if w2.shape[1] != e2.shape[1]:
        return None
    q = torch.mm(x2, wq[0]).view(*hidden_states.shape[:-1], wq.shape[2])
    # K / V of the text rows: the same for every pass over one context.  A captured pass (graphs.GraphedUNet) reads them from
    # persistent buffers that the runner re-fills only when the context changes (16 small GEMMs less per pass)
    kv = KV_PROVIDER.get(id(attn)) if KV_PROVIDER is not None else None
    if kv is None or kv.shape[1] != e2.shape[0] or kv.dtype != e2.dtype:
        kv = torch.bmm(e2.unsqueeze(0).expand(2, -1, -1), w2)
    shp = (*encoder_hidden_states.shape[:-1], w2.shape[2])
    return q, kv[0].view(shp), kv[1].view(shp), True, None, (), True


KV_PROVIDER = None           # {id(cross-attention module): [2, B*77, C] k / v of the current context}, set by graphs.GraphedUNet around a capture
TOKEN_MAJOR = os.environ.get("GD_TOKEN_MAJOR", "1") == "1"
BATCHED_QKV = os.environ.get("GD_BATCHED_QKV", "1") == "1"   # no-grad passes: q / k / v projections of one input as one batched GEMM
SCALED_Q = os.environ.get("GD_SCALED_Q", "1") == "1"      # token-major passes: scale*log2(e) folded into the query projection
SCALED_Q_OPT = os.environ.get("GD_SCALED_Q_OPT", "1") == "1"   # the same for the optimisation pass's hooked (head-major) layers
LN2 = 0.6931471805599453
LOG2E = 1.4426950408889634
FUSED_WARP = os.environ.get("GD_FUSED_WARP", "1") == "1"   # build the warped queries inside the attention launch
# The edit attention with warped queries (q*(1-m) + m*splat(q), U/attention_processors.py:424-428,544-549) is computed only for the rows
# where the soft edit mask m is non-zero (~10 % of a 64^2 map): for m == 0 the warped query IS the reference query, so that row of
# edit_out equals the reference row's attention output, which the same launch computes anyway.  gd_attn_seg_t.q_rows + gd_blend_merge.
# Self-attention layers from 64^2 tokens up (below, the saved work is smaller than the merge launch).
WARP_ROWS = os.environ.get("GD_WARP_ROWS", "1") == "1"
WARP_ROWS_MIN_TOKENS = 64 ** 2
# One hooked optimisation-pass layer in ~12 launches instead of ~20 (include/geodiff_hip.h "R6-R9 fused launches"): merge + blend in one
# pass, both probability maps + the clear of the arg-max scratch in one launch, the loss sums with the removal reduce / fold / assemble
# as the last workgroup's tail, the row dots beside the loss backward, one fold for the attention and removal dq partials.  Same
# arithmetic in the same order; GD_FUSED_LAYER=0 restores the stand-alone launches (the parity tests run both).
FUSED_LAYER = os.environ.get("GD_FUSED_LAYER", "1") == "1"
# The optimisation pass's launches with pre-scaled queries run the forward's PRE-SCALED variant (fixed softmax reference, row sums on
# the matrix pipe: k_attn_fwd_w64 LSUM) instead of the exact-scale rescue variant: with numerator and denominator summed over the same
# rounded probabilities a dominant probability is exact again, which is what the rescue variant was kept for (DESIGN 4a').  fp16 too (r06:
# a probability that leaves fp16's range shows as an infinite row sum and the segment is repeated with exact row maxima).  GD_OPT_PRE=0:
# the round-3 routing.
OPT_PRE = os.environ.get("GD_OPT_PRE", "1") == "1"
# The passes that accumulate losses (use_cfg False: the optimisation pass) hand the layer token-major q / k / v as the projections produce
# them and take a token-major output back: the head_to_batch_dim / batch_to_head_dim permutes of the reference (:201-203,213) and their
# autograd — per layer three copies forward, one back, the gradient's copy and zero fills, and for cross-attention the assembly of the
# key gradient: ~190 launches of ~5 us per pass — become ONE gd_heads_split forward, ONE gd_heads_merge (which also does the blend
# :502-508,617-622) and the same two launches in the backward.  The kernels in between stay head-major.  GD_TOK_OPT=0: the permutes.
TOK_OPT = os.environ.get("GD_TOK_OPT", "1") == "1"
# `self.loss = self.loss + loss` (:494,604) and the four `log[key] = log[key] + term` of generic.py:34-39 — 100 one-float torch launches
# per optimisation pass — are done by the fused loss launch's tail (gd_edit_losses_t.log_acc / loss_in / loss_out): the same f32 adds in
# the same layer order.  GD_TAIL_SUMS=0: the torch adds.
TAIL_SUMS = os.environ.get("GD_TAIL_SUMS", "1") == "1"
# No-grad passes, layers with at most 128 keys (every cross-attention layer: 77 text keys; the 8^2 self-attention layer): the edit rows'
# two attention outputs and their blend (:502-508,617-622 / :831-834) in ONE launch — one workgroup computes both sides for its queries
# and blends in registers (gd_attn_fwd_pair) — instead of two segments + a blend launch.  GD_PAIR_BLEND=0: the two launches.
PAIR_BLEND = os.environ.get("GD_PAIR_BLEND", "1") == "1"
LOG_KEYS = ("sim", "movement", "removal", "smoothness")


def _tok_ok(attn, hidden_states) -> bool:
    """Token-major fast path: no-grad passes with 64-wide heads on the GPU (every SD 2.1 attention layer)."""
    return (TOKEN_MAJOR and not torch.is_grad_enabled() and hidden_states.is_cuda
            and hidden_states.dtype in (torch.float16, torch.bfloat16)
            and attn.to_q.out_features == attn.heads * 64)


def _finish(attn, hidden_states, residual, shape4, lin_args, token_major=False):
    if not token_major:
        hidden_states = attn.batch_to_head_dim(hidden_states)
    hidden_states = attn.to_out[0](hidden_states, *lin_args)
    hidden_states = attn.to_out[1](hidden_states)
    if shape4 is not None:
        b, c, hh, ww = shape4
        hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, hh, ww)
    if getattr(attn, "residual_connection", False):
        hidden_states = hidden_states + residual
    f = getattr(attn, "rescale_output_factor", 1.0)
    return hidden_states if f == 1.0 else hidden_states / f         # x / 1.0 is exact: skipping it changes nothing but a launch


class VanillaAttentionProcessor:
    """attention_processors.py:69-139 (used for inversion and after an edit)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale: float = 1.0):
        residual = hidden_states
        tok = _tok_ok(attn, hidden_states)
        q, k, v, _, shape4, lin_args, qs = _project_qkv(attn, hidden_states, encoder_hidden_states, temb, scale, tok)
        out = attention_tok(q, k, v, attn.scale, attn.heads, q_scaled=qs) if tok else attention(q, k, v, attn.scale)
        return _finish(attn, out, residual, shape4, lin_args, tok)


class EditProcessor:
    """attention_processors.py:141-228."""

    def __init__(self, transform_coords, controller, place_in_unet="down", perform_edit=True, coords_base=(2, 3),
                 coords_edit=(3, 4), use_cfg=True):
        self.transform_coords = transform_coords
        self.place_in_unet = place_in_unet
        self.perform_edit = perform_edit
        self.controller = controller
        self.controller.use_cfg = use_cfg
        self.controller.coords_base = coords_base
        self.controller.coords_edit = coords_edit

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale: float = 1.0):
        residual = hidden_states
        ctrl = self.controller
        # losses (use_cfg False) and stored maps stay on the head-major path
        tok = _tok_ok(attn, hidden_states) and (not self.perform_edit or (
            getattr(ctrl, "supports_token_major", False) and ctrl.use_cfg and not getattr(ctrl, "store_attention_maps", False)))
        # the head-major layer fed with token-major tensors (TOK_OPT): 64-wide heads, 16-bit, the geometry controllers
        tokg = (TOK_OPT and TOKEN_MAJOR and self.perform_edit and not tok and hidden_states.is_cuda
                and hidden_states.dtype in (torch.float16, torch.bfloat16) and attn.to_q.out_features == attn.heads * 64
                and getattr(ctrl, "supports_token_major", False) and not ctrl.use_cfg)
        q, k, v, is_cross, shape4, lin_args, qs = _project_qkv(
            attn, hidden_states, encoder_hidden_states, temb, scale, tok or tokg,
            scaled_q_head_major=self.perform_edit and not tok and getattr(ctrl, "supports_scaled_q_head_major", False), batched_ok=tok)
        if self.perform_edit:
            if tok:
                ctrl.heads_tok = attn.heads
                ctrl.q_scaled_tok = qs
            else:
                ctrl.q_scaled_hm = qs
                ctrl.heads_opt = attn.heads if tokg else 0
            try:
                out = ctrl(q, k, v, is_cross=is_cross, place_in_unet=self.place_in_unet,
                           transform_coords=self.transform_coords, scale=attn.scale, mask=None)
            finally:
                ctrl.heads_tok = 0
                ctrl.heads_opt = 0
                ctrl.q_scaled_tok = False
                ctrl.q_scaled_hm = False
        else:
            out = attention_tok(q, k, v, attn.scale, attn.heads, q_scaled=qs) if tok else attention(q, k, v, attn.scale)
        return _finish(attn, out, residual, shape4, lin_args, tok or tokg)


# ---------------------------------------------------------------------------------------------------------
# masks (attention_processors.py:319-373)
# ---------------------------------------------------------------------------------------------------------
def process_and_cache_masks(masks_cache_dict, h, image_mask, mask_new_warped, amodal_mask, transform_coords, q_base,
                            q_edit_base, coords_dtype=None):
    """Same contract as the reference.  ``coords_dtype``: dtype the resampled coordinates are rounded through
    (the reference's ``.type_as(q_edit_base)`` — fp16 on its GPU path); default = dtype of ``q_edit_base``."""
    names = ["mask_new_warped", "mask_warp", "amodal_mask", "mask_intersection", "mask_1_empty", "mask_wo_edit", "t_coords_q"]
    if h in masks_cache_dict:
        c = masks_cache_dict[h]
        return (masks_cache_dict, image_mask, *[c[n].detach() for n in names])
    masks_cache_dict[h] = {}
    dev = q_base.device
    S = int(np.sqrt(q_base.shape[2]))
    image_mask = image_mask.to(dev).float().detach()
    amodal_mask = amodal_mask.to(dev).float().detach()
    mask_new_warped = mask_new_warped.to(dev).float()
    mask_warp = binarize_tensor(image_mask)[:, None]
    mask_new_warped = reshape_attention_mask(mask_new_warped, in_mat_shape=(1, S))[1:]
    mask_warp = reshape_attention_mask(mask_warp, in_mat_shape=(1, S))[1:]
    amodal_mask = reshape_attention_mask(amodal_mask, in_mat_shape=(1, S))
    amodal_mask = binarize_tensor(amodal_mask - mask_new_warped).detach()
    mask_intersection = binarize_tensor((mask_new_warped + amodal_mask) * mask_warp, 0.5)
    mask_1_empty = binarize_tensor(mask_warp - mask_intersection, 0.5)
    mask_wo_edit = binarize_tensor(torch.ones_like(mask_new_warped) - (mask_1_empty + mask_new_warped))
    cd = coords_dtype if coords_dtype is not None else q_edit_base.dtype
    t_coords_q = reshape_transform_coords(transform_coords.to(dev).float(), in_mat_shape=q_edit_base.shape)
    t_coords_q = t_coords_q.to(cd).tile(q_edit_base.shape[0], 1, 1, 1)
    vals = [mask_new_warped, mask_warp, amodal_mask, mask_intersection, mask_1_empty, mask_wo_edit, t_coords_q]
    for n, t in zip(names, vals):
        masks_cache_dict[h][n] = t.detach()
    return (masks_cache_dict, image_mask, *vals)


def _flat(m: torch.Tensor) -> torch.Tensor:
    return m[0, 0].reshape(-1).float().contiguous()


# Device buffers that outlive a controller: a captured hipGraph of a CFG pass reads the per-resolution tables by address, so an
# edit that wants to reuse the graphs of the previous edit copies its tables INTO these buffers instead of allocating new ones.
_REF_SLOTS: Dict[tuple, dict] = {}       # see _GeometryControllerBase.ref_slots
_PERSISTENT_TABLES: Dict[tuple, torch.Tensor] = {}
# ... which makes them process-wide state: ONE controller at a time may own them (the reference is one edit at a time per process as well: its
# model cache, DISTANCE_CLASS, SPLATTER and GAUSSIAN_FEATURE_SMOOTHER are module globals).  The owner is the controller that built
# its tables last; a hooked call of any other controller whose tables live in these buffers raises instead of reading the owner's geometry.
_TABLE_OWNER: Dict[int, object] = {}     # slot -> weakref to the owning controller (slot 0: a single edit; j: the j-th edit of an EditBatch)


def _check_table_owner(ctrl):
    ref = _TABLE_OWNER.get(getattr(ctrl, "slot", 0))
    owner = ref() if ref is not None else None
    if owner is not None and owner is not ctrl:
        raise RuntimeError("two live edit controllers share the persistent per-resolution tables (GD_PERSISTENT_TABLES=1): another controller "
                           "rebuilt them after this one did; run one edit at a time per process, or set controller.persistent_tables = False")


def _persist(enabled, key: tuple, t: torch.Tensor) -> torch.Tensor:
    """``enabled``: False / True, or the controller itself (its ``persistent_tables`` flag decides and its ``slot`` — the edit's place in an
    EditBatch, 0 for a single edit — is part of the buffer's name: B edits in flight hold B sets of tables)."""
    slot = 0
    if not isinstance(enabled, bool):
        slot = getattr(enabled, "slot", 0)
        enabled = getattr(enabled, "persistent_tables", False)
    if not enabled:
        return t
    key = key + (slot, tuple(t.shape), t.dtype, str(t.device))
    buf = _PERSISTENT_TABLES.get(key)
    if buf is None:
        buf = _PERSISTENT_TABLES[key] = t.clone()
    else:
        buf.copy_(t)
    return buf


_CONST: Dict[tuple, torch.Tensor] = {}
_WEIGHT_VECTORS: Dict[tuple, list] = {}


def _zeros5(dev) -> torch.Tensor:
    k = ("zeros5", str(dev))
    if k not in _CONST:
        _CONST[k] = torch.zeros(5, dtype=torch.float32, device=dev)
    return _CONST[k]


# ---------------------------------------------------------------------------------------------------------
# the fused layer
# ---------------------------------------------------------------------------------------------------------
class _EditLayer(torch.autograd.Function):
    """One hooked attention call of AttentionGeometryEdit / AttentionGeometryRemover (forward :633-664 / :931-959 with
    replace_{self,cross}_attention inlined).  Returns (out [(cb+1)*f, N, D], layer_loss [] f32, terms [5] f32)."""

    @staticmethod
    def forward(ctx, q, k, v, ctrl, is_cross, scale, c, q_pre=False, heads=0, running=False, log_acc=None, ref=None):
        # q_pre: the queries carry scale*log2(e) and ``scale`` is ln 2 (controller forward): every kernel computes
        # exp(scale * q.k - lse) as it stands.  The forward is NOT told (gd_attn_seg_t::q_scaled stays 0, its multiplier becomes
        # ln2 * log2(e) = 1): the pre-scaled variant of the 64-query kernel takes the first key tile's maximum as the softmax reference
        # and never moves it, the exact-scale variant rescues onto a row maximum like k_attn_fwd_mp — after which a DOMINANT probability
        # is exactly 1.0 in 16 bits, as in any online softmax.  The no-grad passes do without that (outputs within the storage
        # rounding either way); the L1 loss terms between two nearly equal attention outputs measure exactly that rounding (the
        # `sim` term of an SDXL-shaped bf16 case moved by 1.6 %), so the optimisation pass pays the 6-9 us per 64^2 launch.
        # heads > 0 (TOK_OPT): q / k / v arrive token-major [B, rows, heads*64] and the output leaves token-major; one split launch here,
        # one merge launch at the end (which also does the blend), everything in between on head-major tensors as before
        # ref (editor.REF_AHEAD): q / k / v hold the EDIT row only; the reference row's token-major q / k / v [1, N | M, heads*64] come from
        # the previous step's CFG pass (detached 16-bit tensors: the reference row never received a gradient here either).  Everything
        # below then runs as for the two-row batch [reference, edit] — the same launches on the same kinds of tensors — and the output /
        # the gradients cover the edit row alone.
        f = c["f"]
        tok_shapes = None
        if ref is not None and not heads:
            raise _lib.GeodiffError("_EditLayer: reference rows handed in need the token-major layer form (64-wide heads, GD_TOK_OPT)")
        ref_hm = None
        if heads:
            tok_shapes = (q.shape, k.shape)
            if ref is not None and ref[0].shape[0] == q.shape[0] and ref[0].dtype == q.dtype and ref[0].shape[2] == q.shape[2]:
                q, k, v, *ref_hm = ops.heads_split((q, k, v, ref[0], ref[1], ref[2]), heads)         # both rows' tensors in one launch (ABI 6)
            else:
                q, k, v = ops.heads_split((q, k, v), heads)
        remover = ctrl._is_remover
        (b0, b1), (e0, e1) = ctrl.coords_base, ctrl.coords_edit
        cb = ctrl.coords_base[-1] * f
        n_van_live = cb // f                                    # vanilla batch rows in front of the edit row in the LIVE tensors
        N, D = q.shape[1], q.shape[2]
        S = c["S"]
        dev, dt = q.device, q.dtype
        grad_mode = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        want_losses = (N >= 32 ** 2) and (not ctrl.use_cfg)
        blend = ctrl.cur_step < int(ctrl.num_steps * ctrl.obj_edit_step)

        if ref is not None:
            if ref[0].dtype != dt or ref[0].shape[1] != N or ref[0].shape[2] != heads * D or ref[1].shape[2] != heads * D:
                raise _lib.GeodiffError("_EditLayer: the reference rows handed in do not match this layer (dtype / tokens / width)")
            q_base, k_base, v_base = ref_hm if ref_hm is not None else ops.heads_split((ref[0], ref[1], ref[2]), heads)
            q_edit, k_edit, v_edit = q, k, v
            van_q, van_k, van_v = q_base, k_base, v_base
            (b0, b1), (e0, e1), cb, n_van_live = (0, 1), (0, 1), f, 0
        else:
            q_base, k_base, v_base = q[b0 * f:b1 * f], k[b0 * f:b1 * f], v[b0 * f:b1 * f]
            q_edit, k_edit, v_edit = q[e0 * f:e1 * f], k[e0 * f:e1 * f], v[e0 * f:e1 * f]
            van_q, van_k, van_v = q[:cb], k[:cb], v[:cb]
        out_full = torch.empty(cb if heads else cb + f, N, D, dtype=dt, device=dev)
        lse_van = torch.empty(cb, N, dtype=torch.float32, device=dev) if want_losses else None
        segs = [(van_q, van_k, van_v, out_full[:cb], lse_van)]
        replace_out = torch.empty(f, N, D, dtype=dt, device=dev)
        # opt-in slow path of the reference (:452-454,562-564): keep the edit row's probability map of layers with N <= 16^2
        store = ctrl.use_cfg and ctrl.store_attention_maps and (not remover) and N <= 16 ** 2
        lse_e = torch.empty(f, N, dtype=torch.float32, device=dev) if (grad_mode or want_losses or store) else None
        ident_out = None
        if not remover:
            # q_warp = q_base*(1-m) + m*splat(q_base) (:424,544), built in the attention kernel's prologue from the splat tables
            # (the fused attention-warp launch; bit-identical to the separate gd_splat_composite launch, GD_FUSED_WARP=0)
            edit_out = torch.empty(f, N, D, dtype=dt, device=dev)
            edit_act = None
            if FUSED_WARP and WARP_ROWS and (not is_cross) and D == 64 and "edit_rows" in c and k_base.shape[1] % 256 == 0:
                # only the rows inside the soft edit mask; the others are the reference rows' outputs (rows_merge below)
                edit_act = torch.empty(f, c["edit_rows"].numel(), D, dtype=dt, device=dev)
                segs.append((q_base, k_base, v_base, edit_act, None, (c["idx"], c["w"], c["m_edit"]), (c["edit_rows"], c["n_edit_rows"])))
            elif FUSED_WARP:
                segs.append((q_base, k_base, v_base, edit_out, None, (c["idx"], c["w"], c["m_edit"])))       # :427-428,548-549
            else:
                q_warp = ops.splat_composite(q_base, c["idx"], c["w"], c["m_edit"], GD_TOKEN_MAJOR)
                segs.append((q_warp, k_base, v_base, edit_out, None))
            K = k_edit if is_cross else k_base                              # :432 / :555
        else:
            edit_out = None                                                 # :786,879 vanilla reference output (below)
            K = k_base                                                      # :790,882
            if not blend:                                                   # :793-796,886-888
                ident_out = torch.empty(f, N, D, dtype=dt, device=dev)
                segs.append((q_edit, k_edit, v_edit, ident_out, None))
        segs.append((q_edit, K, v_base, replace_out, lse_e))                # :433,557 / :791,883
        ops.attn_fwd(segs, scale, q_scaled=2 if (q_pre and OPT_PRE and dt in (torch.bfloat16, torch.float16)) else 0)
        fused = FUSED_LAYER
        blend_done = False
        if (not remover) and edit_act is not None:
            if fused:      # rows outside the soft edit mask: the reference row's output; and the blend of :502-508,617-622, in the same pass
                own_blend = blend and not heads         # token-major output: the merge launch at the end blends
                ops.blend_merge(out_full[b0 * f:b1 * f], edit_act, c["edit_pos"], replace_out if own_blend else None,
                                c["m_edit"] if own_blend else None, eo_out=edit_out, out=out_full[cb:] if own_blend else None)
                blend_done = own_blend
            else:
                ops.rows_merge(out_full[b0 * f:b1 * f], edit_act, c["edit_pos"], out=edit_out)
        if remover:
            edit_out = out_full[b0 * f:b1 * f].clone() if want_losses else out_full[b0 * f:b1 * f]
        if store:
            M = K.shape[1]
            ctrl.attn_store(ops.attn_probs(q_edit, K, lse_e, None, scale)[:, :, :M].float(), is_cross=is_cross,
                            place_in_unet=ctrl.__dict__.get("_place_in_unet", "up"))

        terms = ops.zeros_f32(5, dev)                                      # sim, movement, removal, smoothness, amodal
        loss = ops.zeros_f32(1, dev).view(())
        Pe = Pb = aux = tgt = None
        coefs = rm_coef = None
        if want_losses:
            kind = "cross" if is_cross else "self"
            R = c["rows"].numel()
            use_amodal = (not remover) and N > 32 ** 2                                       # :479-480,596-597
            m_edit_l = c["m_edit"] if not remover else c["zeros"]
            # Everything that depends on the (adaptive) loss weights stays on the device: a captured hipGraph of the
            # optimisation pass then follows the schedule without re-capture, and no host->device copy sits in the layer.
            wv = ctrl.loss_weights_device(kind, dev)                 # [sim, movement, removal, smoothness, amodal]
            if fused:
                best = None
                if R > 0:
                    best = torch.empty(f, R, 2, dtype=torch.int64, device=dev)
                    # base_att (:307-317) and replace_att[:, inpaint rows] in one launch, which also clears `best`
                    Pb, Pe = ops.attn_probs_pair(q_base, k_base, lse_van[b0 * f:b1 * f], q_edit, K, lse_e, c["rows"], c.get("n_rows"), scale, zero=best)
                    ops.removal_corr_max_nz(Pe, Pb, c["m_inp"], c["m_wo"], c.get("n_rows"), best)
                if use_amodal:
                    tgt = ops.amodal_target(edit_out, c["nn_idx"], c["nn_w"], c["m_edit"], S)    # :291-293
                res = ops.edit_losses_fused(
                    edit_out, replace_out, tgt, c["m_wo"], m_edit_l, c.get("w_dist"), c.get("m_amodal"), S, best, c["rows"] if R > 0 else None,
                    c.get("n_rows"), c["inv5"], c["inv_rm"], wv, c["inv5_bwd"], use_amodal, log_acc=log_acc,
                    running=running.detach() if torch.is_tensor(running) else running)
                terms, loss, coefs, rm_coef, aux = res[:5]
                if running is not False:       # the tail added this layer's loss to the controller's running loss: that sum is what leaves
                    loss = res[5]
                    ctrl._tail_summed = True
                if aux is not None:
                    ctrl._last_removal_aux = aux          # diagnostics: arg-max indices / values of this layer
            else:
                rm = ops.zeros_f32(1, dev)
                if R > 0:
                    Pb = ops.attn_probs(q_base, k_base, lse_van[b0 * f:b1 * f], None, scale)     # base_att (:307-317)
                    Pe = ops.attn_probs(q_edit, K, lse_e, c["rows"], scale, n_valid=c.get("n_rows"))   # replace_att[:, inpaint rows]
                    aux, rm = ops.removal_fwd(Pe, Pb, c["m_inp"], c["m_wo"], c["rows"], S, n_valid=c.get("n_rows"))
                    ctrl._last_removal_aux = aux          # diagnostics: arg-max indices / values of this layer
                if use_amodal:
                    tgt = ops.amodal_target(edit_out, c["nn_idx"], c["nn_w"], c["m_edit"], S)    # :291-293
                sums = ops.edit_losses_fwd(edit_out, replace_out, tgt, c["m_wo"], m_edit_l, c.get("w_dist"), c.get("m_amodal"), S)
                # terms = [sim, movement, removal, smoothness_h + smoothness_w, amodal | 0], loss = sum terms * wv,
                # coefs = d(loss)/d(sum_i) for the backward (sim, movement, amodal, smooth_h, smooth_w), rm_coef = wv[removal] * inv_rm
                terms, loss, coefs, rm_coef = ops.loss_assemble(sums, rm, c["inv5"], c["inv_rm"], wv, c["inv5_bwd"], use_amodal)

        # output (:502-508,617-624 / :831-834,922-925)
        if heads:
            last, blend_with = replace_out, None
            if (not remover) and blend:
                last, blend_with = edit_out, (replace_out, c["m_edit"])
            elif remover and not blend:
                last, blend_with = ident_out, (replace_out, c["m_inp"])
            nb_ = n_van_live                                    # (reference rows handed in: the output is the edit row alone)
            out_full = ops.heads_merge([out_full[i * f:(i + 1) * f] for i in range(nb_)] + [last], heads, N, D, dt, dev,
                                       blend=None if blend_with is None else (nb_, blend_with[0], blend_with[1]))
        elif not remover:
            if blend:
                if not blend_done:
                    ops.blend_tokens(edit_out, replace_out, c["m_edit"], out=out_full[cb:])
            else:
                out_full[cb:].copy_(replace_out)
        else:
            if blend:
                out_full[cb:].copy_(replace_out)            # ro*m_inp + ro*m_wo == ro for complementary binary masks
            else:
                ops.blend_tokens(ident_out, replace_out, c["m_inp"], out=out_full[cb:])

        if grad_mode:
            ctx.save_for_backward(q_edit, K, v_base, replace_out, lse_e, edit_out if want_losses else None, tgt, Pe, Pb, coefs, rm_coef)
            ctx.aux, ctx.c = aux, c
            ctx.meta = dict(f=f, cb=cb, e0=e0, e1=e1, is_cross=is_cross, scale=scale, remover=remover, blend=blend,
                            want_losses=want_losses, q_shape=q.shape, k_shape=k.shape, S=S, fused=fused, heads=heads, tok_shapes=tok_shapes,
                            n_van_live=n_van_live)
        ctx.mark_non_differentiable(terms)
        return out_full, loss, terms

    @staticmethod
    def backward(ctx, g_out, g_loss, _g_terms):
        q_edit, K, v_base, replace_out, lse_e, edit_out, tgt, Pe, Pb, coefs, rm_coef = ctx.saved_tensors
        m, c = ctx.meta, ctx.c
        f, cb, S = m["f"], m["cb"], m["S"]
        dev, dt = q_edit.device, q_edit.dtype
        if m["remover"] and not m["blend"]:
            raise NotImplementedError("gradient through the remover's identity attention (cur_step >= obj_edit_step) is "
                                      "never taken by the reference driver (optimize_steps <= obj_edit_step)")
        heads = m.get("heads", 0)
        if heads and g_out is not None:           # token-major [B, N, heads*D]: the loss backward reads the edit row's slice in place
            gout = g_out[m["n_van_live"]:].contiguous()
        else:
            gout = g_out[cb:].contiguous() if g_out is not None else None
        gtok = bool(heads)
        gscale = g_loss.reshape(1).float().contiguous() if (m["want_losses"] and g_loss is not None) else None
        have_loss = m["want_losses"] and gscale is not None
        eo = edit_out if edit_out is not None else replace_out
        m_edit_l = c["m_edit"] if not m["remover"] else c["zeros"]
        # the reference rows receive no gradient (they are detached in the reference as well): dq is written straight into the edit rows of
        # the full-size gradient, the removal loss's contribution is folded into it in place (one rounding, as adding an f32 tensor would)
        e0f, e1f = m["e0"] * f, m["e1"] * f
        if heads:
            grad_q = None
            dq_view = torch.empty(f, m["q_shape"][1], m["q_shape"][2], dtype=dt, device=dev)
        else:
            grad_q = torch.empty(m["q_shape"], dtype=dt, device=dev)
            if e0f:
                grad_q[:e0f].zero_()
            if e1f < grad_q.shape[0]:
                grad_q[e1f:].zero_()
            dq_view = grad_q[e0f:e1f]
        need_dk = m["is_cross"] and not m["remover"]
        with_rm = have_loss and Pe is not None
        if m.get("fused") and with_rm:
            # [loss backward + row dots] [dq (+ dk)] [removal dS K] [one fold of both sets of partials]
            rm_args, rm_ws = ops.removal_bwd_args(Pe, Pb, q_edit, K, c["rows"], ctx.aux, c["m_inp"], c["m_wo"], 1.0, gscale, rm_coef, m["scale"],
                                                  c.get("n_rows"), need_dk)
            dro = ops.edit_losses_bwd_rowdot(eo, replace_out, tgt, c["m_wo"], m_edit_l, c.get("w_dist"), c.get("m_amodal"), gout, coefs, gscale,
                                             blend=(m["blend"] and not m["remover"]), S=S, rm=rm_args, gout_tok=gtok)
            dk32, kchunks, part_ptr, bwd_ws = ops.attn_bwd_nofold(q_edit, K, v_base, replace_out, lse_e, dro, m["scale"], need_dk, dq_view)
            rm_args.dk_f32 = dk32.data_ptr() if dk32 is not None else None
            ops.removal_bwd_nofold(rm_args, dt)
            ops.edit_dq_fold(part_ptr if kchunks > 1 else None, kchunks, f, dq_view.shape[1], dq_view.shape[2], rm_ws, K.shape[1], c["rows"].numel(),
                             c["inp_pos"], ctx.aux["wgt"], dq_view)
            del bwd_ws
        else:
            dro = ops.edit_losses_bwd(eo, replace_out, tgt if have_loss else None, c["m_wo"], m_edit_l, c.get("w_dist"),
                                      c.get("m_amodal"), gout, coefs if have_loss else _zeros5(dev), gscale,
                                      blend=(m["blend"] and not m["remover"]), S=S, gout_tok=gtok)
            _, dk32 = ops.attn_bwd(q_edit, K, v_base, replace_out, lse_e, dro, m["scale"], need_dk=need_dk, dq_out=dq_view)
            if with_rm:
                ops.removal_bwd(Pe, Pb, q_edit, K, c["rows"], ctx.aux, c["m_inp"], c["m_wo"], 1.0, gscale * rm_coef, m["scale"], None, dk32,
                                n_valid=c.get("n_rows"), dq16=dq_view)
        grad_k = None
        if heads:          # token-major gradients, zero rows included, one launch each (dk: f32 -> 16-bit in the same pass)
            (Bq, Nq, _), (Bk, Mk, _) = m["tok_shapes"]
            D = m["q_shape"][2]
            grad_q = ops.heads_merge([dq_view if b == m["e0"] else None for b in range(Bq)], heads, Nq, D, dt, dev)
            if dk32 is not None:
                grad_k = ops.heads_merge([dk32 if b == m["e0"] else None for b in range(Bk)], heads, Mk, D, dt, dev)
        elif dk32 is not None:
            grad_k = torch.zeros(m["k_shape"], dtype=dt, device=dev)
            grad_k[m["e0"] * f:m["e1"] * f] = dk32.to(dt)
        g_run = g_loss if (len(ctx.needs_input_grad) > 9 and ctx.needs_input_grad[9]) else None      # d(running + loss) / d running = 1
        return grad_q, grad_k, None, None, None, None, None, None, None, g_run, None, None


# ---------------------------------------------------------------------------------------------------------
# controllers
# ---------------------------------------------------------------------------------------------------------
class _GeometryControllerBase(AttentionStore, abc.ABC):
    _is_remover = False

    def step_callback(self, x_t, transform_coords=None):
        if self.local_blend is not None:
            x_t = self.local_blend(x_t, self.attention_store, transform_coords)
        return x_t

    def initialize_default_loss_weights(self):
        # aliases rather than copies, as the reference does (:667-668): the adaptive schedule's in-place updates
        # therefore also change the "defaults"
        self.loss_weight_dict = self.default_loss_weights

    def loss_weights_device(self, kind: str, dev) -> torch.Tensor:
        """[sim, movement, removal, smoothness, amodal] of ``loss_weight_dict[kind]`` as a device vector with a stable
        address; re-uploaded only when the host values changed (the adaptive schedule edits the dict in place)."""
        lw = self.loss_weight_dict[kind]
        rem = self._is_remover
        host = (float(lw["sim"]), 0.0 if rem else float(lw.get("movement", 0.0)), float(lw["removal"]),
                float(lw["smoothness"]), 0.0 if rem else float(lw.get("amodal", 0.0)))
        # one buffer per (kind, controller type, device) for the whole process: captured optimisation passes of earlier edits read it
        key = (kind, rem, str(dev), getattr(self, "slot", 0))
        ent = _WEIGHT_VECTORS.get(key)
        if ent is None:
            ent = _WEIGHT_VECTORS[key] = [None, torch.zeros(5, dtype=torch.float32, device=dev)]
        if ent[0] != host:
            # pinned staging + non-blocking copy: a pageable source makes the copy wait for everything queued before it (the whole CFG
            # pass in front of an optimisation pass: 7 ms of host stall per pass, then an idle device while the host catches up)
            if len(ent) < 3:
                ent.append([torch.empty(5, dtype=torch.float32).pin_memory() for _ in range(4)])
                ent.append(0)
            stage = ent[2][ent[3] % 4]
            ent[3] += 1
            stage.copy_(torch.tensor(host, dtype=torch.float32))
            ent[1].copy_(stage, non_blocking=True)
            ent[0] = host
        return ent[1]

    def sync_loss_weights(self, dev):
        """Upload both weight vectors now (called before replaying a captured optimisation pass)."""
        for kind in ("self", "cross"):
            self.loss_weights_device(kind, dev)

    def _common_init(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, controller,
                     empty_scale, use_all, obj_edit_step, mode):
        self.mode = mode
        self.prev_controller = controller
        self.last_cross_mask = None
        self.thre = 0.00001
        self.empty_scale = empty_scale
        self.use_all = use_all
        self.loss = 0.0
        self.batch_size = len(prompts)
        # get_time_words_attention_alpha (ptp_utils.py:110-128): only its shape/indexing is consumed
        alpha = torch.zeros(num_steps + 1, len(prompts) - 1, 1, 1, 77)
        bounds = cross_replace_steps["default_"] if isinstance(cross_replace_steps, dict) else cross_replace_steps
        if isinstance(bounds, float):
            bounds = (0.0, bounds)
        alpha[int(bounds[0] * (num_steps + 1)):int(bounds[1] * (num_steps + 1))] = 1
        self.cross_replace_alpha = alpha
        if type(self_replace_steps) is float:
            self_replace_steps = 0, self_replace_steps
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.local_blend = local_blend
        self.mask_inpaint = None
        self.obj_edit_step = obj_edit_step
        self.num_steps = num_steps
        self.mask_new_warped = None
        self.mask_wo_edit = None
        self.mask_1_empty = None
        self.amodal_mask = None
        self.coords_base = (2, 3)
        self.coords_edit = (3, 4)
        self.use_cfg = True
        self.loss_log_dict = None
        self.loss_weight_dict = None
        self.store_attention_maps = False
        self.masks_cache_dict: Dict[int, dict] = {}
        # dtype the resampled warp coordinates are rounded through (".type_as(q)" in the reference, whose q is fp16)
        self.coords_dtype = torch.float16

    # -- per-resolution device tables --------------------------------------------------------------------
    def _tables(self, S: int, f: int, q: torch.Tensor, transform_coords, D: int = 64):
        """D: the TRUE head dim (loss normalisers, U/attention_processors.py:231-305); only q's device / dtype are read."""
        c = self.masks_cache_dict.get(S)
        if c is not None and "f" in c:
            if c["D"] != D:
                raise ValueError(f"layers of latent size {S} disagree on the head dim ({c['D']} vs {D})")
            if getattr(self, "persistent_tables", False):
                _check_table_owner(self)
            return c
        if getattr(self, "persistent_tables", False):
            import weakref
            if not any("f" in t for t in self.masks_cache_dict.values()):      # first table of this controller: it takes the buffers over
                _TABLE_OWNER[getattr(self, "slot", 0)] = weakref.ref(self)
            else:
                _check_table_owner(self)
        dev = q.device
        N = S * S
        q_base_like = torch.empty(1, f, N, 1, device=dev, dtype=q.dtype)
        q_img_like = torch.empty(f, 1, S, S, device=dev, dtype=q.dtype)
        if not self._is_remover:
            if self.mask_new_warped is None:                                   # :390-396,520-526
                t_m = reshape_transform_coords(transform_coords.to(dev).float(), in_mat_shape=self.image_mask.shape)
                t_m = t_m.tile(self.image_mask.shape[0], 1, 1, 1).to(self.coords_dtype)
                self.mask_new_warped = binarize_tensor(
                    warp_utils.warp_grid_edit(self.image_mask[:, None].to(dev).float(), t_m)).detach()
            res = process_and_cache_masks(self.masks_cache_dict, S, self.image_mask, self.mask_new_warped.detach(),
                                          self.amodal_mask, transform_coords, q_base_like, q_img_like,
                                          coords_dtype=self.coords_dtype)
            self.masks_cache_dict, self.image_mask = res[0], res[1]
            m_new, mask_warp, amodal, inter, m_empty, m_wo, t_q = res[2:]
            c = self.masks_cache_dict[S]
            pt = self if getattr(self, "persistent_tables", False) else False
            c["m_edit"] = _persist(pt, ("m_edit", S), _flat(m_new))
            c["m_amodal"] = _flat(amodal)
            idx_, w_ = warp_utils.SPLATTER.tables(t_q[0].reshape(-1, 3))
            c["idx"], c["w"] = _persist(pt, ("idx", S), idx_), _persist(pt, ("w", S), w_)
            if N >= WARP_ROWS_MIN_TOKENS:
                # rows whose warped query differs from the reference query (m_edit > 0), as a list padded to a bucketed length (launch
                # dimensions repeat from edit to edit: hipGraph reuse), its length on the device, and the inverse map row -> list slot
                er = torch.nonzero(c["m_edit"] > 0).reshape(-1).to(torch.int32)
                bucket = N // 16
                R_e = int(er.numel())                                      # (host sync: once per resolution per edit, like the mask sums)
                R_pad = max(bucket, -(-R_e // bucket) * bucket)
                pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
                pos[er.long()] = torch.arange(R_e, dtype=torch.int32, device=dev)
                er = torch.cat([er, torch.zeros(R_pad - R_e, dtype=torch.int32, device=dev)]).contiguous()
                c["edit_rows"] = _persist(pt, ("edit_rows", S, R_pad), er)
                c["n_edit_rows"] = _persist(pt, ("n_edit_rows", S), torch.tensor([R_e], dtype=torch.int32, device=dev))
                c["edit_pos"] = _persist(pt, ("edit_pos", S), pos)
        else:
            self.image_mask = self.image_mask.to(dev).float().detach()         # :758,852
            mask_warp = binarize_tensor(self.image_mask)[:, None]
            mask_warp = reshape_attention_mask(mask_warp, in_mat_shape=(1, S))[1:]
            m_empty = binarize_tensor(mask_warp, 0.5)                          # :765,858
            m_wo = binarize_tensor(torch.ones_like(mask_warp) - m_empty)       # :775,867
            c = self.masks_cache_dict.setdefault(S, {})
            c["mask_1_empty"], c["mask_wo_edit"] = m_empty, m_wo
        pt = self if getattr(self, "persistent_tables", False) else False
        rem = self._is_remover
        c["m_inp"] = _persist(pt, ("m_inp", S, rem), _flat(m_empty))
        c["m_wo"] = _persist(pt, ("m_wo", S, rem), _flat(m_wo))
        c["zeros"] = _persist(pt, ("zeros", S), torch.zeros(N, dtype=torch.float32, device=dev))
        rows = torch.nonzero(c["m_inp"] > 0.5).reshape(-1).to(torch.int32).contiguous()
        sums = [c["m_wo"].sum(), c["m_inp"].sum()]
        if not rem:
            c["m_amodal"] = _persist(pt, ("m_amodal", S), c["m_amodal"])
            sums.append(c["m_edit"].sum())
            if N > 32 ** 2:
                nn_idx, nn_w, w_dist = ops.nn_table(c["m_edit"], S)
                c["nn_idx"], c["nn_w"], c["w_dist"] = (_persist(pt, ("nn_idx", S), nn_idx), _persist(pt, ("nn_w", S), nn_w),
                                                       _persist(pt, ("w_dist", S), w_dist))
                sums.append((c["w_dist"] * c["m_amodal"]).sum())
        host = torch.stack(sums).tolist()                                      # one sync per resolution per edit
        c["s_wo"], c["s_inp"] = host[0], host[1]
        c["s_edit"] = host[2] if len(host) > 2 else 0.0
        c["s_am"] = host[3] if len(host) > 3 else 0.0
        # inpaint-row list.  With persistent tables (hipGraph reuse across edits) its length is rounded up to a bucket so that the
        # launch dimensions of the loss kernels repeat from edit to edit; the padding slots carry weight 0 (gd_removal_loss_reduce)
        R = rows.numel()
        c["n_rows"] = None
        if pt and N >= 32 ** 2:
            bucket = max(64, N // 16)
            R_pad = max(bucket, -(-R // bucket) * bucket)
            pad = rows[:1].expand(R_pad - R) if R else torch.zeros(R_pad, dtype=torch.int32, device=dev)
            rows = _persist(pt, ("rows", S, rem, R_pad), torch.cat([rows, pad]).contiguous())
            c["n_rows"] = _persist(pt, ("n_rows", S, rem), torch.tensor([R], dtype=torch.int32, device=dev))
        c["rows"] = rows
        # inverse of the inpaint-row list (row -> slot, -1 elsewhere; live slots only): what the merged dq fold looks rows up in
        inp_pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
        if R:
            inp_pos[rows[:R].long()] = torch.arange(R, dtype=torch.int32, device=dev)
        c["inp_pos"] = _persist(pt, ("inp_pos", S, rem), inp_pos)
        # reciprocals of the loss denominators (U/attention_processors.py:231-305) on the device: sim, movement, amodal, smooth_h,
        # smooth_w, removal
        use_amodal = (not rem) and N > 32 ** 2
        cnt = float(f * S * (S - 1) * D)
        inv = torch.tensor([1.0 / (f * D * c["s_wo"] + 1e-8), 1.0 / (f * D * c["s_edit"] + 1e-8), 1.0 / (f * D * c["s_am"] + 1e-8),
                            1.0 / cnt, 1.0 / cnt, 1.0 / (c["s_inp"] * f + 1e-8)], dtype=torch.float32, device=dev)
        c["inv5"], c["inv_rm"] = _persist(pt, ("inv5", S, rem), inv[:5].contiguous()), _persist(pt, ("inv_rm", S, rem), inv[5:6].contiguous())
        tb = inv[:5].clone()
        if not use_amodal:
            tb[2] = 0.0
        c["inv5_bwd"] = _persist(pt, ("inv5_bwd", S, rem), tb)
        c["S"], c["f"], c["D"] = S, f, D
        if N >= 32 ** 2:                                                       # :413-415,575-576 / :778-780
            self.mask_wo_edit = m_wo.detach()
            self.mask_1_empty = m_empty.detach()
            self.mask_inpaint = m_empty[0, 0].detach().clone()
        return c

    def table_signature(self):
        """What a captured pass bakes in besides graph_key(): per resolution the (padded) inpaint-row count, the head count and the
        slot count K of the splat tables — i.e. every launch dimension / buffer shape that comes from the per-resolution tables
        (``_persist`` keys its buffers on shape: a different K lives in different buffers, so it must be a different graph)."""
        return tuple(sorted((S, c["rows"].numel() if S * S >= 32 ** 2 else 0, c["f"],            # losses exist only at N >= 32^2
                             int(c["idx"].shape[-1]) if "idx" in c else 0,
                             int(c["edit_rows"].numel()) if "edit_rows" in c else 0)
                            for S, c in self.masks_cache_dict.items() if "f" in c))

    def tables_built(self, layers) -> bool:
        return all(S in self.masks_cache_dict and "f" in self.masks_cache_dict[S] for S, *_ in layers)

    def prebuild_tables(self, layers, q_like: torch.Tensor, transform_coords):
        """Build the per-resolution tables for the (S, heads) pairs of a previous pass now (one host sync each) instead of lazily
        inside the first hooked call, so that a captured pass can be replayed as the very first pass of an edit."""
        for S, f, *rest in layers:
            self._tables(S, f, q_like, transform_coords, rest[0] if rest else 64)

    def graph_key(self):
        """Everything a no-grad UNet pass of this controller branches on (the launch sequence is static for a given key)."""
        active = self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]
        blend = self.cur_step < int(self.num_steps * self.obj_edit_step)
        return (type(self).__name__, active, blend, getattr(self, "n_batch", None), self.coords_base, self.coords_edit,
                self.use_cfg, self.store_attention_maps, bool(self.rows_identical), self.ref_stash_serial if self.use_ref_stash else None)

    def after_graph_replay(self):
        """A replayed pass ran no Python: advance the counters as its num_att_layers hooked calls would have."""
        self.cur_att_layer = 0
        self.cur_step += 1
        self.between_steps()

    # REFERENCE ROWS OF THE CFG PASS FROM THE OPTIMISATION PASS (editor.REF_FROM_OPT).  At a step with an optimisation pass the reference
    # sample goes through the UNet twice with the same latent, timestep and text rows (the update touches the edit row only, and the
    # reference row's text gradient is zero: its tensors are detached in the hooked layers): once beside the edit row in the
    # optimisation pass, once as `cond_ref` in the CFG pass, where only its per-layer q / k / v and attention output are used.  The
    # optimisation pass leaves those four tensors per hooked layer (ref_stash: token-major [1, N | M, C] views of its own tensors, in
    # call order; None for a layer that ran plain attention) and the CFG pass of the same step runs [uncond_edit, cond_edit] only.
    # Inside hipGraphs the tensors are static addresses of the optimisation pass's graph: ref_stash_serial names that graph and is
    # part of the CFG pass's graph key.
    collect_ref = False          # set by the driver around an optimisation pass
    use_ref_stash = False        # set by the driver around the CFG pass that takes its reference rows from ref_stash
    ref_stash = None
    ref_stash_serial = None
    ref_stash_t = None
    _ref_pos = 0

    def _leave_ref(self, entry):
        if self.collect_ref:
            if self.cur_att_layer == 0 or self.ref_stash is None:
                self.ref_stash = []
            self.ref_stash.append(entry)

    def _take_ref(self):
        entry = self.ref_stash[self._ref_pos]
        self._ref_pos += 1
        return entry

    # THE REFERENCE ROW OF AN OPTIMISATION STEP ONE STEP AHEAD (editor.REF_AHEAD).  The reference sample of step i+1 is known while step i
    # runs (the inversion trajectory's entry for t_{i+1}, U/editor.py:375-377) and no row of a UNet batch depends on another outside the
    # hooked layers.  The CFG pass of step i therefore carries it as one more vanilla row (row 0, its own timestep) and keeps, per hooked
    # call, that row's token-major (q, k, v, attention output) (collect_ahead -> ref_stash, same entry format as above; `out` is row 0
    # of the pass's vanilla segment).  The optimisation pass of step i+1 then runs forward + backward on the EDIT ROW ALONE with the
    # reference q / k / v handed to _EditLayer (use_ahead), and that step's CFG pass reads the same entries (use_ref_stash).
    # ALL REFERENCE ROWS OF AN EDIT IN ONE PASS (editor.REF_AHEAD = 1, the default): every optimisation step's reference sample is known once the
    # inversion is done, so one no-grad pass runs them all as a batch of vanilla rows with their own timesteps right behind the inversion
    # (collect_ahead = "all": every row of every hooked call is kept, plain attention for all of them).  Before optimisation step m, row m of
    # every kept tensor is copied (one launch: ops.RowCopyTable) into persistent one-row tensors (ref_slots) — the addresses both the
    # optimisation pass on the edit row alone and that step's 2-row CFG pass read, the same for every step and every edit: no captured pass
    # depends on which pass produced the rows.  Against the carrying form above: no 4-row CFG passes (the UNet's GEMMs pick worse tiles at
    # M = 16,384 than at 12,288) and the first optimisation step runs on the edit row alone as well.
    collect_ahead = False        # set by the driver: True around the CFG pass that carries the next step's reference row, "all" around the batched pass
    use_ahead = False            # set by the driver around the optimisation pass that takes its reference rows from ref_stash
    _ahead = None

    def _leave_ahead(self, q, k, v, out):
        if self.collect_ahead:
            if self.cur_att_layer == 0 or self._ahead is None:
                self._ahead = []
            if self.collect_ahead == "all":
                # (row copies address the kept tensors in place, replay after replay: a non-contiguous one voids the set -> the drivers fall back)
                ok = all(t.is_contiguous() for t in (q, k, v, out))
                self._ahead.append((q.detach(), k.detach(), v.detach(), out.detach()) if ok else False)
            else:
                self._ahead.append((q[0:1].detach(), k[0:1].detach(), v[0:1].detach(), out[0:1].detach()))

    def ref_slots(self, stash):
        """Persistent one-row tensors for the reference row of the current optimisation step (one set per device / dtype / layer-shape list,
        shared by all edits of the process: captured passes read them by address), the table that copies row m of ``stash`` into them, and
        the serial number that names them in graph keys."""
        sig = (str(stash[0][0].device), stash[0][0].dtype, tuple(tuple(t.shape[1:]) for e in stash for t in e))
        ent = _REF_SLOTS.get(sig)
        if ent is None:
            from . import graphs
            slots = [tuple(torch.empty(1, *t.shape[1:], dtype=t.dtype, device=t.device) for t in e) for e in stash]
            ent = _REF_SLOTS[sig] = dict(slots=slots, serial=("slots", next(graphs._SERIAL)), table=None, stash_id=None)
        sid = tuple(t.data_ptr() for e in stash for t in e)
        if ent["stash_id"] != sid:                               # (a captured batched pass keeps its addresses: built once per process)
            ent["table"] = ops.RowCopyTable([(t, d) for e, se in zip(stash, ent["slots"]) for t, d in zip(e, se)])
            ent["stash_id"] = sid
        return ent

    supports_token_major = True
    heads_tok = 0
    heads_opt = 0                # > 0: forward() was handed token-major q / k / v by a pass that accumulates losses (TOK_OPT)
    q_scaled_tok = False
    q_scaled_hm = False
    supports_scaled_q_head_major = True
    # Set by the driver for an optimisation pass whose reference and edit samples are the SAME sample in every layer (the first pass of
    # a removal edit: both rows start from x_T with the same text, and the remover's output for the edit row equals the vanilla
    # output, so the rows never diverge).  In exact arithmetic their q / k / v are then equal, the attention outputs
    # compared by the L1 losses are equal, and d|x|/dx = sign(0) = 0 (the reference's CPU run gives exactly that).  On the device the
    # two rows of a batch drift apart by the convolution library's run-to-run rounding (~1e-5 per layer, MIOpen split-K kernels), the
    # L1 terms see +-noise and their sign() gradient becomes full-magnitude noise (measured: the remover's first latent update 34 %
    # off the reference, 2 % with ideal 16-bit storage).  _tie_rows restores the symmetry where it matters.
    rows_identical = False

    def _tie_rows(self, t, f):
        """The edit rows take the VALUES of the base rows and keep their own autograd history: base.detach() + (edit - edit.detach())."""
        (b0, b1), (e0, e1) = self.coords_base, self.coords_edit
        te = t[e0 * f:e1 * f]
        tied = t[b0 * f:b1 * f].detach() + (te - te.detach())
        return torch.cat([t[:e0 * f], tied, t[e1 * f:]], 0)

    def _forward_tok(self, q, k, v, is_cross: bool, transform_coords, scale: float, heads: int, ref=None):
        """No-grad CFG pass on token-major q/k/v [B, N, heads*64]; the caller (EditProcessor) routes passes that accumulate
        losses (use_cfg False) through the head-major _EditLayer instead.  Same launches as _EditLayer.forward with batch
        rows in place of head blocks; the edit attention, whose output only feeds the blend, is skipped when not blending."""
        (b0, b1), (e0, e1) = self.coords_base, self.coords_edit
        cb = self.coords_base[-1] if ref is None else e0         # vanilla roles ahead of the edit row
        S = int(math.isqrt(q.shape[1]))
        c = self._tables(S, heads, q, transform_coords)
        remover = self._is_remover
        blend = self.cur_step < int(self.num_steps * self.obj_edit_step)
        out_full = torch.empty(cb + 1, q.shape[1], q.shape[2], dtype=q.dtype, device=q.device)
        if ref is None:
            q_base, k_base, v_base, van_base = q[b0:b1], k[b0:b1], v[b0:b1], out_full[b0:b1]
        else:                                                    # the reference row as the optimisation pass of this step left it
            if not self.q_scaled_tok or ref[0].shape[1:] != q.shape[1:] or ref[0].dtype != q.dtype:
                raise RuntimeError("ref_stash does not match this pass (query scaling / shape / dtype): disable GD_REF_FROM_OPT")
            q_base, k_base, v_base, van_base = ref
        q_edit, k_edit, v_edit = q[e0:e1], k[e0:e1], v[e0:e1]
        v0 = 0
        if self.collect_ahead and k.shape[1] >= 4096 and cb > 1:
            # the carried row (row 0, REF_AHEAD) as its own launch at 64^2 tokens and up: with it the launch below has 25 heads = 325-335
            # units of 256 queries for 256 CUs — 120 us as one launch (two rounds, or the even split's three segments per workgroup)
            # against 66 + 33 us for the one-round 20-head launch and the 5-head launch in unit parts (tools/attn_one.py FORM=cfg4n[_split])
            ops.attn_fwd([(q[:1], k[:1], v[:1], out_full[:1], None)], scale, heads=heads, q_scaled=self.q_scaled_tok)
            v0 = 1
        segs = [(q[v0:cb], k[v0:cb], v[v0:cb], out_full[v0:cb], None)]
        replace_out = out_full[cb:]
        edit_out = ident_out = edit_act = None
        if PAIR_BLEND and FUSED_WARP and k.shape[1] <= 128 and q.shape[2] == heads * 64 and ((not remover and blend) or (remover and not blend)):
            # short key lists: both attention outputs of the edit rows and their blend in one launch
            if not remover:
                side_a = (q_base, k_base, v_base, out_full[cb:], None, (c["idx"], c["w"], c["m_edit"]))    # edit_out (:427-428,548-549)
                side_b, m_blend = (q_edit, k_edit if is_cross else k_base, v_base), c["m_edit"]           # replace_out (:433,557)
            else:
                side_a = (q_edit, k_edit, v_edit, out_full[cb:], None)                                     # identity attention (:793-796)
                side_b, m_blend = (q_edit, k_base, v_base), c["m_inp"]                                     # replace_out (:791,883)
            ops.attn_fwd_pair(segs + [side_a], side_b, m_blend, scale, heads=heads, q_scaled=self.q_scaled_tok)
            return out_full
        if not remover:
            K = k_edit if is_cross else k_base
            if blend:
                edit_out = torch.empty_like(q_edit)
                replace_out = torch.empty_like(q_edit)
                if FUSED_WARP and WARP_ROWS and (not is_cross) and "edit_rows" in c and k_base.shape[1] % 256 == 0:
                    edit_act = torch.empty(1, c["edit_rows"].numel(), q.shape[2], dtype=q.dtype, device=q.device)
                    segs.append((q_base, k_base, v_base, edit_act, None, (c["idx"], c["w"], c["m_edit"]), (c["edit_rows"], c["n_edit_rows"])))
                elif FUSED_WARP:                                # warped queries built in the attention kernel's prologue
                    segs.append((q_base, k_base, v_base, edit_out, None, (c["idx"], c["w"], c["m_edit"])))
                else:
                    q_warp = ops.splat_composite(q_base, c["idx"], c["w"], c["m_edit"], GD_TOKEN_MAJOR)   # all heads in one row
                    segs.append((q_warp, k_base, v_base, edit_out, None))
        else:
            K = k_base
            if not blend:
                ident_out = torch.empty_like(q_edit)
                replace_out = torch.empty_like(q_edit)
                segs.append((q_edit, k_edit, v_edit, ident_out, None))
        segs.append((q_edit, K, v_base, replace_out, None))
        ops.attn_fwd(segs, scale, heads=heads, q_scaled=self.q_scaled_tok)
        if edit_act is not None and FUSED_LAYER:                # rows outside the soft edit mask: the reference row's output — merged and
            ops.blend_merge(van_base, edit_act, c["edit_pos"], replace_out, c["m_edit"], eo_out=None, out=out_full[cb:])   # blended in one pass
            return out_full
        if edit_act is not None:
            ops.rows_merge(van_base, edit_act, c["edit_pos"], out=edit_out)
        if edit_out is not None:
            ops.blend_tokens(edit_out, replace_out, c["m_edit"], out=out_full[cb:])
        elif ident_out is not None:
            ops.blend_tokens(ident_out, replace_out, c["m_inp"], out=out_full[cb:])
        return out_full

    def forward(self, q, k, v, is_cross: bool, place_in_unet: str, transform_coords=None, scale=None, mask=None):
        nb = getattr(self, "n_batch", None) or (2 * self.batch_size if self.use_cfg else self.batch_size)
        active = is_cross or (self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1])
        heads = self.heads_tok
        if heads:
            if self.collect_ahead == "all":      # the batched reference pass: vanilla rows only, every row kept
                out = attention_tok(q, k, v, scale, heads, q_scaled=self.q_scaled_tok)
                self._leave_ahead(q, k, v, out)
                return out
            ref = self._take_ref() if self.use_ref_stash else None
            if not active:
                out = attention_tok(q, k, v, scale, heads, q_scaled=self.q_scaled_tok)
                self._leave_ahead(q, k, v, out)
                return out
            if is_cross:
                _ = self.cross_replace_alpha[self.cur_step]
            out = self._forward_tok(q, k, v, is_cross, transform_coords, float(scale), heads, ref=ref)
            self._leave_ahead(q, k, v, out)
            return out
        if self.collect_ahead == "all":      # the batched reference pass on a layer off the token-major path (head dims 40 / 80 / 160): plain
            self._ahead = None               # attention, nothing kept — the driver sees an incomplete set and falls back
            return attention(q, k, v, scale)
        ho = self.heads_opt              # token-major q / k / v [B, N, heads*64] (EditProcessor, TOK_OPT)
        ref_in = None
        if self.use_ahead:               # the live batch is the edit row alone; this layer's reference row was left by the previous CFG pass
            ref_in = self._take_ref()
            if not ho or not self.q_scaled_hm or self.rows_identical or not isinstance(ref_in, tuple):
                raise _lib.GeodiffError("use_ahead needs the token-major optimisation-pass layers with pre-scaled queries (GD_TOK_OPT, "
                                        "GD_SCALED_Q_OPT) and a complete ref_stash: disable GD_REF_AHEAD")
        f = ho if ho else q.shape[0] // nb
        self._place_in_unet = place_in_unet
        q_pre = bool(self.q_scaled_hm)
        if q_pre:                        # q carries scale*log2(e) (_project_qkv): s = ln2 * q'.k for everything that takes a scale
            scale = LN2
        if not active:
            self._leave_ref(None)                                              # (plain attention in the CFG pass of this step too)
            if ho:
                B, N, C = q.shape
                hm = lambda t: t.view(B, t.shape[1], ho, C // ho).permute(0, 2, 1, 3).reshape(B * ho, t.shape[1], C // ho)
                return attention(hm(q), hm(k), hm(v), scale).view(B, ho, N, C // ho).permute(0, 2, 1, 3).reshape(B, N, C)
            return attention(q, k, v, scale)                                   # :646-647
        if is_cross:
            _ = self.cross_replace_alpha[self.cur_step]                        # :654 (indexing only; value unused)
        S = int(math.isqrt(q.shape[1]))
        D = 64 if ho else q.shape[2]
        c = self._tables(S, f, q, transform_coords, D)
        if self.rows_identical and torch.is_grad_enabled():
            q, k, v = (self._tie_rows(t, 1 if ho else f) for t in (q, k, v))       # token-major: one batch row per latent
        if D % 64:      # SD1.x heads (40 / 80 / 160): zero columns up to the kernels' 64 / 128 / 192; the loss normalisers keep the true D
            q, k, v = pad_head_dim(q), pad_head_dim(k), pad_head_dim(v)
        lossy = (q.shape[1] >= 32 ** 2) and (not self.use_cfg)
        kind = "cross" if is_cross else "self"
        running, log_acc = False, None
        if lossy and TAIL_SUMS and FUSED_LAYER and q.is_cuda:
            # the running loss and the logged sums are kept by the loss launch's tail (TAIL_SUMS).  The log of this kind is either fresh
            # (all 0.0: first lossy layer of the pass -> its entries become views of a zeroed device vector) or already those views.
            log = self.loss_log_dict[kind]
            acc = self.__dict__.get("_log_acc_" + kind)
            if all((not torch.is_tensor(x)) and x == 0.0 for x in log.values()):
                acc = self.__dict__["_log_acc_" + kind] = ops.zeros_f32(4, q.device)
                for i, key in enumerate(LOG_KEYS):
                    if key in log:
                        log[key] = acc[i]
            views = acc is not None and all(torch.is_tensor(log[key]) and log[key].data_ptr() == acc[i].data_ptr()
                                            for i, key in enumerate(LOG_KEYS) if key in log)
            lo = self.loss
            if torch.is_tensor(lo):
                ok_loss = lo.is_cuda and lo.dtype == torch.float32 and lo.numel() == 1
            else:
                ok_loss, lo = (lo == 0.0), None
            if views and ok_loss:
                running, log_acc = lo, acc
        self._tail_summed = False
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, loss, terms = _EditLayer.apply(q, k, v, self, is_cross, float(scale), c, q_pre, ho, running, log_acc, ref_in)
        if self.collect_ref:
            (b0, b1) = self.coords_base
            ok = ho and q_pre and D == 64 and not self.rows_identical and q.dtype in (torch.float16, torch.bfloat16)
            # (token-major 16-bit rows with pre-scaled queries: what _forward_tok takes; anything else ends the collection for this pass)
            self._leave_ref((q[b0:b1].detach(), k[b0:b1].detach(), v[b0:b1].detach(), out[b0:b1].detach()) if ok else False)
        if not ho:
            out = out[..., :D]
        if lossy:
            if self._tail_summed:
                self.loss = loss                                               # already self.loss + this layer's loss
            else:
                self.loss = self.loss + loss                                   # :494,604 / :822,914
                log = self.loss_log_dict[kind]
                named = {"sim": terms[0], "movement": terms[1], "removal": terms[2], "smoothness": terms[3]}
                for key in log:                                                # generic.py:34-39
                    log[key] = log[key] + named[key]
            self.loss_log_dict["num_layers"] += 1
        return out


class AttentionGeometryEdit(_GeometryControllerBase):
    """attention_processors.py:377-736."""

    def initialize_loss_log_dict(self):
        self.loss_log_dict = {"self": {"sim": 0.0, "movement": 0.0, "removal": 0.0, "smoothness": 0.0},
                              "cross": {"sim": 0.0, "movement": 0.0, "removal": 0.0, "smoothness": 0.0},
                              "num_layers": 0}

    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, equalizer=None, local_blend=None,
                 controller=None, image_mask=None, empty_scale=0.2, use_all=True, obj_edit_step=0.0, tokenizer=None,
                 device="cuda:0", mode="bilinear"):
        super().__init__()
        self._common_init(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, controller,
                          empty_scale, use_all, obj_edit_step, mode)
        if image_mask is not None:
            image_mask = torch.from_numpy(np.asarray(image_mask)[None])
            self.image_mask = image_mask.tile((len(prompts), 1, 1))
        self.default_loss_weights = {"self": {"sim": 110, "movement": 13.5, "removal": 1.67, "smoothness": 35.0, "amodal": 80.5},
                                     "cross": {"sim": 60, "movement": 6.34, "removal": 1.6, "smoothness": 20.0, "amodal": 3.5}}
        self.initialize_loss_log_dict()
        self.initialize_default_loss_weights()


class AttentionGeometryRemover(_GeometryControllerBase):
    """attention_processors.py:741-1023."""

    _is_remover = True

    def initialize_loss_log_dict(self):
        self.loss_log_dict = {"self": {"sim": 0.0, "removal": 0.0, "smoothness": 0.0},
                              "cross": {"sim": 0.0, "removal": 0.0, "smoothness": 0.0},
                              "num_layers": 0}

    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, equalizer=None, local_blend=None,
                 controller=None, image_mask=None, empty_scale=0.2, use_all=True, obj_edit_step=0.0, tokenizer=None,
                 device="cuda:0", mode="bilinear"):
        super().__init__()
        self._common_init(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, controller,
                          empty_scale, use_all, obj_edit_step, mode)
        if image_mask is not None:
            image_mask = torch.from_numpy(np.asarray(image_mask)[None]).tile((len(prompts), 1, 1))
            self.image_mask = torch_dilate(image_mask[:, None].float(), 5)[:, 0]               # :986
        self.default_loss_weights = {"self": {"sim": 110.0, "removal": 3.6, "smoothness": 35.0},
                                     "cross": {"sim": 60.0, "removal": 3.6, "smoothness": 20.0}}
        self.initialize_default_loss_weights()
        self.initialize_loss_log_dict()
