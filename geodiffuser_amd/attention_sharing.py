"""AttentionControl / AttentionStore base classes and attention helpers.

Mirror of the reference's GeoDiffuser/utils/attention_sharing.py (same class / function names and counter
semantics); the arithmetic runs in the HIP flash-attention kernels, so no [B*f, N, M] map is
materialised unless a caller explicitly asks for one.
"""
from __future__ import annotations

import os

import abc
from typing import Optional

import torch

from . import ops

LOW_RESOURCE = False


class _VanillaAttention(torch.autograd.Function):
    """out = softmax(scale q k^T) v through gd_attn_fwd; backward through gd_attn_bwd (dQ) and gd_attn_bwd_dkv (dK, dV).

    On the GeoDiffuser edit path only dQ is ever asked for (every k / v that would receive gradient is detached in the reference:
    attention_sharing.py:242, attention_processors.py:433,555-557); null-text optimisation (inversion.py:213-259) differentiates the
    UNet w.r.t. its text context and needs all three."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        out = torch.empty_like(q)
        lse = torch.empty(q.shape[0], q.shape[1], dtype=torch.float32, device=q.device)
        ops.attn_fwd([(q, k, v, out, lse)], scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        g = g.contiguous()
        dq = dk = dv = None
        if ctx.needs_input_grad[0]:
            dq, _ = ops.attn_bwd(q, k, v, out, lse, g, ctx.scale, False)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dk32, dv32 = ops.attn_bwd_dkv(q, k, v, out, lse, g, ctx.scale)
            dk = dk32.to(k.dtype) if ctx.needs_input_grad[1] else None
            dv = dv32.to(v.dtype) if ctx.needs_input_grad[2] else None
        return dq, dk, dv, None


def pad_head_dim(t: torch.Tensor) -> torch.Tensor:
    """[..., D] -> [..., ceil(D / 64) * 64] with zero columns (differentiable)."""
    D = t.shape[-1]
    return t if D % 64 == 0 else torch.nn.functional.pad(t, (0, -D % 64))


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(scale q k^T) v, [BH,N,D] x [BH,M,D] -> [BH,N,D] (HIP, flash-style).  Head dims that are not a multiple of 64 (SD1.x:
    40 / 80 / 160) are zero-padded to the next multiple — the kernels exist for 64 / 128 / 192; zero columns change no score and
    give zero outputs / gradients; ``scale`` stays the true head dim's."""
    D = q.shape[-1]
    if D % 64:
        q, k, v = pad_head_dim(q), pad_head_dim(k), pad_head_dim(v)
        return attention(q, k, v, scale)[..., :D]
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _VanillaAttention.apply(q, k, v, scale)
    out = torch.empty_like(q)
    ops.attn_fwd([(q, k, v, out, None)], scale)
    return out


# Opt-in (GD_ATTN_FP8=1, BASELINE configs[4]): vanilla no-grad self-attention on the fp8 matrix instruction.  An APPROXIMATION (e4m3 has 3
# mantissa bits: ~5e-2 on an attention output against the 1e-3 of the 16-bit path); its parity contract is oracle/ref_cpu.py:
# attention_fp8_oracle.  The hooked edit layers (warped queries, losses) always run the 16-bit kernels.
FP8_ATTENTION = os.environ.get("GD_ATTN_FP8", "0") == "1"
_LN2 = 0.6931471805599453


def attention_tok(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, heads: int, q_scaled: bool = False) -> torch.Tensor:
    """No-grad attention on the projections' own layout: q [B,N,heads*64], k/v [B,M,heads*64] -> [B,N,heads*64].
    q_scaled: q already carries scale*log2(e) (attention_processors._project_qkv)."""
    out = torch.empty_like(q)
    if FP8_ATTENTION and k.shape[1] % 64 == 0 and k.shape[1] >= 256:
        s_eff = _LN2 if q_scaled else float(scale)                      # pre-scaled queries: the scores already are exponents of 2
        ops.attn_fwd_fp8(ops.fp8_quantize(q, k, v, s_eff, heads=heads), s_eff, out, heads=heads)
        return out
    ops.attn_fwd([(q, k, v, out, None)], scale, heads=heads, q_scaled=q_scaled)
    return out


def compute_attention(q, k, scale, mask=None, fg_mask_warp=None, fg_mask=None, inpaint_mask=None):
    """attention_sharing.py:30-47: the materialised probability map softmax(scale q k^T), fp32 like the reference's
    autocast softmax.  The mask arguments are accepted and ignored — they are no-ops in the reference (the masked
    assignment goes to a temporary).  Only for callers that really want the map (AttentionStore at
    N <= 16^2); the hot path never calls this."""
    q, k = q.contiguous(), k.contiguous()
    BH, N, D = q.shape
    out = torch.empty_like(q)
    lse = torch.empty(BH, N, dtype=torch.float32, device=q.device)
    ops.attn_fwd([(q, k, k, out, lse)], scale)          # only lse is needed; k stands in for v
    P = ops.attn_probs(q, k, lse, None, scale)
    return P[:, :, : k.shape[1]].float()


def get_base_edit_qkv(q, k, v, batch_size, coords_base=None, coords_edit=None, use_cfg=True):
    """attention_sharing.py:210-242: [B*f, N, D] -> ([1,f,N,D] base (detached), edit) views."""
    nb = 2 * batch_size if use_cfg else batch_size
    h = q.shape[0] // nb
    q = q.reshape(nb, h, *q.shape[1:])
    k = k.reshape(nb, h, *k.shape[1:])
    v = v.reshape(nb, h, *v.shape[1:])
    q_base, k_base, v_base = (t[coords_base[0]:coords_base[1]] for t in (q, k, v))
    q_edit, k_edit, v_edit = (t[coords_edit[0]:coords_edit[1]] for t in (q, k, v))
    return q_base.detach(), k_base.detach(), v_base.detach(), q_edit, k_edit, v_edit


class AttentionControl(abc.ABC):
    """attention_sharing.py:110-153 — layer counter -> step counter, reproduced exactly: ``cur_step`` gates the
    self-replace window and obj_edit_step, and the driver undoes an optimisation pass with ``cur_step -= 1``."""

    def step_callback(self, x_t, transform_coords):
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return self.num_att_layers if LOW_RESOURCE else 0

    @abc.abstractmethod
    def forward(self, q, k, v, is_cross: bool, place_in_unet: str, transform_coords=None, scale=None, mask=None):
        raise NotImplementedError

    def __call__(self, q, k, v, is_cross: bool, place_in_unet: str, transform_coords=None, scale=None, mask=None):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            out = self.forward(q, k, v, is_cross, place_in_unet, transform_coords=transform_coords, scale=scale, mask=mask)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()
        return out

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0


class AttentionStore(AttentionControl):
    """attention_sharing.py:158-207.  Maps are kept only for N <= 16^2 ("to avoid memory overhead")."""

    @staticmethod
    def get_empty_store():
        return {"down_cross": [], "mid_cross": [], "up_cross": [], "down_self": [], "mid_self": [], "up_self": []}

    def forward(self, q, k, v, is_cross: bool, place_in_unet: str, transform_coords=None, scale=None, mask=None):
        key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
        if q.shape[1] <= 16 ** 2:
            self.step_store[key].append(compute_attention(q.detach(), k.detach(), scale))
        return attention(q, k, v, scale)

    def attn_store(self, attn, is_cross: bool, place_in_unet: str):
        key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
        if attn.shape[1] <= 16 ** 2:
            self.step_store[key].append(attn.detach())

    def between_steps(self):
        if len(self.attention_store) == 0:
            self.attention_store = self.step_store
        else:
            for key in self.step_store:
                self.attention_store[key] = self.attention_store[key] + self.step_store[key]
                if self.cur_step == 1:
                    self.attention_store["length_" + key] = len(self.step_store[key])
        self.step_store = self.get_empty_store()

    def get_average_attention(self):
        return {key: [item / self.cur_step for item in self.attention_store[key]] for key in self.attention_store}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}

    def __init__(self):
        super().__init__()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
