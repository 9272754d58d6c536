"""SD2.1-base-shaped UNet harness (plain PyTorch-ROCm modules).

diffusers is not installable in the build image, and the benchmark needs *a* UNet2DConditionModel of the public
``stabilityai/stable-diffusion-2-1-base`` shape to host the attention processors.  This module
reproduces that architecture's module tree and names (``down_blocks.0.attentions.1.transformer_blocks.0.attn1`` ...),
the ``attn_processors`` / ``set_attn_processor`` members the reference drives (GeoDiffuser/utils/attention_processors.py:30,52,58-64)
and the ``Attention`` members its processors touch (``to_q/to_k/to_v/to_out``, ``head_to_batch_dim``, ``batch_to_head_dim``,
``scale``, ...).  When diffusers IS importable the same processors attach to the real model instead.

Convolutions / linears / norms are stock PyTorch-ROCm (MIOpen / rocBLAS) — plumbing, not the accelerated path; every
attention goes through a processor and therefore through the HIP kernels.  Weights are random-init (seeded) or loaded
from a diffusers-format state dict (the parameter names match).
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


# Element-wise fusions on the no-grad passes (gd_bias_residual / gd_geglu / gd_add_layer_norm / GroupNorm with the time-embedding
# add folded in, one batched time-embedding projection per pass): ~550 -> ~300 kernels per UNet pass.  GD_UNET_FUSED=0 = stock ops.
FUSED = os.environ.get("GD_UNET_FUSED", "1") == "1"
_DBG = {k: os.environ.get(k, "1") == "1" for k in ("GD_FUSE_RES", "GD_FUSE_TF", "GD_FUSE_GEGLU", "GD_FUSE_GN")}


def _fast(x: torch.Tensor, grad_ok: bool = False) -> bool:
    """no-grad passes; with ``grad_ok`` also passes that differentiate w.r.t. activations only (the caller checks that the
    parameters involved are frozen and routes through an autograd Function with a HIP backward)."""
    return FUSED and (grad_ok or not torch.is_grad_enabled()) and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)


class _GroupNormFn(torch.autograd.Function):
    """Fused channels-last GroupNorm (+SiLU, + per-(batch, channel) add) with a HIP backward w.r.t. x (frozen gamma / beta)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu, add_bc):
        from . import ops
        y, scratch = ops.group_norm_nhwc(x, weight, bias, groups, eps, silu, add_bc=add_bc, return_scratch=True)
        ctx.save_for_backward(x, weight, bias, scratch)
        ctx.add_bc = add_bc
        ctx.cfg = (groups, eps, silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, weight, bias, scratch = ctx.saved_tensors
        groups, eps, silu = ctx.cfg
        dy = dy.contiguous(memory_format=torch.channels_last)
        return ops.group_norm_nhwc_bwd(x, ctx.add_bc, weight, bias, dy, groups, eps, silu, scratch), None, None, None, None, None, None


class _AddLayerNormFn(torch.autograd.Function):
    """(s, y) = (a + b, LayerNorm(a + b)) with frozen gamma / beta: one HIP launch forward, one backward (the gradient that reaches the
    residual stream s directly is added inside it); a and b both receive that gradient."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        from . import ops
        s, y = ops.add_layer_norm(a, b, gamma, beta, eps)
        ctx.save_for_backward(s, gamma)
        ctx.eps = eps
        return s, y

    @staticmethod
    def backward(ctx, gs, gy):
        from . import ops
        s, gamma = ctx.saved_tensors
        if gy is None:
            ds = gs
        else:
            ds = ops.layer_norm_bwd(s, gamma, gy.contiguous(), None if gs is None else gs.contiguous(), ctx.eps)
        return ds, ds, None, None, None


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm(x) with frozen gamma / beta on the same kernels (the first norm of a transformer block: nothing is added)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        from . import ops
        _, y = ops.add_layer_norm(x, None, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x, gamma = ctx.saved_tensors
        return ops.layer_norm_bwd(x, gamma, gy.contiguous(), None, ctx.eps), None, None, None


class _GegluFn(torch.autograd.Function):
    """x[..., :C] * gelu(x[..., C:]) with a one-launch HIP backward (the optimisation pass differentiates through the feed-forward layers)."""

    @staticmethod
    def forward(ctx, x):
        from . import ops
        ctx.save_for_backward(x)
        return ops.geglu(x)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        (x,) = ctx.saved_tensors
        return ops.geglu_bwd(x, dy.contiguous())


class _BiasResidualFn(torch.autograd.Function):
    """y = x + bias[c] + res with a frozen bias: both gradients are the incoming gradient (no kernel)."""

    @staticmethod
    def forward(ctx, x, bias, res):
        from . import ops
        return ops.bias_residual(x, bias, res)

    @staticmethod
    def backward(ctx, g):
        return g, None, g


def group_norm_fused(x, weight, bias, groups, eps, silu, add_bc=None):
    """Fused GroupNorm on a channels-last 16-bit GPU tensor; differentiable w.r.t. x when the parameters are frozen."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _GroupNormFn.apply(x, weight, bias, groups, eps, silu, None if add_bc is None else add_bc.detach())
    from . import ops
    return ops.group_norm_nhwc(x, weight, bias, groups, eps, silu, add_bc=add_bc)


# 3x3 convolutions on the hand-written implicit-GEMM kernel (conv3x3.hip) instead of the library call; GD_CONV3X3=0 = F.conv2d (MIOpen).
CONV3X3 = os.environ.get("GD_CONV3X3", "1") == "1"
CONV1X1 = os.environ.get("GD_CONV1X1", "1") == "1"      # 1x1 shortcut convolutions as F.linear on channels_last views
# id(weight) -> (version, data_ptr, weight of the backward-data convolution).  One flipped / transposed copy per frozen 3x3 weight that
# an optimisation pass has differentiated through: about the 3x3 weight set again (~1 GB for SD2.1 in 16 bits, ~3 GB for SDXL), built on
# the first optimisation pass, kept across edits (a rebuild costs ~20 ms), dropped with its weight, by ``release_backward_weights()``
# (called from graphs.reset_opt_graphs()) or never built at all with GD_CONV3X3=0.
_WBWD = {}


def release_backward_weights():
    """Free the backward-data copies of the 3x3 convolution weights (they are rebuilt on the next optimisation pass)."""
    _WBWD.clear()



def _weight_bwd(w):
    """Weight of the convolution that maps dL/dout to dL/din for a stride-1, padding-1 3x3 convolution: taps flipped, channel roles
    swapped (w_b[c, k, ky, kx] = w[k, c, 2-ky, 2-kx]); built once per (frozen) weight."""
    c = _WBWD.get(id(w))
    if c is None or c[0] != w._version or c[1] != w.data_ptr():
        wb = w.detach().flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
        if c is None:
            weakref.finalize(w, _WBWD.pop, id(w), None)          # the entry goes when the weight does
        c = _WBWD[id(w)] = (w._version, w.data_ptr(), wb)
    return c[2]


class _Conv3x3Fn(torch.autograd.Function):
    """Stride-1 3x3 convolution (+ residual) with a frozen weight: the gradient w.r.t. the input is the same kernel on the flipped
    weight, the residual's gradient is the incoming one."""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        from . import ops
        ctx.save_for_backward(weight)
        ctx.x_grad = x.requires_grad
        return ops.conv3x3(x, weight, bias, res=res)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        (weight,) = ctx.saved_tensors
        g = g.contiguous(memory_format=torch.channels_last)
        gx = ops.conv3x3(g, _weight_bwd(weight)) if ctx.x_grad else None
        return gx, None, None, (g if ctx.needs_input_grad[3] else None)


def conv1x1(x, weight, bias=None):
    """1x1 convolution of a channels_last tensor = a GEMM over its [pixels, C] rows: F.linear (hipBLASLt) on zero-copy views, which is
    faster on these shapes than the library's convolution kernels (and than gd_conv3x3's tile kernel on one tap)."""
    if CONV1X1 and x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and x.dtype in (torch.float16, torch.bfloat16):
        n, c, h, w = x.shape
        k = weight.shape[0]
        y = F.linear(x.permute(0, 2, 3, 1).reshape(n * h * w, c), weight.reshape(k, c), bias)
        return y.view(n, h, w, k).permute(0, 3, 1, 2)
    return F.conv2d(x, weight, bias)


def conv3x3(x, weight, bias=None, stride=1, upsample=False, res=None):
    """F.conv2d(x [upsampled 2x nearest], weight, bias, stride, padding=1) [+ res] — on gd_conv3x3 when the call qualifies (16-bit
    channels_last GPU tensors, C % 64 == 0, K % 8 == 0, frozen weights), on the library otherwise (conv_in / conv_out, fp32, CPU)."""
    if CONV3X3 and x.is_cuda and not weight.requires_grad and (bias is None or not bias.requires_grad):
        from . import ops
        if ops.conv3x3_supported(x, weight, stride) and (res is None or res.is_contiguous(memory_format=torch.channels_last)):
            if torch.is_grad_enabled() and (x.requires_grad or (res is not None and res.requires_grad)):
                # the backward-data kernel needs C' = K % 64 == 0; strided / upsampled calls keep autograd's library backward
                if stride == 1 and not upsample and weight.shape[0] % 64 == 0:
                    return _Conv3x3Fn.apply(x, weight, bias, res)
            else:
                return ops.conv3x3(x, weight, bias, stride=stride, upsample=upsample, res=res)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, weight, bias, stride=stride, padding=1)
    return y if res is None else y + res


class UNetOutput(dict):
    """``out["sample"]`` / ``out.sample`` / ``out[0]`` like diffusers' UNet2DConditionOutput."""

    @property
    def sample(self):
        return self["sample"]

    def __getitem__(self, k):
        if isinstance(k, int):
            return list(self.values())[k]
        return super().__getitem__(k)


class GroupNormAct(nn.GroupNorm):
    """nn.GroupNorm with an optional fused SiLU.  On no-grad passes over channels-last 16-bit GPU activations it runs the fused
    HIP kernel (gd_group_norm_nhwc: no NCHW round trip, 2 launches); otherwise stock PyTorch (autograd-capable)."""

    def fusable(self, x) -> bool:
        frozen = not (self.weight.requires_grad or self.bias.requires_grad)
        cpg = self.num_channels // self.num_groups
        return (_DBG["GD_FUSE_GN"] and _fast(x, grad_ok=frozen) and x.dim() == 4 and (cpg >= 8 or cpg == 4) and self.num_channels % 8 == 0
                and x.is_contiguous(memory_format=torch.channels_last) and self.weight.dtype == x.dtype)

    def forward(self, x, silu: bool = False):
        if self.fusable(x):
            return group_norm_fused(x, self.weight, self.bias, self.num_groups, self.eps, silu)
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.silu(y) if silu else y


class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.is_cross_attention = cross_attention_dim is not None
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv_dim, inner, bias=False)
        self.to_v = nn.Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = None

    def set_processor(self, processor):
        self.processor = processor

    def head_to_batch_dim(self, t: torch.Tensor) -> torch.Tensor:
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def batch_to_head_dim(self, t: torch.Tensor) -> torch.Tensor:
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        if _DBG["GD_FUSE_GEGLU"] and self.proj.out_features % 16 == 0:
            if _fast(x):
                from . import ops
                return ops.geglu(self.proj(x))
            if _fast(x, grad_ok=True) and x.requires_grad and not self.proj.weight.requires_grad:
                return _GegluFn.apply(self.proj(x).contiguous())
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        if _DBG["GD_FUSE_TF"] and _fast(x) and x.is_contiguous() and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048:
            from . import ops
            a = self.attn1(ops.add_layer_norm(x, None, self.norm1.weight, self.norm1.bias, self.norm1.eps)[1])
            x, h = ops.add_layer_norm(a.contiguous(), x, self.norm2.weight, self.norm2.bias, self.norm2.eps)   # x = a + x; h = LN(x)
            a = self.attn2(h, encoder_hidden_states=ctx)
            x, h = ops.add_layer_norm(a.contiguous(), x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return self.ff(h) + x
        if (_DBG["GD_FUSE_TF"] and _fast(x, grad_ok=True) and x.requires_grad and x.is_contiguous() and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048
                and not any(p.requires_grad for n_ in (self.norm1, self.norm2, self.norm3) for p in n_.parameters())):
            # optimisation pass: the same fused add + LayerNorm with a one-launch backward
            h = _LayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            a = self.attn1(h)
            x, h = _AddLayerNormFn.apply(a.contiguous(), x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            a = self.attn2(h, encoder_hidden_states=ctx)
            x, h = _AddLayerNormFn.apply(a.contiguous(), x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return self.ff(h) + x
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=ctx) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    """use_linear_projection=True variant (SD2.x)."""

    def __init__(self, channels, heads, dim_head, cross_attention_dim, depth=1):
        super().__init__()
        self.norm = GroupNormAct(32, channels, eps=1e-6)
        self.proj_in = nn.Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, dim_head, cross_attention_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(channels, channels)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        res = x
        x = self.norm(x)
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, ctx)
        x = self.proj_out(x)
        x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
        return x + res


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch=1280):
        super().__init__()
        self.norm1 = GroupNormAct(32, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = GroupNormAct(32, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.fused_ok = cin % 32 == 0 and cout % 32 == 0 and cin // 32 >= 8 and cout // 32 >= 8      # fused GroupNorm: >= 8 channels / group
        self._tb = None            # set by the UNet for one forward: this block's slice of the batched time-embedding projection

    def _fused(self, x, tb):
        """no-grad pass: conv biases folded away (conv1's into tb, conv2's + shortcut's into one epilogue with the residual add),
        the time-embedding add folded into GroupNorm 2."""
        from . import ops
        h = conv3x3(self.norm1(x, silu=True), self.conv1.weight)
        h = group_norm_fused(h, self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps, True, add_bc=tb)
        if self.conv_shortcut is not None:
            x = conv1x1(x, self.conv_shortcut.weight)
        if CONV3X3 and ops.conv3x3_supported(h, self.conv2.weight):           # bias + residual in the convolution's epilogue
            return conv3x3(h, self.conv2.weight, self._out_bias(), res=x)
        h = conv3x3(h, self.conv2.weight)
        if torch.is_grad_enabled() and (h.requires_grad or x.requires_grad):
            return _BiasResidualFn.apply(h, self._out_bias(), x)
        return ops.bias_residual(h, self._out_bias(), x)

    def _out_bias(self):
        ver = self.conv2.bias._version + (self.conv_shortcut.bias._version if self.conv_shortcut is not None else 0)
        c = self.__dict__.get("_ob")
        if c is None or c[0] != ver or c[1].dtype != self.conv2.bias.dtype or c[1].device != self.conv2.bias.device:
            b = self.conv2.bias.detach()
            if self.conv_shortcut is not None:
                b = b + self.conv_shortcut.bias.detach()
            c = self.__dict__["_ob"] = (ver, b.contiguous())
        return c[1]

    def forward(self, x, temb):
        tb, self._tb = self._tb, None
        if _DBG["GD_FUSE_RES"] and tb is not None and x.is_contiguous(memory_format=torch.channels_last) and self.norm1.fusable(x) and self.norm2.fusable(x):
            return self._fused(x, tb)
        h = conv3x3(self.norm1(x, silu=True), self.conv1.weight, self.conv1.bias)
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = conv3x3(self.norm2(h, silu=True), self.conv2.weight, self.conv2.bias)
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return conv3x3(x, self.conv.weight, self.conv.bias, stride=2)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return conv3x3(x, self.conv.weight, self.conv.bias, upsample=True)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, heads, ctx_dim, n_layers=2, attn=True, down=True, temb_ch=1280, depth=1):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_ch) for i in range(n_layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, ctx_dim, depth) for _ in range(n_layers)]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, heads, ctx_dim, temb_ch=1280, depth=1):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch), ResnetBlock2D(ch, ch, temb_ch)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, ch // heads, ctx_dim, depth)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, heads, ctx_dim, n_layers=3, attn=True, up=True, temb_ch=1280, depth=1):
        super().__init__()
        res = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            r_in = prev if i == 0 else cout
            res.append(ResnetBlock2D(r_in + skip, cout, temb_ch))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, ctx_dim, depth) for _ in range(n_layers)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class UNet2DConditionModel(nn.Module):
    """SD2.1-base: block_out_channels (320,640,1280,1280), heads (5,10,20,20) (head dim 64), ctx 1024, 32 attention
    processors (16 transformer blocks x {attn1, attn2})."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), heads=(5, 10, 20, 20),
                 cross_attention_dim=1024, layers_per_block=2, attn_levels=None, transformer_depth=None, mid_depth=None,
                 addition_time_embed_dim=None, addition_text_embed_dim=None):
        """Defaults: SD2.1-base.  SDXL-base (``sdxl_unet``): three levels (320, 640, 1280), attention on levels 1 and 2 only with 2 and
        10 transformer blocks per Transformer2DModel, context 2048 (two text encoders), and the "text_time" additional embedding
        (pooled text embedding + 6 micro-conditioning ids through a second TimestepEmbedding, added to the time embedding)."""
        super().__init__()
        ch = block_out_channels
        L = len(ch)
        attn_levels = tuple(attn_levels) if attn_levels is not None else tuple(i < L - 1 for i in range(L))
        depth = tuple(transformer_depth) if transformer_depth is not None else (1,) * L
        mid_depth = mid_depth if mid_depth is not None else depth[-1]
        temb_ch = ch[0] * 4
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb_ch)
        self.add_time_dim = addition_time_embed_dim
        if addition_time_embed_dim:
            self.add_embedding = TimestepEmbedding(addition_text_embed_dim + 6 * addition_time_embed_dim, temb_ch)
        self.default_added_cond = None         # (text_embeds [1 | B, E], time_ids [1 | B, 6]) used when the caller passes no added_cond_kwargs
        self.down_blocks = nn.ModuleList()
        cin = ch[0]
        for i, cout in enumerate(ch):
            last = i == L - 1
            self.down_blocks.append(DownBlock(cin, cout, heads[i], cross_attention_dim, layers_per_block, attn=attn_levels[i], down=not last,
                                              temb_ch=temb_ch, depth=depth[i]))
            cin = cout
        self.mid_block = MidBlock(ch[-1], heads[-1], cross_attention_dim, temb_ch=temb_ch, depth=mid_depth)
        self.up_blocks = nn.ModuleList()
        rev, rheads, rattn, rdepth = list(reversed(ch)), list(reversed(heads)), list(reversed(attn_levels)), list(reversed(depth))
        prev = rev[0]
        for i, cout in enumerate(rev):
            cin_skip = rev[min(i + 1, L - 1)]
            last = i == L - 1
            self.up_blocks.append(UpBlock(cin_skip, cout, prev, rheads[i], cross_attention_dim, layers_per_block + 1, attn=rattn[i], up=not last,
                                          temb_ch=temb_ch, depth=rdepth[i]))
            prev = cout
        self.conv_norm_out = GroupNormAct(32, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)
        self.t_dim = ch[0]
        from .attention_processors import VanillaAttentionProcessor
        self.set_attn_processor(VanillaAttentionProcessor())

    # -- the two members the reference drives -----------------------------------------------------------
    def _attn_modules(self):
        mods = self.__dict__.get("_attn_mods")
        if mods is None:                       # the module tree is static: walk it once
            mods = [(f"{name}.processor", m) for name, m in self.named_modules() if isinstance(m, Attention)]
            self.__dict__["_attn_mods"] = mods
        return mods

    @property
    def attn_processors(self) -> Dict[str, object]:
        return {name: m.processor for name, m in self._attn_modules()}

    def set_attn_processor(self, processor: Union[object, Dict[str, object]]):
        for name, m in self._attn_modules():
            m.set_processor(processor[name] if isinstance(processor, dict) else processor)

    def _frozen(self) -> bool:
        return not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters())

    def _project_all_temb(self, temb):
        """One GEMM for the time-embedding projections of all ResNet blocks (22 GEMMs + 22 SiLUs otherwise), with each block's
        conv1 bias folded into the projection bias; every fusable block gets its [B, cout] column slice for this forward."""
        blocks = self.__dict__.get("_res_blocks")
        if blocks is None:
            blocks = self.__dict__["_res_blocks"] = [m for m in self.modules() if isinstance(m, ResnetBlock2D) and m.fused_ok]
        if not blocks:
            return
        ver = sum(b.time_emb_proj.weight._version + b.time_emb_proj.bias._version + b.conv1.bias._version for b in blocks)
        c = self.__dict__.get("_temb_cat")
        w0 = blocks[0].time_emb_proj.weight
        if c is None or c[0] != ver or c[1].dtype != w0.dtype or c[1].device != w0.device:
            W = torch.cat([b.time_emb_proj.weight.detach() for b in blocks], 0).contiguous()
            bias = torch.cat([b.time_emb_proj.bias.detach() + b.conv1.bias.detach() for b in blocks], 0).contiguous()
            c = self.__dict__["_temb_cat"] = (ver, W, bias)
        tb_all = F.linear(F.silu(temb), c[1], c[2])                     # [B, sum cout]
        off = 0
        for b in blocks:
            n = b.time_emb_proj.out_features
            b._tb = tb_all[:, off:off + n]
            off += n

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states=None, return_dict=True, **kw):
        dt = self.dtype
        x = sample.to(dt)
        ctx = encoder_hidden_states.to(dt)
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], device=x.device)
        t = t.reshape(-1).to(x.device).expand(x.shape[0])
        temb = self.time_embedding(timestep_embedding(t, self.t_dim).to(dt))
        if self.add_time_dim:                  # SDXL "text_time" conditioning (diffusers UNet2DConditionModel.get_aug_embed)
            cond = kw.get("added_cond_kwargs") or self.default_added_cond
            if cond is None:
                raise ValueError("this UNet needs added_cond_kwargs={'text_embeds', 'time_ids'} (or unet.default_added_cond)")
            text_embeds, time_ids = (cond["text_embeds"], cond["time_ids"]) if isinstance(cond, dict) else cond
            B = x.shape[0]
            text_embeds = text_embeds.to(x.device, dt).expand(B, -1) if text_embeds.shape[0] == 1 else text_embeds.to(x.device, dt)
            time_ids = time_ids.to(x.device).expand(B, -1) if time_ids.shape[0] == 1 else time_ids.to(x.device)
            tid = timestep_embedding(time_ids.reshape(-1), self.add_time_dim).reshape(B, -1).to(dt)
            temb = temb + self.add_embedding(torch.cat([text_embeds, tid], dim=-1))
        if self.conv_in.weight.is_contiguous(memory_format=torch.channels_last) and not self.conv_in.weight.is_contiguous():
            x = x.contiguous(memory_format=torch.channels_last)
        if _fast(x, grad_ok=self._frozen()) and x.is_contiguous(memory_format=torch.channels_last):
            self._project_all_temb(temb)
        x = self.conv_in(x)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, ctx)
            skips.extend(outs)
        x = self.mid_block(x, temb, ctx)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, ctx)
        x = self.conv_out(self.conv_norm_out(x, silu=True))
        if not return_dict:
            return (x,)
        return UNetOutput(sample=x)


def sdxl_unet(tiny: bool = False, ctx_dim: int = None, text_embed_dim: int = None) -> UNet2DConditionModel:
    """SDXL-base topology (public ``stabilityai/stable-diffusion-xl-base-1.0`` UNet config; NOT in /root/reference, whose SDXL line is
    commented out, U/diffusion.py:106): block_out_channels (320, 640, 1280), head dim 64 (5 / 10 / 20 heads), attention on the two
    lower levels only with 2 / 10 transformer blocks each (mid: 10), context 2048, text_time additional embedding (1280 + 6 x 256).
    At 1024^2 (latent 128^2) the hooked attention layers sit at 64^2 tokens (10 heads) and 32^2 tokens (20 heads): shapes the D = 64
    kernels already serve.  ``tiny``: same topology, narrow."""
    if tiny:
        return UNet2DConditionModel(block_out_channels=(64, 128, 128), heads=(1, 2, 2), cross_attention_dim=ctx_dim or 96,
                                    attn_levels=(False, True, True), transformer_depth=(0, 1, 2), mid_depth=2,
                                    addition_time_embed_dim=32, addition_text_embed_dim=text_embed_dim or 64)
    return UNet2DConditionModel(block_out_channels=(320, 640, 1280), heads=(5, 10, 20), cross_attention_dim=ctx_dim or 2048,
                                attn_levels=(False, True, True), transformer_depth=(0, 2, 10), mid_depth=10,
                                addition_time_embed_dim=256, addition_text_embed_dim=text_embed_dim or 1280)


def sd14_unet(tiny: bool = False, ctx_dim: int = None) -> UNet2DConditionModel:
    """SD1.x topology — the reference's DEFAULT model (CompVis/stable-diffusion-v1-4, U/editor.py:58): the SD2.1 block structure with
    8 heads on every level, i.e. head dims 40 / 80 / 160 / 160, and a 768-wide text context.  ``tiny``: 4 heads over 160 / 320 / 640 / 640
    channels — the same head dims."""
    if tiny:
        return UNet2DConditionModel(block_out_channels=(160, 320, 640, 640), heads=(4, 4, 4, 4), cross_attention_dim=ctx_dim or 64)
    return UNet2DConditionModel(block_out_channels=(320, 640, 1280, 1280), heads=(8, 8, 8, 8), cross_attention_dim=ctx_dim or 768)


def tiny_unet(ctx_dim=64) -> UNet2DConditionModel:
    """A small model with the same topology (head dim 64) for smoke tests."""
    return UNet2DConditionModel(block_out_channels=(64, 128, 128, 128), heads=(1, 2, 2, 2), cross_attention_dim=ctx_dim)
