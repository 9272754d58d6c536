"""Bilinear forward splatting (softmax splatting).  Mirror of GeoDiffuser/utils/softsplat.py:232-524: ``softsplat(tenIn, tenFlow,
tenMetric, strMode)`` with the modes ``sum | avg | linear | soft`` (``-addeps | -zeroeps | -clipeps``) and the autograd Function
``softsplat_func`` — the reference's three cupy/CUDA kernel strings are gd_softsplat_fwd / gd_softsplat_bwd here (f32, like the
reference's ``custom_fwd(cast_inputs=torch.float32)``).  Dead on the reference's live path; present for completeness.
"""
from __future__ import annotations

import torch

from . import ops


class softsplat_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tenIn, tenFlow):
        tenIn, tenFlow = tenIn.float().contiguous(), tenFlow.float().contiguous()
        ctx.save_for_backward(tenIn, tenFlow)
        return ops.softsplat_fwd(tenIn, tenFlow)

    @staticmethod
    def backward(ctx, tenOutgrad):
        tenIn, tenFlow = ctx.saved_tensors
        return ops.softsplat_bwd(tenIn, tenFlow, tenOutgrad.float().contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def softsplat(tenIn: torch.Tensor, tenFlow: torch.Tensor, tenMetric, strMode: str):
    base = strMode.split("-")[0]
    assert base in ["sum", "avg", "linear", "soft"]
    if strMode in ("sum", "avg"):
        assert tenMetric is None
    if base in ("linear", "soft"):
        assert tenMetric is not None
    if strMode == "avg":
        tenIn = torch.cat([tenIn, tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])], 1)
    elif base == "linear":
        tenIn = torch.cat([tenIn * tenMetric, tenMetric], 1)
    elif base == "soft":
        tenIn = torch.cat([tenIn * tenMetric.exp(), tenMetric.exp()], 1)
    tenOut = softsplat_func.apply(tenIn, tenFlow)
    if base in ("avg", "linear", "soft"):
        tenNormalize = tenOut[:, -1:, :, :]
        suffix = strMode.split("-")[1] if "-" in strMode else "addeps"
        if suffix == "addeps":
            tenNormalize = tenNormalize + 0.0000001
        elif suffix == "zeroeps":
            tenNormalize = torch.where(tenNormalize == 0.0, torch.ones_like(tenNormalize), tenNormalize)
        elif suffix == "clipeps":
            tenNormalize = tenNormalize.clip(0.0000001, None)
        tenOut = tenOut[:, :-1, :, :] / tenNormalize
    return tenOut
