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


# mode -> how the per-pixel importance that travels with the features (as one extra channel) is formed; None: plain summation
_WEIGHTING = {
    "sum": None,
    "avg": lambda metric, like: like.new_ones(like.shape[0], 1, *like.shape[2:]),
    "linear": lambda metric, like: metric,
    "soft": lambda metric, like: metric.exp(),
}
# suffix -> what keeps the division by the splatted importance finite
_EPS = 1e-7
_GUARD = {
    "addeps": lambda z: z + _EPS,
    "zeroeps": lambda z: z.masked_fill(z == 0.0, 1.0),
    "clipeps": lambda z: z.clamp_min(_EPS),
}


def softsplat(tenIn: torch.Tensor, tenFlow: torch.Tensor, tenMetric, strMode: str):
    """Contract of U/softsplat.py:231-272: ``strMode`` = ``sum | avg | linear | soft`` with an optional ``-addeps | -zeroeps | -clipeps``
    suffix (default addeps; an unknown suffix: no guard, as in the reference); exactly ``sum`` / ``avg`` take no metric, ``linear`` / ``soft``
    need one.  One splat launch carries the features
    scaled by the importance and the importance itself; the quotient of the two is the normalised splat."""
    parts = strMode.split("-")
    mode, guard = parts[0], (parts[1] if len(parts) > 1 else "")
    if mode not in _WEIGHTING:
        raise AssertionError(f"softsplat: unknown mode {strMode!r}")
    # the reference's assertions, as it writes them (:235-238): the metric must be absent only for the EXACT strings 'sum' / 'avg' (with a
    # suffix it is accepted and ignored), and present for linear / soft whatever the suffix
    if strMode in ("sum", "avg") and tenMetric is not None:
        raise AssertionError(f"softsplat: mode {strMode!r} takes no metric")
    if mode in ("linear", "soft") and tenMetric is None:
        raise AssertionError(f"softsplat: mode {strMode!r} needs a metric")
    weigh = _WEIGHTING[mode]
    if weigh is None:
        return softsplat_func.apply(tenIn, tenFlow)
    if mode == "avg" and guard:
        # the reference tests the FULL string for 'avg' (U/softsplat.py:243): with a suffix nothing is appended and the caller's own last
        # channel is the importance
        splatted = softsplat_func.apply(tenIn, tenFlow)
    else:
        importance = weigh(tenMetric, tenIn)
        carried = tenIn if mode == "avg" else tenIn * importance
        splatted = softsplat_func.apply(torch.cat([carried, importance], 1), tenFlow)
    # (a suffix the reference does not know leaves the normaliser as it is, :254-266: a plain division)
    return splatted[:, :-1] / _GUARD.get(guard or "addeps", lambda z: z)(splatted[:, -1:])
