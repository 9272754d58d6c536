"""Post-process of the decoded image: masked per-channel histogram matching, on the device.

Same name / arguments / result as GeoDiffuser/utils/image_processing.py:24-77 (``masked_histogram_matching``): for each
colour channel the 256-bin histogram of ``source`` inside ``mask_source`` is mapped onto the histogram of ``template``
inside ``mask`` through their cumulative distributions; the float64 image ``lut[source]`` is returned.  The three
kernels behind ``gd_hist_match`` (include/geodiff_hip.h) produce exact counts and a binary64 LUT that is bit-identical
to the reference's numpy arithmetic.  numpy in -> numpy out like the reference; torch tensors in -> tensor out
(no host round trip).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

DEVICE = None          # default: the current CUDA device


def _as_u8(x, dev, what):
    if isinstance(x, torch.Tensor):
        t = x.to(dev)
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    if t.dtype != torch.uint8:
        raise TypeError(f"masked_histogram_matching: {what} must be uint8 (the reference indexes a 256-entry table with it)")
    return t.contiguous()


def _as_sel(m, dev, shape):
    """The reference's ``mask > 0.5`` evaluated in the mask's own precision, as uint8."""
    if isinstance(m, torch.Tensor):
        sel = (m > 0.5).to(dev)
    else:
        sel = torch.from_numpy(np.ascontiguousarray(np.asarray(m) > 0.5)).to(dev)
    if tuple(sel.shape) != tuple(shape):
        raise ValueError(f"masked_histogram_matching: mask shape {tuple(sel.shape)} != image shape {tuple(shape)}")
    return sel.to(torch.uint8).reshape(-1).contiguous()


def masked_histogram_matching(source, template, mask=None, mask_source=None):
    dev = torch.device(DEVICE) if DEVICE is not None else torch.device("cuda", torch.cuda.current_device())
    want_numpy = not isinstance(source, torch.Tensor)
    src = _as_u8(source, dev, "source")
    tmpl = _as_u8(template, dev, "template")
    if src.dim() == 2:
        src, tmpl = src[..., None], tmpl[..., None]
    H, W, C = src.shape
    if mask is None:                                   # image_processing.py:31-33 (identity mask)
        mask = torch.ones(H, W, device=dev)
    if mask_source is None:
        mask_source = mask
    out, _, _ = ops.hist_match(src.reshape(H * W, C), tmpl.reshape(H * W, C).contiguous(), _as_sel(mask_source, dev, (H, W)),
                               _as_sel(mask, dev, (H, W)))
    out = out.reshape(H, W, C)
    return out.cpu().numpy() if want_numpy else out
