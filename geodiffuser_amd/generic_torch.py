"""Mask / coordinate resampling helpers (setup-time torch ops, cached per resolution by the controller).

Mirror of the reference's GeoDiffuser/utils/generic_torch.py for the functions the hot path uses; same names
and argument meaning.  These run once per resolution per edit on whatever device the masks
live on — they are plumbing, not the accelerated path.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _resize_bilinear(x: torch.Tensor, s_z: int) -> torch.Tensor:
    # T.Resize(size, antialias=False, BILINEAR) == F.interpolate(bilinear, align_corners=False)   (generic_torch.py:185,205)
    return F.interpolate(x, size=(s_z, s_z), mode="bilinear", align_corners=False, antialias=False)


def binarize_tensor(t: torch.Tensor, thresh: float = 0.5) -> torch.Tensor:
    """generic_torch.py:122-124 (strict >)."""
    return (t > thresh) * 1.0


def reshape_transform_coords(transform_coords: torch.Tensor, in_mat=None, in_mat_shape=None) -> torch.Tensor:
    """generic_torch.py:156-186: [1,H,W,3] -> [1,s,s,3] where s = last dim of in_mat / in_mat_shape."""
    s_z = in_mat.shape[-1] if in_mat is not None else in_mat_shape[-1]
    return _resize_bilinear(transform_coords.permute(0, 3, 1, 2), s_z).permute(0, 2, 3, 1)


def reshape_attention_mask(mask: torch.Tensor, in_mat=None, in_mat_shape=None) -> torch.Tensor:
    """generic_torch.py:189-207."""
    s_m = int(mask.shape[-1])
    s_z = in_mat.shape[-1] if in_mat is not None else in_mat_shape[-1]
    m = torch.reshape(mask, (mask.shape[0], mask.shape[1], s_m, s_m))
    return _resize_bilinear(m, s_z)


def torch_erode(A: torch.Tensor, kernel: int = 3) -> torch.Tensor:
    """generic_torch.py:210-221."""
    k = torch.ones(1, 1, kernel, kernel).type_as(A)
    return (F.conv2d(A, k, padding=kernel // 2) == k.sum()) * 1.0


def torch_dilate(A: torch.Tensor, kernel: int = 3) -> torch.Tensor:
    """generic_torch.py:223-235."""
    k = torch.ones(1, 1, kernel, kernel).type_as(A)
    return (F.conv2d(A, k, padding=kernel // 2) >= 1) * 1.0


def norm_tensor(A: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """generic_torch.py:87-88."""
    return torch.sqrt(torch.sum(A * A) + eps)


class CoordinateDistances:
    """generic_torch.py:126-140.  The hot path never reads the [1,N,N] table (the HIP kernels evaluate the
    pixel-centre distance analytically); it is built on demand for the nearest-neighbour table and for callers
    that ask for it."""

    def __init__(self):
        self.coord_distance_dict = {}
        self.theta = torch.eye(3)[:2][None]

    def get_coord_distance(self, size, device="cuda"):
        key = (size, str(device))
        if key not in self.coord_distance_dict:
            grid = F.affine_grid(self.theta.to(device), (1, 1, size, size), align_corners=False)
            d = grid.reshape(1, -1, 2)
            self.coord_distance_dict[key] = torch.sqrt(torch.sum(torch.square(d[:, :, None] - d[:, None]), -1) + 1e-12)
        return self.coord_distance_dict[key]

    def clear(self):
        self.coord_distance_dict = {}


def gaussian_kernel_5x5(device=None) -> torch.Tensor:
    """GaussianSmoothing(dim=2, kernel_size=5) weights — generic_torch.py:26-55,143."""
    ks, sigma = 5, (5 // 2 * 2 / 6.0)
    ax = torch.arange(ks, dtype=torch.float32, device=device)
    g1 = 1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-((ax - (ks - 1) / 2) / (2 * sigma)) ** 2)
    k = g1[:, None] * g1[None, :]
    return k / k.sum()
