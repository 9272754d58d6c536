"""Forward point splat — mirror of the reference's warp entry points (GeoDiffuser/utils/warp_utils.py:28-179,798-837).

``warp_grid_edit`` / ``RasterizePointsXYsBlending`` keep the reference's names and argument meaning; the work is done
by the HIP rasterizer + compositor (geodiffuser_amd/csrc/raster.hip, splat.hip).  Difference by design: the point
rasterisation is cached — it depends only on (coords, S, radius, K) and the reference's per-step decay of those
constants is dead code — so it runs once per resolution per edit instead of once per hooked call.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import ops
from ._lib import GD_CHANNEL_MAJOR, GD_TOKEN_MAJOR


class RasterizePointsXYsBlending(torch.nn.Module):
    """warp_utils.py:28-176.  ``forward(pts3D [b,N,3], src [b,F,N]) -> [b,F,S,S]`` (fp16, as the reference)."""

    def __init__(self, radius=1.3, points_per_pixel=15, accumulation="alphacomposite", tau=1.0, rad_pow=2):
        super().__init__()
        if accumulation != "alphacomposite":
            raise NotImplementedError("only 'alphacomposite' is live in the reference (warp_utils.py:155)")
        self.radius = radius
        self.points_per_pixel = points_per_pixel
        self.accumulation = accumulation
        self.tau = tau
        self.rad_pow = rad_pow
        self.rasterization_dict: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}

    def clear_cache(self):
        self.rasterization_dict = {}

    @torch.no_grad()
    def tables(self, pts_one_cloud: torch.Tensor, cache_key=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(idx [S,S,K] int32, w [S*S,K] f32) for ONE cloud [N,3] given in the reference's coordinate
        convention (x right, y down, align-corners NDC); x,y are negated here (warp_utils.py:90-91) on a copy."""
        N = pts_one_cloud.shape[0]
        S = int(round(math.sqrt(N)))
        key = None
        if cache_key is not None:
            key = (cache_key, S, float(self.radius), int(self.points_per_pixel), float(self.tau), float(self.rad_pow))
            hit = self.rasterization_dict.get(key)
            if hit is not None:
                return hit
        pts = pts_one_cloud.to(torch.float32).clone()
        pts[:, :2] = -pts[:, :2]
        radius_ndc = float(self.radius) / float(S) * 2.0                       # :94
        idx, dist2 = ops.rasterize_points(pts.contiguous(), S, radius_ndc, int(self.points_per_pixel))
        w = ops.splat_weights(idx, dist2, radius_ndc, float(self.rad_pow), float(self.tau))
        if key is not None:
            self.rasterization_dict[key] = (idx, w)
        return idx, w

    @torch.no_grad()
    def forward(self, pts3D: torch.Tensor, src: torch.Tensor, shared_cloud: Optional[bool] = None) -> torch.Tensor:
        assert pts3D.size(2) == 3 and pts3D.size(1) == src.size(2)
        b, Fc, N = src.shape
        S = int(round(math.sqrt(N)))
        src_c = src.contiguous()
        if src_c.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            src_c = src_c.float()
        out = torch.empty(b, Fc, N, dtype=src_c.dtype, device=src.device)
        if shared_cloud is None:
            shared_cloud = b == 1 or all(torch.equal(pts3D[i], pts3D[0]) for i in range(1, b))
        if shared_cloud:
            idx, w = self.tables(pts3D[0])
            ops.splat_composite(src_c, idx, w, None, GD_CHANNEL_MAJOR, out=out)
        else:
            for i in range(b):
                idx, w = self.tables(pts3D[i])
                ops.splat_composite(src_c[i:i + 1], idx, w, None, GD_CHANNEL_MAJOR, out=out[i:i + 1])
        return out.reshape(b, Fc, S, S).to(torch.half)                          # :176


SPLATTER = RasterizePointsXYsBlending()


def warp_grid_edit(src, t_coords, padding_mode=None, mode=None, align_corners=False, depth=None, use_softsplat=True,
                   splatting_radius=None, splatting_tau=None, splatting_points_per_pixel=None):
    """warp_utils.py:798-837.  src [b,f,h,w], t_coords [b,h,w,3] -> [b,f,h,w] (fp16).  CPU inputs are moved to the
    GPU and the result moved back, as the reference does (:809-825)."""
    if not use_softsplat:
        raise NotImplementedError("use_softsplat=False is a dead branch in the reference (undefined MODE, warp_utils.py:828)")
    if splatting_radius is not None:
        SPLATTER.radius = splatting_radius
    if splatting_tau is not None:
        SPLATTER.tau = splatting_tau
    if splatting_points_per_pixel is not None:
        SPLATTER.points_per_pixel = splatting_points_per_pixel
    store_device = src.device
    if not src.is_cuda:
        src = src.to("cuda")
        t_coords = t_coords.to("cuda")
    b, f, h, w = src.shape
    out = SPLATTER(t_coords.reshape(b, h * w, -1), src.reshape(b, f, h * w))
    if store_device.type != "cuda":
        out = out.to(store_device)
    return out
