"""Geometry pre-pass (once per edit): depth + object mask + 4x4 transform -> the 3-channel warp grid and the amodal mask.

Mirror of GeoDiffuser/utils/vis_utils.py:404-479 (``get_transform_coordinates``) and the functions it reaches in
warp_utils.py (``forward_splatting_pytorch3d_warp`` :407-492, ``pixel2cam`` :738-747, ``cam2pixel_vanilla`` :599-645,
``get_mesh`` :364-399, ``splatter_mesh`` :235-298).  Everything stays on the GPU (the reference round-trips
GPU -> numpy -> CPU tensor -> GPU); the point splat and the mesh coverage are HIP kernels.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, warp_utils


def rotateAxis(degrees, axis):
    """vis_utils.py:26-66."""
    r = np.radians(degrees)
    c, s = np.cos(r), np.sin(r)
    if axis == 2:
        m = [[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    elif axis == 1:
        m = [[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]]
    else:
        m = [[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]]
    return torch.tensor(m)


def translateMatrix(x, y, z):
    """vis_utils.py:68-75."""
    t = torch.eye(4)
    t[0, 3] += x; t[1, 3] += y; t[2, 3] += z
    return t


def camera_matrix(focal_x, focal_y, c_x, c_y):
    """vis_utils.py:79-88."""
    return np.array([[focal_x, 0, c_x], [0, focal_y, c_y], [0, 0, 1]])


def get_mesh(t_coords_hw3: torch.Tensor, mask_hw: torch.Tensor):
    """warp_utils.py:304-399 (get_coordinate_array, get_indexing_grid, create_triangles, get_mesh) on the device: vertices = (-x, -y, Z)
    of the masked pixels in row-major order (float32 [V,3]); faces int32 [F,3] = for every 2x2 pixel quad the triangles
    (tl, tr, bl) and (bl, tr, br) — first all upper, then all lower triangles, each kept iff ITS OWN three corners are in the mask
    (:354-361: the test is per triangle, not per quad)."""
    m = mask_hw >= 0.5
    mapping = (torch.cumsum(m.reshape(-1).to(torch.int32), 0, dtype=torch.int32) - 1).reshape(m.shape)
    mapping = torch.where(m, mapping, torch.full_like(mapping, -1))
    tl, trr, bl, br = mapping[:-1, :-1], mapping[:-1, 1:], mapping[1:, :-1], mapping[1:, 1:]
    up = torch.stack([tl, trr, bl], 0).reshape(3, -1)
    lo = torch.stack([bl, trr, br], 0).reshape(3, -1)
    faces = torch.cat([up, lo], -1)
    faces = faces[:, faces.min(0).values > -1].t().contiguous()
    verts = t_coords_hw3[m].float().clone()
    verts[:, :2] = -verts[:, :2]                                                      # warp_utils.py:373-375
    return verts.contiguous(), faces


@torch.no_grad()
def get_transform_coordinates(image, depth, obj_mask=None, transform_in=torch.eye(4), use_softsplat=True, focal_length=550,
                              return_mesh=False, device="cuda", as_torch=False, preview=True):
    """-> (t_coords [H,W,3] f32, projected image [H,W,3] in [0,1][, amodal mask [1,1,H,W]]).
    numpy arrays like the reference unless ``as_torch``.  ``preview=False``: the projected preview image (a notebook aid: the edit drivers
    discard it, U/editor.py:546-547) is not computed — None is returned in its place (one upload, one 512^2 rasterisation and one
    composite less per edit); ``image`` may then be anything with the right ``shape``."""
    if preview:
        image = np.asarray(image)
    H, W = image.shape[0], image.shape[1]
    K = camera_matrix(focal_length, focal_length, W / 2.0, H / 2.0)
    depth = np.array(depth, copy=True)
    if np.sum(depth) == 0.5 * (depth.shape[0] * depth.shape[1]):                      # :410 constant-depth case
        depth = np.ones_like(depth) * 0.5
    else:
        depth = depth / (depth.max() + 1e-8)
        depth[depth > 0.95] = 1.0
    mask = (depth < 0.95) * 1.0
    if obj_mask is not None:
        mask = np.asarray(obj_mask) * mask
    dev = torch.device(device)
    mask_t = ((torch.as_tensor(mask)[None, None] >= 0.5) * 1.0).to(dev).float()
    d = torch.from_numpy(depth)[None, None].float().to(dev)
    Kt = torch.from_numpy(K)[None].float().to(dev)
    # pixel2cam (warp_utils.py:728-747)
    ii = torch.arange(0, H, device=dev).view(1, H, 1).expand(1, H, W).float()
    jj = torch.arange(0, W, device=dev).view(1, 1, W).expand(1, H, W).float()
    pix = torch.stack((jj, ii, torch.ones(1, H, W, device=dev)), dim=1).reshape(1, 3, -1)
    cam = (Kt.inverse() @ pix).reshape(1, 3, H, W) * d
    cam_flat = cam.reshape(1, 3, -1)
    # transform about the object centroid (warp_utils.py:423-437)
    center = torch.mean(cam_flat[:, :, mask_t.reshape(-1) >= 0.5], -1)
    T = torch.eye(4, device=dev)
    T[:3, 3] += -center[0]
    pose = (T[None].inverse() @ transform_in[None].to(dev).float() @ T[None]).float()
    rot, tr = pose[:, :3, :3], pose[:, :3, 3:]
    # cam2pixel_vanilla (warp_utils.py:599-645)
    pc = Kt @ (rot @ cam_flat + tr)
    X, Y, Z = pc[:, 0], pc[:, 1], pc[:, 2].clamp(min=1e-3)
    t_coords = torch.stack([2 * (X / Z) / (W - 1) - 1, 2 * (Y / Z) / (H - 1) - 1, Z], dim=2).reshape(1, H, W, 3)

    amodal = None
    if return_mesh:                                                                    # get_mesh + splatter_mesh
        verts, faces = get_mesh(t_coords[0], mask_t[0, 0])
        amodal = ops.mesh_coverage(verts, faces, H)[None, None]
    if not preview:
        out = (t_coords[0], None) if as_torch else (t_coords[0].cpu().numpy(), None)
        return out + (((amodal if as_torch else amodal.cpu().numpy()),) if return_mesh else ())
    # preview image (warp_utils.py:470); depth_projected of the reference is garbage and unused
    img = torch.from_numpy(image)[None].permute(0, 3, 1, 2).float().to(dev)
    idx, w = warp_utils.SPLATTER.tables(t_coords[0].reshape(-1, 3))
    from ._lib import GD_CHANNEL_MAJOR
    proj = ops.splat_composite(img.reshape(1, 3, H * W).contiguous(), idx, w, None, GD_CHANNEL_MAJOR).reshape(1, 3, H, W)
    valid = (t_coords[..., :2].abs().max(dim=-1)[0] <= 1)
    proj = (proj * valid[:, None]).clamp(0, 1)[0].permute(1, 2, 0)
    if as_torch:
        out = (t_coords[0], proj) + ((amodal,) if return_mesh else ())
    else:
        out = (t_coords[0].cpu().numpy(), proj.cpu().numpy()) + ((amodal.cpu().numpy(),) if return_mesh else ())
    return out
