"""Seeded synthetic inputs shared by ``oracle/gen_golden.py`` (which records what the reference
computes for them) and the tests (which regenerate the same inputs and compare).

Only IEEE elementwise numpy arithmetic and ``np.random.default_rng`` are used, so the inputs are
bit-identical wherever they are regenerated; the fixtures therefore store OUTPUTS only.
Nothing here is reference code.
"""
from __future__ import annotations

import numpy as np

IMG = 512
FOCAL = 550.0


def ellipse_mask(cx=236.0, cy=262.0, ax=70.0, ay=58.0, size=IMG) -> np.ndarray:
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    return ((((u - cx) / ax) ** 2 + ((v - cy) / ay) ** 2) <= 1.0).astype(np.float32)


def coords_translate(dx_px=64.0, dy_px=-24.0, z=0.5, size=IMG) -> np.ndarray:
    """2-D translation on constant depth: t_coords[v,u] = (x_ndc, y_ndc, Z) (align-corners NDC)."""
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    x = np.float32(2.0) * (u + np.float32(dx_px)) / np.float32(size - 1) - np.float32(1.0)
    y = np.float32(2.0) * (v + np.float32(dy_px)) / np.float32(size - 1) - np.float32(1.0)
    zz = np.full_like(x, np.float32(z))
    return np.stack([x, y, zz], -1)[None].astype(np.float32)


def _centroid(mask: np.ndarray):
    """Exact (integer-sum) centroid of a binary mask, as float32 (cu, cv)."""
    vs, us = np.nonzero(mask > 0.5)
    n = max(int(us.size), 1)
    return np.float32(int(us.sum()) / n), np.float32(int(vs.sum()) / n)


def coords_rotate_y(deg=20.0, mask=None, size=IMG) -> np.ndarray:
    """Rotation about the vertical axis through the object's centre pixel on a tilted depth plane
    (elementwise float32 arithmetic only, so it regenerates bit-identically)."""
    import math
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    if mask is None:
        mask = ellipse_mask(size=size)
    cu, cv = _centroid(mask)

    def cam(uu, vv):
        d = np.float32(0.55) + np.float32(0.2) * (uu / np.float32(size) - np.float32(0.5))
        return ((uu - np.float32(size / 2)) / np.float32(FOCAL) * d,
                (vv - np.float32(size / 2)) / np.float32(FOCAL) * d, d)

    cx, cy, cz = cam(u, v)
    c = cam(cu, cv)
    co = np.float32(math.cos(math.radians(deg)))
    si = np.float32(math.sin(math.radians(deg)))
    x0, y0, z0 = cx - c[0], cy - c[1], cz - c[2]
    px = co * x0 + si * z0 + c[0]
    py = y0 + c[1]
    pz = np.maximum(-si * x0 + co * z0 + c[2], np.float32(1e-3))
    uu = np.float32(FOCAL) * px / pz + np.float32(size / 2)
    vv = np.float32(FOCAL) * py / pz + np.float32(size / 2)
    x = np.float32(2.0) * uu / np.float32(size - 1) - np.float32(1.0)
    y = np.float32(2.0) * vv / np.float32(size - 1) - np.float32(1.0)
    return np.stack([x, y, pz], -1)[None].astype(np.float32)


def coords_scale(s=0.8, z=0.5, mask=None, size=IMG) -> np.ndarray:
    """Uniform in-plane scaling about the mask centroid (many-to-one splat: deep pixel queues)."""
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    if mask is None:
        mask = ellipse_mask(size=size)
    cu, cv = _centroid(mask)
    uu = (u - cu) * np.float32(s) + cu
    vv = (v - cv) * np.float32(s) + cv
    x = np.float32(2.0) * uu / np.float32(size - 1) - np.float32(1.0)
    y = np.float32(2.0) * vv / np.float32(size - 1) - np.float32(1.0)
    zz = (np.float32(z) + np.float32(0.1) * (v / np.float32(size))).astype(np.float32)
    return np.stack([x, y, zz], -1)[None].astype(np.float32)


def make_coords(kind: str, mask: np.ndarray) -> np.ndarray:
    if kind == "translate":
        return coords_translate()
    if kind == "rotate":
        return coords_rotate_y(mask=mask)
    if kind == "scale":
        return coords_scale(mask=mask)
    raise ValueError(kind)


def amodal_input(mask: np.ndarray, dx=64, dy=-24) -> np.ndarray:
    """A synthetic 'projected amodal mask' input [1,1,H,W]: the mask shifted by the translation and
    grown by 6 px (stands in for the mesh coverage; it is only an INPUT of the path under test)."""
    m = np.roll(np.roll(mask, dy, axis=0), dx, axis=1)
    g = np.zeros_like(m)
    for oy in range(-6, 7):
        for ox in range(-6, 7):
            if ox * ox + oy * oy <= 36:
                g = np.maximum(g, np.roll(np.roll(m, oy, axis=0), ox, axis=1))
    return g[None, None].astype(np.float32)


def half_round(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).astype(np.float32)


def make_qkv(seed: int, B: int, f: int, N: int, M: int, D: int, spike: bool = True):
    """q [B*f,N,D], k/v [B*f,M,D]; values rounded through fp16 so the same numbers can be fed to the
    fp16 HIP path exactly.  A mild structure (shared low-rank term) keeps the softmax peaky enough
    that the removal-loss arg-max is well separated."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B * f, N, D), dtype=np.float32)
    k = rng.standard_normal((B * f, M, D), dtype=np.float32)
    v = rng.standard_normal((B * f, M, D), dtype=np.float32)
    if spike and M == N:
        pos = rng.standard_normal((1, N, D), dtype=np.float32)
        q = q + np.float32(1.5) * pos
        k = k + np.float32(1.5) * pos
    return half_round(q * np.float32(1.2)), half_round(k * np.float32(1.2)), half_round(v)


# Controller-forward golden cases (G6).  name -> parameters
CONTROLLER_CASES = {
    # name:            kind      S   f  D   cross  cfg    cur_step  coords       quant
    "edit_self_opt_32":   dict(kind="edit", S=32, f=2, D=16, cross=False, cfg=False, cur_step=3, coords="translate", quant=True, seed=11),
    "edit_cross_opt_32":  dict(kind="edit", S=32, f=2, D=16, cross=True, cfg=False, cur_step=3, coords="translate", quant=True, seed=12),
    "edit_self_cfg_32":   dict(kind="edit", S=32, f=2, D=16, cross=False, cfg=True, cur_step=3, coords="rotate", quant=True, seed=13),
    "edit_cross_cfg_32":  dict(kind="edit", S=32, f=2, D=16, cross=True, cfg=True, cur_step=46, coords="rotate", quant=True, seed=14),
    "edit_self_late_16":  dict(kind="edit", S=16, f=2, D=16, cross=False, cfg=True, cur_step=48, coords="translate", quant=True, seed=15),
    "edit_self_opt_64":   dict(kind="edit", S=64, f=1, D=8, cross=False, cfg=False, cur_step=0, coords="rotate", quant=True, seed=16),
    "edit_cross_opt_64":  dict(kind="edit", S=64, f=1, D=8, cross=True, cfg=False, cur_step=0, coords="scale", quant=True, seed=17),
    "edit_self_opt_32_noquant": dict(kind="edit", S=32, f=2, D=16, cross=False, cfg=False, cur_step=3, coords="scale", quant=False, seed=18),
    "rem_self_opt_32":    dict(kind="remover", S=32, f=2, D=16, cross=False, cfg=False, cur_step=3, coords="translate", quant=False, seed=21),
    "rem_cross_opt_32":   dict(kind="remover", S=32, f=2, D=16, cross=True, cfg=False, cur_step=3, coords="translate", quant=False, seed=22),
    "rem_self_cfg_past_32": dict(kind="remover", S=32, f=2, D=16, cross=False, cfg=True, cur_step=46, coords="translate", quant=False, seed=23),
    "rem_cross_cfg_16":   dict(kind="remover", S=16, f=2, D=16, cross=True, cfg=True, cur_step=10, coords="translate", quant=False, seed=24),
}
NUM_STEPS = 50
SELF_REPLACE = 0.95
OBJ_EDIT_STEP = 0.9


# ---- N2: masked histogram matching ---------------------------------------------------------------------
HIST_CASES = ("smooth_96", "sparse_levels_64", "disjoint_masks_80", "full_mask_48", "tiny_mask_40", "constant_32")


def hist_case(name: str):
    """-> (source uint8 [H,W,3], template uint8 [H,W,3], mask float64 [H,W] | None, mask_source float64 [H,W] | None)."""
    size = int(name.rsplit("_", 1)[1])
    rng = np.random.default_rng(abs(hash_name(name)))
    yy, xx = np.mgrid[0:size, 0:size]
    blob = (((xx - size * 0.45) / (size * 0.3)) ** 2 + ((yy - size * 0.55) / (size * 0.25)) ** 2 <= 1.0) * 1.0
    if name.startswith("smooth"):
        src = np.clip(rng.normal(120, 40, (size, size, 3)), 0, 255).astype(np.uint8)
        tmpl = np.clip(rng.normal(90, 25, (size, size, 3)) + xx[..., None] * 0.5, 0, 255).astype(np.uint8)
        return src, tmpl, 1.0 - blob, 1.0 - blob
    if name.startswith("sparse_levels"):                    # many empty bins -> plateaus (duplicate quantiles) in both CDFs
        src = (rng.integers(0, 6, (size, size, 3)) * 50).astype(np.uint8)
        tmpl = (rng.integers(0, 4, (size, size, 3)) * 80 + 7).astype(np.uint8)
        return src, tmpl, blob, 1.0 - blob
    if name.startswith("disjoint_masks"):
        src = rng.integers(0, 256, (size, size, 3)).astype(np.uint8)
        tmpl = rng.integers(30, 200, (size, size, 3)).astype(np.uint8)
        return src, tmpl, blob * 0.75, (1.0 - blob) * 0.51            # soft values either side of 0.5
    if name.startswith("full_mask"):
        src = rng.integers(0, 256, (size, size, 3)).astype(np.uint8)
        tmpl = rng.integers(0, 256, (size, size, 3)).astype(np.uint8)
        return src, tmpl, None, None                                   # the reference's identity-mask branch
    if name.startswith("tiny_mask"):
        src = rng.integers(0, 256, (size, size, 3)).astype(np.uint8)
        tmpl = rng.integers(0, 256, (size, size, 3)).astype(np.uint8)
        m = np.zeros((size, size)); m[3, 4] = 1.0; m[10, 2] = 1.0; m[11, 30] = 1.0
        ms = np.zeros((size, size)); ms[0, 0] = 1.0
        return src, tmpl, m, ms
    if name.startswith("constant"):
        src = np.full((size, size, 3), 77, np.uint8)
        tmpl = np.full((size, size, 3), 201, np.uint8); tmpl[::2] = 13
        return src, tmpl, blob, blob
    raise KeyError(name)


def hash_name(name: str) -> int:
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h


# ---- N3: experiment folders ------------------------------------------------------------------------------
EXP_CASES = (("Mix", 1), ("Mix", 2), ("Removal", 1), ("Rotation_2D", 1))
TRANSFORM_CASES = (dict(translation_x=0.1), dict(translation_x=-0.2, translation_y=0.05, translation_z=0.3, rotation_y=25.0),
                   dict(rotation_x=10.0, rotation_y=-30.0, rotation_z=45.0), dict(scale_x=0.8, scale_y=1.2, scale_z=0.5, rotation_z=-12.5),
                   dict(translation_z=-0.1, scale_x=-1.0, rotation_x=90.0), dict())


def exp_case(cat: str, idx: int):
    """Small synthetic experiment (48 x 64 image): arrays in the types the reference's UI hands to ``save_exp``."""
    rng = np.random.default_rng(hash_name(f"{cat}/{idx}"))
    h, w = 48, 64
    image = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    mask = ((((xx - 30) / 14.0) ** 2 + ((yy - 22) / 10.0) ** 2) <= 1.0).astype(np.float32)
    depth = (0.5 + 0.3 * xx / w + 0.1 * rng.random((h, w))).astype(np.float32)
    e = dict(image=image, mask=mask if idx == 1 else (mask * 255).astype(np.uint8), depth=depth, depth_vis=depth / depth.max(),
             transform=np.eye(4, dtype=np.float32) + (rng.random((4, 4)).astype(np.float32) - 0.5) * 0.1, h=480 + idx, w=640)
    if cat == "Mix" and idx == 2:
        e["transformed"] = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        e["background"] = rng.random((h, w, 3)).astype(np.float32)          # float RGB in [0, 1]
    return e


# ---- N4: attention-map capture (store_attention_maps) ---------------------------------------------------
STORE_LAYERS = ((16, False, "down"), (16, True, "mid"), (32, False, "up"))     # (S, is_cross, place_in_unet) of one "step"


# ---- loop-level fixture (G18): the reference's text2image_ldm_stable driving a narrow UNet of the SD topology ------------------
LOOP = dict(size=256, steps=6, guidance=3.0, skip_optim=2, optimize_steps=0.65, latent_replace=0.4, lr=0.03, obj_edit_step=0.9,
            self_replace=0.95, cross_replace=0.95, seed=77)


# BASELINE configs[0]: single 256 x 256 image, 2-D translation edit, 20-step DDIM (the reference's own CPU-runnable case)
LOOP_CFG0 = dict(LOOP, steps=20, seed=78)


# BASELINE configs[1] shape: single 512 x 512 image, 3-D rotation edit (a bounded number of DDIM steps: the reference's formulation on CPU
# materialises [10, 4096, 4096] fp32 maps for every 64^2 layer of an optimisation pass)
LOOP_CFG1 = dict(LOOP, size=512, steps=4, seed=79, transform="rotate", amodal_shift=(64, -24))


# SDXL-topology loop case (BASELINE configs[4] topology, narrow model): 512^2 so that the hooked 32^2-token layers evaluate the losses
LOOP_SDXL = dict(LOOP_CFG1, seed=80)


# BASELINE configs[1] at its stated LENGTH: 512^2, 3-D rotation, 50-step DDIM with the batch driver's geometry_editor column
# (large_scale_editor.py:286-299: optimize 0.65 -> 17 optimisation passes, latent_replace 0.1, obj_edit_step 0.9, self / cross 0.95),
# narrow model: pins the step-count-dependent integer gates at T = 50 (int(50*0.95) = 47, int(50*0.9) = 45, the 0.4 T / 0.8 T phases of the
# adaptive schedule, the latent-replace window i < 5)
LOOP_CFG1_T50 = dict(LOOP_CFG1, steps=50, seed=81, latent_replace=0.1)


# BASELINE configs[3] at its stated length: object removal, 768^2 (96^2 / 48^2-token hooked layers), 75 steps, the batch driver's
# geometry_remover column (large_scale_editor.py:199-212,254-262: guidance 5, optimize 0.85 -> 32 optimisation passes, latent_replace 0.4,
# self / cross 0.9, obj_edit_step 1.0; l_eff = lr*(50 - i)*... changes sign at i > 50, U/editor.py:207), narrow model, eps-prediction
# (the reference has no v-prediction path)
LOOP_REM768_T75 = dict(LOOP, size=768, steps=75, seed=82, guidance=5.0, optimize_steps=0.85, latent_replace=0.4,
                       ellipse=dict(cx=354.0, cy=393.0, ax=105.0, ay=87.0))


def loop_inputs(c=None):
    """-> dict(mask [S,S] f32, coords [1,S,S,3] f32, x_T [1,4,S/8,S/8], ddim_latents list of steps+1 [1,4,S/8,S/8])."""
    c = c or LOOP
    size = c["size"]
    if c.get("transform") == "rotate":                     # the 512^2 ellipse and the 3-D rotation about its centroid of the controller cases
        mask = ellipse_mask(size=size)
        coords = coords_rotate_y(mask=mask, size=size)
    else:
        mask = ellipse_mask(size=size, **c.get("ellipse", dict(cx=118.0, cy=131.0, ax=35.0, ay=29.0)))
        coords = coords_translate(dx_px=32.0, dy_px=-12.0, z=0.5, size=size)
    rng = np.random.default_rng(c["seed"])
    traj = [rng.standard_normal((1, 4, size // 8, size // 8)).astype(np.float32) for _ in range(c["steps"] + 1)]
    return dict(mask=mask.astype(np.float32), coords=coords.astype(np.float32), x_T=traj[-1], ddim_latents=traj)       # coords [1,256,256,3]


def mesh_masks(size=64) -> dict:
    """Object masks for the mesh fixtures (G12) and the mesh-coverage parity tests: shapes on which a per-triangle corner test
    (the reference, U/warp_utils.py:331-362) and a per-quad test give different face lists."""
    out = {"ellipse": ellipse_mask(cx=30, cy=33, ax=11, ay=9, size=size)}
    m = np.zeros((size, size), dtype=np.float32)
    m[20:40, 18:44] = 1.0
    m[27:30, 25:28] = 0.0            # hole
    m[20, 30] = 0.0                  # one-pixel notch on the top edge
    m[39, 18] = 0.0                  # missing corner
    out["holes"] = m
    m = np.zeros((size, size), dtype=np.float32)
    for i in range(24):              # diagonal staircase, two pixels thick
        m[16 + i, 14 + i:16 + i + 1] = 1.0
        m[17 + i, 14 + i:16 + i + 1] = 1.0
    m[45:47, 10:50] = 1.0            # thin horizontal bar
    m[10:50, 52] = 1.0               # one-pixel-wide column: no faces
    out["thin"] = m
    return out


def unet_pass_inputs(seed=91, size=32):
    """One UNet pass of the narrow model: latents [2,4,size,size], text context [2,77,64] (G24)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((2, 4, size, size)).astype(np.float32)
    ctx = rng.standard_normal((2, 77, 64)).astype(np.float32)
    return x, ctx


NULL_TEXT = dict(steps=4, inner=3, guidance=3.0, eps=1e-5, seed=83, size=32)


def null_text_inputs():
    """A seeded 'inversion trajectory' of steps + 1 latents [1,4,32,32] for the null-text fixture (G25)."""
    c = NULL_TEXT
    rng = np.random.default_rng(c["seed"])
    x0 = rng.standard_normal((1, 4, c["size"], c["size"])).astype(np.float32)
    # a smooth trajectory (each latent a small step from the previous one), like a real inversion
    traj = [x0]
    for _ in range(c["steps"]):
        traj.append((traj[-1] + 0.05 * rng.standard_normal(x0.shape)).astype(np.float32))
    return traj
