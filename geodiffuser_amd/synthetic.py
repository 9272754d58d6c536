"""Synthetic edit inputs: image, mask, depth, transform for seed s — no datasets or checkpoints are
reachable from the benchmark environment."""
from __future__ import annotations

import numpy as np
import torch

from .vis_utils import rotateAxis, translateMatrix


def make_edit(seed: int, size: int = 512, kind: str = "rotate"):
    """-> image u8 [S,S,3], depth f32 [S,S], mask f32 [S,S] in {0,1}, transform f32 [4,4]."""
    rng = np.random.default_rng(1234 + seed)
    img = rng.random((size // 8, size // 8, 3), dtype=np.float32)
    img = np.kron(img, np.ones((8, 8, 1), dtype=np.float32))                       # low-pass "content"
    image = (img * 255).astype(np.uint8)
    sc = size / 512.0
    cx, cy = rng.uniform(160, 352, size=2) * sc
    ax, ay = rng.uniform(48, 96, size=2) * sc
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    mask = ((((u - cx) / ax) ** 2 + ((v - cy) / ay) ** 2) <= 1.0).astype(np.float32)
    if kind == "translate":                                                         # the reference's own 2-D example
        depth = np.ones((size, size), dtype=np.float32) * 0.5                       # depth_predictor.py:476-485
        T = translateMatrix(0.1, 0.0, 0.0)
    else:
        depth = np.where(mask > 0.5, 0.5 + 0.2 * (u / size - 0.5), 0.9).astype(np.float32)
        if kind == "rotate":
            T = rotateAxis(float(rng.uniform(-30, 30)), 1).float()
        else:                                                                       # mixed
            T = (translateMatrix(float(rng.uniform(-0.15, 0.15)), float(rng.uniform(-0.15, 0.15)), float(rng.uniform(-0.1, 0.1)))
                 @ rotateAxis(float(rng.uniform(-30, 30)), 1).float())
    return image, depth, mask, T.float()


# batch `geometry_editor` column of SURVEY.md section 5 (large_scale_editor.py:264-299)
EDITOR_KW = dict(cross_replace_steps={"default_": 0.95}, self_replace_steps=0.95, optimize_steps=0.65, lr=0.03, latent_replace=0.1,
                 optimize_embeddings=True, optimize_latents=True, obj_edit_step=0.9, perform_inversion=False, guidance_scale=3.0,
                 skip_optim_steps=2, num_ddim_steps=50, splatting_radius=1.3, edit_type="geometry_editor",
                 splatting_tau=1.0, splatting_points_per_pixel=15, use_adaptive_optimization=True,
                 loss_weights_dict={"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30, "amodal": 80.5},
                                    "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15, "amodal": 3.5}})
REMOVER_KW = dict(EDITOR_KW, self_replace_steps=0.9, cross_replace_steps={"default_": 0.9}, optimize_steps=0.85, latent_replace=0.4,
                  obj_edit_step=1.0, guidance_scale=5.0, edit_type="geometry_remover",
                  loss_weights_dict={"self": {"sim": 55, "removal": 4.6, "smoothness": 30}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15}})


def editor_kwargs(kind="geometry_editor"):
    import copy
    return copy.deepcopy(EDITOR_KW if kind == "geometry_editor" else REMOVER_KW)
