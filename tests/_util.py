"""Shared helpers for the parity tests (inputs regenerated from seeds, see tests/golden/cases.py)."""
from __future__ import annotations

import os

import numpy as np
import torch

import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def warped_mask(kind: str) -> torch.Tensor:
    """[2,1,512,512] binarised 512^2 warp of the object mask (bit-packed in G0)."""
    bits = load("G0_warped_mask_512")[kind]
    m = np.unpackbits(bits)[: 512 * 512].reshape(512, 512).astype(np.float32)
    return torch.from_numpy(m)[None, None].tile(2, 1, 1, 1)


def rel_err(a, b) -> float:
    """max|a-b| / max|b| — the tolerance metric used by every floating-point parity test."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def case_inputs(case):
    S, f, D = case["S"], case["f"], case["D"]
    N = S * S
    M = 77 if case["cross"] else N
    B = 4 if case["cfg"] else 2
    q, k, v = (torch.from_numpy(a) for a in cases.make_qkv(case["seed"], B, f, N, M, D))
    mask = cases.ellipse_mask()
    coords = torch.from_numpy(cases.make_coords(case["coords"], mask))
    return q, k, v, mask, coords


def case_gout(case, shape):
    return torch.from_numpy(np.random.default_rng(case["seed"] + 1000).standard_normal(tuple(shape), dtype=np.float32)) * 0.01
