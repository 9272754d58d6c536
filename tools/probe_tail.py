"""Probe (development aid): what a grid-level key split of under-filled / tail launches could cost, emulated with existing kernels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load()
dt = torch.bfloat16

def t(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

CASES = [  # (label, BH, N, M, qb, ks)
    ("5h now (4x2)", 5, 4096, 4096, 4, 2),
    ("5h gs3 emu: 15 x M=1536 (4x1)", 15, 4096, 1536, 4, 1),
    ("5h gs3 emu: 15 x M=1280 (4x1)", 15, 4096, 1280, 4, 1),
    ("5h gs4 emu: 20 x M=1024 (4x1)", 20, 4096, 1024, 4, 1),
    ("5h gs2 emu: 10 x M=2048 (4x2)", 10, 4096, 2048, 4, 2),
    ("5h gs3 emu: 15 x M=1536 (2x2)", 15, 4096, 1536, 2, 2),
    ("15h now (4x1)", 15, 4096, 4096, 4, 1),
    ("16h full (4x1) = 512 WGs", 16, 4096, 4096, 4, 1),
    ("20h now (4x1)", 20, 4096, 4096, 4, 1),
    ("tail: 16 x M=1024 (4x1) = 512 quarter WGs", 16, 4096, 1024, 4, 1),
    ("tail: 8 x M=2048 (4x1) = 256 half WGs", 8, 4096, 2048, 4, 1),
    ("tail: 4h full (4x1) = 128 WGs", 4, 4096, 4096, 4, 1),
    ("tail: 4h (4x2) = 128 WGs of 8 waves", 4, 4096, 4096, 4, 2),
]
res = {c[0]: [] for c in CASES}
bufs = {}
for rnd in range(4):
    for (label, BH, N, M, qb, ks) in CASES:
        key = (BH, N, M)
        if key not in bufs:
            q = (torch.randn(BH, N, 64, device="cuda") * 0.3).to(dt); k = (torch.randn(BH, M, 64, device="cuda") * 1.2).to(dt)
            v = torch.randn(BH, M, 64, device="cuda").to(dt)
            bufs[key] = (q, k, v, torch.empty_like(q))
        q, k, v, o = bufs[key]
        lib.gd_attn_fwd_set_config(qb, ks)
        res[label].append(t(lambda: ops.attn_fwd([(q, k, v, o, None)], 0.125, nsplit=1, q_scaled=True)))
for (label, BH, N, M, qb, ks) in CASES:
    r = sorted(res[label]); med = r[len(r) // 2]
    print(f"{label:48s} {med:7.1f} us  {4.0 * BH * N * M * 64 / med / 1e6:7.1f} TF/s", flush=True)
