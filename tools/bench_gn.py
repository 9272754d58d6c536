"""Fused GroupNorm(+SiLU) timings on the UNet's shapes (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
def t(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, C, H in ((1, 320, 64), (3, 320, 64), (3, 640, 32), (3, 1280, 16), (3, 1280, 8), (3, 2560, 8), (3, 1920, 16), (3, 960, 64), (2, 128, 512), (2, 512, 64)):
    x = torch.randn(B, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.ones(C, device="cuda").bfloat16(); b = torch.zeros(C, device="cuda").bfloat16()
    us = t(lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, True))
    mb = 3 * x.numel() * 2 / 1e6
    print(f"B={B} C={C:4d} H={H:3d}: {us:7.1f} us for the pair  ({mb:6.1f} MB algorithmic -> {mb/us*1e-3*1e3:6.0f} GB/s)", flush=True)
