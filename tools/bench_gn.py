"""GroupNorm (+SiLU) at the UNet's shapes, inside a hipGraph: us per call (stats + apply) and effective bandwidth (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load()
dev, dt = "cuda", torch.bfloat16
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0
# (B, C, H, count per CFG pass) — the SD2.1 UNet's GroupNorm census at batch 3
for (B, C, H, cnt) in ((1, 320, 64, 0), (1, 640, 32, 0), (1, 1280, 16, 0), (1, 1280, 8, 0), (3, 320, 64, 13), (3, 640, 64, 2), (3, 960, 64, 1), (3, 320, 32, 1), (3, 640, 32, 9), (3, 960, 32, 1), (3, 1280, 32, 2), (3, 1920, 32, 1),
                       (3, 640, 16, 1), (3, 1280, 16, 9), (3, 1920, 16, 1), (3, 2560, 16, 2), (3, 1280, 8, 11), (3, 2560, 8, 6), (1, 320, 64, 0)):
    x = torch.randn(B, C, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    g = torch.ones(C, device=dev, dtype=dt); b = torch.zeros(C, device=dev, dtype=dt)
    lib.gd_group_norm_set_single_launch(0)
    t2 = bench(lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, True))
    lib.gd_group_norm_set_single_launch(1)
    t = bench(lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, True))
    mb = x.numel() * 2 * 3 / 1e6
    tot += t * cnt
    print(f"B={B} C={C:5d} H={H:3d}: two launches {t2:6.1f} us, now {t:6.1f} us  ({mb / t * 1e-3 * 1e3:6.0f} GB/s over 2 reads + 1 write)  x{cnt}")
print(f"weighted per CFG pass: {tot / 1e3:.2f} ms")
