"""Micro-benchmark of the warp/splat kernels: achieved HBM GB/s against algorithmic bytes."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests/golden"): sys.path.insert(0, p)
import cases
from geodiffuser_amd import ops
from geodiffuser_amd._lib import GD_TOKEN_MAJOR, GD_CHANNEL_MAJOR
import torch.nn.functional as F
dev = "cuda"
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
mask = cases.ellipse_mask()
coords = torch.from_numpy(cases.make_coords("rotate", mask)).to(dev)
res = {}
for S, f in ((64, 5), (32, 10), (16, 20)):
    N = S * S
    t = F.interpolate(coords.permute(0, 3, 1, 2), size=(S, S), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).half().float()[0].reshape(-1, 3).clone()
    t[:, :2] = -t[:, :2]
    r = 1.3 / S * 2.0
    tr = bench(lambda: ops.rasterize_points(t.contiguous(), S, r, 15), n=20)
    idx, d2 = ops.rasterize_points(t.contiguous(), S, r, 15)
    w = ops.splat_weights(idx, d2, r, 2.0, 1.0)
    q = torch.randn(f, N, 64, device=dev).bfloat16(); m = torch.rand(N, device=dev)
    out = torch.empty_like(q)
    tc = bench(lambda: ops.splat_composite(q, idx, w, m, GD_TOKEN_MAJOR, out=out))
    bytes_alg = 2 * f * N * 64 * 2 + N * 15 * 8 + N * 4
    res[f"composite_{S}"] = dict(us=tc * 1e6, alg_bytes=bytes_alg, GBps=bytes_alg / tc / 1e9)
    res[f"rasterize_{S}"] = dict(us=tr * 1e6, alg_bytes=N * 12 + N * 15 * 12)
    print(f"S={S} f={f}: composite+blend {tc*1e6:7.2f} us  {bytes_alg/1e6:6.2f} MB -> {bytes_alg/tc/1e9:8.1f} GB/s ({bytes_alg/tc/8e12*100:4.1f}% of 8 TB/s);  rasterize(4 kernels + alloc) {tr*1e6:7.1f} us", flush=True)
# 512^2 one-off warps
t = coords.half().float()[0].reshape(-1, 3).clone(); t[:, :2] = -t[:, :2]
r = 1.3 / 512 * 2.0
tr = bench(lambda: ops.rasterize_points(t.contiguous(), 512, r, 15), n=10)
idx, d2 = ops.rasterize_points(t.contiguous(), 512, r, 15); w = ops.splat_weights(idx, d2, r, 2.0, 1.0)
img = torch.rand(1, 3, 512 * 512, device=dev)
tc = bench(lambda: ops.splat_composite(img, idx, w, None, GD_CHANNEL_MAJOR), n=20)
b_r = 262144 * 12 + 262144 * 15 * 12
b_c = 2 * 3 * 262144 * 4 + 262144 * 15 * 8
print(f"512^2: rasterize {tr*1e6:8.1f} us ({b_r/1e6:.1f} MB alg -> {b_r/tr/1e9:7.1f} GB/s);  composite 3ch f32 {tc*1e6:7.1f} us ({b_c/1e6:.1f} MB -> {b_c/tc/1e9:7.1f} GB/s)")
res["rasterize_512"] = dict(us=tr * 1e6, alg_bytes=b_r, GBps=b_r / tr / 1e9); res["composite_512"] = dict(us=tc * 1e6, alg_bytes=b_c, GBps=b_c / tc / 1e9)
print(json.dumps(res))
