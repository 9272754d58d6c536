"""fp8 attention forward vs the 16-bit pipelined kernel at the 64^2 launch shapes (HIP events; development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dev, dt = "cuda", torch.bfloat16
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for (BH, N, M) in ((5, 4096, 4096), (10, 4096, 4096), (15, 4096, 4096), (20, 4096, 4096), (32, 4096, 4096), (20, 1024, 1024)):
    q = (torch.randn(BH, N, 64, device=dev) * 1.2).to(dt); k = (torch.randn(BH, M, 64, device=dev) * 1.2).to(dt); v = torch.randn(BH, M, 64, device=dev).to(dt)
    out = torch.empty_like(q)
    qs = (q.float() * (0.125 * 1.4426950408889634)).to(dt)
    t16 = bench(lambda: ops.attn_fwd([(qs, k, v, out, None)], 0.125, q_scaled=True))
    qz = ops.fp8_quantize(q, k, v, 0.125)
    tq = bench(lambda: ops.fp8_quantize(q, k, v, 0.125))
    t8 = bench(lambda: ops.attn_fwd_fp8(qz, 0.125, out))
    fl = 4.0 * BH * N * M * 64
    print(f"BH={BH:3d} N={N} M={M}: 16-bit (pre-scaled q) {t16:7.1f} us = {fl / t16 / 1e6:7.1f} TF/s | fp8 {t8:7.1f} us = {fl / t8 / 1e6:7.1f} TF/s | quantise q,k,v {tq:6.1f} us")
