"""Micro-benchmark + correctness check of the attention kernels on the SD2.1 shapes (development aid)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dev = "cuda"
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
dtype = torch.bfloat16 if "--fp16" not in sys.argv else torch.float16
torch.manual_seed(0)
for (name, BH, N, M) in (("64^2 cfg 25 heads", 25, 4096, 4096), ("64^2 opt 15 heads", 15, 4096, 4096), ("32^2 cfg 50 heads", 50, 1024, 1024),
                         ("16^2 cfg 100 heads", 100, 256, 256), ("64^2 cross 25 heads", 25, 4096, 77)):
    q = (torch.randn(BH, N, 64, device=dev) * 1.2).to(dtype); k = (torch.randn(BH, M, 64, device=dev) * 1.2).to(dtype); v = torch.randn(BH, M, 64, device=dev).to(dtype)
    out = torch.empty_like(q); lse = torch.empty(BH, N, device=dev)
    t = bench(lambda: ops.attn_fwd([(q, k, v, out, lse)], 0.125))
    fl = 4.0 * BH * N * M * 64
    # correctness on a slice
    s = torch.einsum("bnd,bmd->bnm", q[:2, :256].float(), k[:2].float()) * 0.125
    ref = torch.einsum("bnm,bmd->bnd", torch.softmax(s, -1), v[:2].float())
    err = float((out[:2, :256].float() - ref).abs().max() / ref.abs().max())
    print(f"fwd {name}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s  ({fl/t/2.5e15*100:4.1f}% of peak)  err {err:.1e}", flush=True)
    if M == N and N >= 1024:
        g = (torch.randn(BH, N, 64, device=dev) * 0.1).to(dtype)
        tb = bench(lambda: ops.attn_bwd(q, k, v, out, lse, g, 0.125, False), n=10)
        print(f"bwd_dq {name}: {tb*1e6:8.1f} us  {6.0*BH*N*M*64/tb/1e12:7.1f} TFLOP/s", flush=True)
