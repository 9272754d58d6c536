"""Three projections vs one batched GEMM with a stride-0 broadcast input (development aid)."""
import os, sys, torch
dev, dt = "cuda", torch.bfloat16
def bench(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, C) in ((4096, 320), (12288, 320), (1024, 640), (3072, 640), (256, 1280), (768, 1280), (64, 1280)):
    x = torch.randn(M, C, device=dev, dtype=dt); ws = [torch.randn(C, C, device=dev, dtype=dt) for _ in range(3)]
    w3 = torch.stack([w.t() for w in ws], 0).contiguous()
    t3 = bench(lambda: [torch.nn.functional.linear(x, w) for w in ws])
    tb = bench(lambda: torch.bmm(x.unsqueeze(0).expand(3, -1, -1), w3))
    wcat = torch.cat(ws, 0)
    tc = bench(lambda: torch.nn.functional.linear(x, wcat))
    print(f"M={M:6d} C={C:5d}: 3 x linear {t3:6.1f} us | bmm(expand) {tb:6.1f} us | one [3C] GEMM {tc:6.1f} us")
