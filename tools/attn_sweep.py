import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
N = M = 4096
for BH in (2, 4, 6, 8, 10, 12, 14, 15, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32, 40, 48, 64):
    q = (torch.randn(BH, N, 64, device="cuda") * 1.2).bfloat16(); k = (torch.randn(BH, M, 64, device="cuda") * 1.2).bfloat16(); v = torch.randn(BH, M, 64, device="cuda").bfloat16()
    out = torch.empty_like(q)
    t = bench(lambda: ops.attn_fwd([(q, k, v, out, None)], 0.125))
    print(f"BH={BH:3d} WGs={BH*32:5d} ({BH*32/256:5.2f}/CU)  {t*1e6:7.1f} us  {4.0*BH*N*M*64/t/1e12:7.1f} TF/s", flush=True)
