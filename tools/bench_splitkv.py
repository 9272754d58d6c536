"""Split-KV on the under-filled 64^2 / 32^2 launches of an edit (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for BH, N in ((5, 4096), (10, 4096), (15, 4096), (20, 4096), (25, 4096), (10, 1024), (30, 1024), (50, 1024)):
    q = torch.randn(BH, N, 64, device="cuda").bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q); o = torch.empty_like(q)
    row = []
    for ns in (1, 2, 3, 4, 6, 8):
        if ns > N // 64 // 4: continue
        us = t(lambda: ops.attn_fwd([(q, k, v, o, None)], 0.125, nsplit=ns))
        row.append(f"ns={ns}: {us:6.1f} us {4.0*BH*N*N*64/us*1e-6:6.0f} TF/s")
    print(f"BH={BH:3d} N={N}: " + " | ".join(row), flush=True)
