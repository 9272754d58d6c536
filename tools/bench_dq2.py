"""k_attn_bwd_dq2: time against the number of key runs (GD_DQ2_KC) and heads — fixed cost per workgroup vs cost per key tile (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
N = M = 4096
for H in (1, 2, 5, 10, 16):
    q = (torch.randn(H, N, 64, device=dev) * 1.2).to(dt); k = (torch.randn(H, M, 64, device=dev) * 1.2).to(dt); v = torch.randn(H, M, 64, device=dev).to(dt)
    g = (torch.randn(H, N, 64, device=dev) * 0.1).to(dt)
    out = torch.empty_like(q); lse = torch.empty(H, N, device=dev)
    ops.attn_fwd([(q, k, v, out, lse)], 0.125)
    line = f"H={H:2d} ({32*H} query tiles):"
    for kc in (1, 2, 3, 4, 6, 8, 16):
        os.environ["GD_DQ2_KC"] = str(kc)
        t = bench(lambda: ops.attn_bwd(q, k, v, out, lse, g, 0.125, False))
        line += f"  kc={kc}: {t:6.1f}"
    os.environ.pop("GD_DQ2_KC")
    print(line + "   us (dq2 + fold, incl. host)", flush=True)
