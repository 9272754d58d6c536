"""Time gd_attn_fwd on a few 64^2 launch sizes under the current GD_ATTN_CFG / GD_ATTN_DIAG environment (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
N = M = 4096
tag = f"cfg={os.environ.get('GD_ATTN_CFG','auto')} diag={os.environ.get('GD_ATTN_DIAG','0')}"
for BH in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32").split(",")]:
    q = (torch.randn(BH, N, 64, device="cuda") * 1.2).bfloat16(); k = (torch.randn(BH, M, 64, device="cuda") * 1.2).bfloat16(); v = torch.randn(BH, M, 64, device="cuda").bfloat16()
    o = torch.empty_like(q)
    best = []
    for rnd in range(4):
        for _ in range(3): ops.attn_fwd([(q, k, v, o, None)], 0.125, nsplit=1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): ops.attn_fwd([(q, k, v, o, None)], 0.125, nsplit=1)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 30 * 1e3)
    us = sorted(best)[1]
    print(f"{tag} BH={BH:3d}: {us:7.1f} us  {4.0*BH*N*M*64/us*1e-6:6.0f} TF/s-equivalent", flush=True)
