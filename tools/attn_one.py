"""Launch the 64^2 forward attention a few times (for rocprofv3 --pmc runs)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
BH, N, M = int(os.environ.get("BH", "32")), 4096, 4096
torch.manual_seed(0)
q = (torch.randn(BH, N, 64, device="cuda") * 1.2).bfloat16(); k = (torch.randn(BH, M, 64, device="cuda") * 1.2).bfloat16(); v = torch.randn(BH, M, 64, device="cuda").bfloat16()
out = torch.empty_like(q); lse = torch.empty(BH, N, device="cuda")
QS = os.environ.get("QS", "0") == "1"
if QS:
    q = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.attn_fwd([(q, k, v, out, lse if not QS else None)], 0.125, q_scaled=QS)
torch.cuda.synchronize()
