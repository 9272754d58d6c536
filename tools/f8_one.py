"""Launch the fp8 attention forward a few times at N = M = 4096 (for rocprofv3 --pmc runs)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
BH = int(os.environ.get("BH", "32"))
torch.manual_seed(0)
q = (torch.randn(BH, 4096, 64, device="cuda") * 1.2).bfloat16(); k = (torch.randn(BH, 4096, 64, device="cuda") * 1.2).bfloat16(); v = torch.randn(BH, 4096, 64, device="cuda").bfloat16()
out = torch.empty_like(q)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    qz = ops.fp8_quantize(q, k, v, 0.125)
    ops.attn_fwd_fp8(qz, 0.125, out)
torch.cuda.synchronize()
