"""Launch gd_conv3x3 a few times on one UNet shape (for rocprofv3 --pmc runs): CONV_SHAPE = n,C,H,K (default 3,1920,32,640)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
n, C, H, K = (int(v) for v in os.environ.get("CONV_SHAPE", "3,1920,32,640").split(","))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dt = torch.bfloat16
torch.manual_seed(0)
x = torch.randn(n, C, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
w = (torch.randn(K, C, 3, 3, device="cuda") * 0.02).to(dt).contiguous(memory_format=torch.channels_last)
for _ in range(reps):
    y = ops.conv3x3(x, w)
torch.cuda.synchronize()
