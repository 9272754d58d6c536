import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import geodiffuser_amd.unet_sd21 as U
from geodiffuser_amd.diffusion import load_model
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
unet = pipe.unet
x = torch.randn(3, 4, 64, 64, device="cuda").bfloat16(); ctx = torch.randn(3, 77, 1024, device="cuda").bfloat16(); t = torch.tensor([500], device="cuda")
orig = U.GroupNormAct.forward
def dbg(self, x, silu=False):
    fused = (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and self.num_channels // self.num_groups >= 8 and x.is_contiguous(memory_format=torch.channels_last))
    print("GN", tuple(x.shape), x.stride(), "fused" if fused else "torch", flush=True)
    y = orig(self, x, silu)
    torch.cuda.synchronize()
    return y
U.GroupNormAct.forward = dbg
with torch.no_grad():
    y = unet(x, t, encoder_hidden_states=ctx)["sample"]
torch.cuda.synchronize()
print("ok", float(y.float().abs().max()))
