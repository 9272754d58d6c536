"""Which kernels does ONE no-grad UNet pass launch?  (torch.profiler on an eager pass with the vanilla processor; development aid)"""
import os, sys
os.environ["GD_GRAPHS"] = "0"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd.diffusion import load_model
from torch.profiler import profile, ProfilerActivity
B = int(os.environ.get("B", "3"))
p, tok, _ = load_model(device="cuda:0", dtype=torch.bfloat16)
x = torch.randn(B, 4, 64, 64, device="cuda").bfloat16()
ctx = p.text_encoder(tok([""] * B).input_ids.to("cuda"))[0]
with torch.no_grad():
    for _ in range(3):
        p.unet(x, 500, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        p.unet(x, 500, encoder_hidden_states=ctx)
        torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
ev.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ev); n = sum(e.count for e in ev)
print(f"batch {B}: {n} kernels, {tot / 1e3:.2f} ms of kernel time")
for e in ev[:45]:
    print(f"{e.count:5d} x {e.device_time_total / e.count:7.1f} us = {e.device_time_total / 1e3:7.3f} ms  {e.key[:110]}")
