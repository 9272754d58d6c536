"""One captured no-grad UNet pass at batch NB (vanilla attention, full SD2.1-base width, 64^2 latents), replayed REPS times: ms per pass.
Under `rocprofv3 --kernel-trace --stats` the per-kernel totals of two batch sizes can be compared (tools/probe_pass.sh).
    NB=4 REPS=30 python tools/probe_pass.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
nb, reps = int(os.environ.get("NB", "3")), int(os.environ.get("REPS", "30"))
dt = torch.bfloat16
pipe, tok, sched = load_model(device="cuda:0", dtype=dt)
unet = pipe.unet
unet.set_attn_processor(VanillaAttentionProcessor())
ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
with torch.no_grad():
    emb = pipe.text_encoder(ids.cuda())[0]
    x = torch.randn(nb, 4, 64, 64, device="cuda", dtype=dt); ctx = emb.expand(nb, -1, -1).contiguous(); t = torch.tensor([500] * nb, device="cuda")
    for _ in range(2):
        unet(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = unet(x, t, encoder_hidden_states=ctx)["sample"]
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.replay()
torch.cuda.synchronize()
print(f"batch {nb}: {1e3 * (time.perf_counter() - t0) / reps:.2f} ms per captured no-grad pass ({reps} replays)")
