"""Would the batched reference pass hide behind the first steps of the edit loop?  Captured vanilla UNet passes (full SD2.1-base width, 64^2
latents): main stream = [forward + backward at batch 1, no-grad batch 2, no-grad batch 3, no-grad batch 3] (one optimisation step + the two
plain steps behind it), side stream = one no-grad pass at batch 16 — serial against concurrent.  Development probe (HISTORY 14)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
dt = torch.bfloat16
pipe, tok, sched = load_model(device="cuda:0", dtype=dt)
unet = pipe.unet
unet.set_attn_processor(VanillaAttentionProcessor())
ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
with torch.no_grad():
    emb = pipe.text_encoder(ids.cuda())[0]


def nograd(nb):
    x = torch.randn(nb, 4, 64, 64, device="cuda", dtype=dt); ctx = emb.expand(nb, -1, -1).contiguous(); t = torch.tensor([500] * nb, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            unet(x, t, encoder_hidden_states=ctx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            unet(x, t, encoder_hidden_states=ctx)
    return g


def fwdbwd(nb):
    x = torch.randn(nb, 4, 64, 64, device="cuda", dtype=torch.float32).requires_grad_(True)
    ctx = emb.float().expand(nb, -1, -1).contiguous().requires_grad_(True); t = torch.tensor([500] * nb, device="cuda")
    def run():
        out = unet(x, t, encoder_hidden_states=ctx)["sample"]
        return torch.autograd.grad(out.float().square().mean(), [x, ctx])
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    return g


main_graphs = [fwdbwd(1), nograd(2), nograd(3), nograd(3)]
side_graph = nograd(16)
side = torch.cuda.Stream()


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def serial():
    side_graph.replay()
    for g in main_graphs:
        g.replay()


def concurrent():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        side_graph.replay()
    for g in main_graphs:
        g.replay()
    torch.cuda.current_stream().wait_stream(side)


print(f"main alone: {timed(lambda: [g.replay() for g in main_graphs]):.2f} ms; batch-16 pass alone: {timed(side_graph.replay):.2f} ms")
print(f"serial: {timed(serial):.2f} ms; concurrent (side stream): {timed(concurrent):.2f} ms")
