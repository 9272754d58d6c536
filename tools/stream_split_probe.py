"""Would one edit fill the chip better as two concurrent half-width streams?  A no-grad UNet pass at batch 3 (one captured graph) against a batch-1 and a
batch-2 pass replayed concurrently on two streams, and against two batch-1 passes side by side (vanilla processors; development aid)."""
import os, sys, time, torch
torch.backends.cudnn.benchmark = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
pipe.unet.set_attn_processor(VanillaAttentionProcessor())
dev = "cuda:0"
def capture(batch, stream):
    x = torch.randn(batch, 4, 64, 64, device=dev, dtype=torch.bfloat16); ctx = torch.randn(batch, 77, 1024, device=dev, dtype=torch.bfloat16)
    t = torch.tensor([500], device=dev)
    with torch.no_grad():
        with torch.cuda.stream(stream):
            for _ in range(3):
                pipe.unet(x, t, encoder_hidden_states=ctx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            out = pipe.unet(x, t, encoder_hidden_states=ctx)["sample"]
    return g, out
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g3, _ = capture(3, s1); g1, _ = capture(1, s1); g2, _ = capture(2, s2); g1b, _ = capture(1, s2)
def run(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
def both(a, b):
    with torch.cuda.stream(s1): a.replay()
    with torch.cuda.stream(s2): b.replay()
def one(a):
    with torch.cuda.stream(s1): a.replay()
print(f"batch 3, one stream: {run(lambda: one(g3)):.2f} ms per pass")
print(f"batch 1, one stream: {run(lambda: one(g1)):.2f} ms;  batch 2, one stream: {run(lambda: one(g2)):.2f} ms")
print(f"batch 1 || batch 2 on two streams: {run(lambda: both(g1, g2)):.2f} ms per pair")
print(f"batch 1 || batch 1 on two streams: {run(lambda: both(g1, g1b)):.2f} ms per pair")
