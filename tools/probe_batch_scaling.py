"""How the UNet's own cost scales with the batch (the ceiling of in-process multi-edit batching): hipGraph-captured no-grad passes at
batch nb (vanilla attention) and forward + backward passes, full SD2.1-base width, 64^2 latents.
    python tools/probe_batch_scaling.py [bf16|fp16]"""
import sys, time
import torch
sys.path.insert(0, ".")
torch.backends.cudnn.benchmark = True
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor

dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
pipe, tok, sched = load_model(device="cuda:0", dtype=dt)
unet = pipe.unet
unet.set_attn_processor(VanillaAttentionProcessor())
for p in unet.parameters():
    p.requires_grad = False
ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
with torch.no_grad():
    emb = pipe.text_encoder(ids.cuda())[0]


def timed(g, n=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("no-grad pass (inversion / CFG shape), captured:")
base = None
for nb in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
    x = torch.randn(nb, 4, 64, 64, device="cuda", dtype=dt)
    ctx = emb.expand(nb, -1, -1).contiguous()
    t = torch.tensor([500], device="cuda")
    with torch.no_grad():
        for _ in range(2):
            unet(x, t, encoder_hidden_states=ctx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = unet(x, t, encoder_hidden_states=ctx)["sample"]
    ms = timed(g)
    base = base or ms
    print(f"  batch {nb:3d}: {ms:8.2f} ms  = {ms / nb:7.2f} ms per row  ({ms / base:5.2f} x batch 1)", flush=True)
    del g, out

print("forward + backward to the latent and the context (optimisation-pass shape, vanilla attention), captured:")
base = None
for nb in (1, 2, 4, 8, 16, 32):
    x = torch.randn(nb, 4, 64, 64, device="cuda", dtype=torch.float32).requires_grad_(True)
    ctx = emb.float().expand(nb, -1, -1).contiguous().requires_grad_(True)
    t = torch.tensor([500], device="cuda")

    def run():
        out = unet(x, t, encoder_hidden_states=ctx)["sample"]
        return torch.autograd.grad(out.float().square().mean(), [x, ctx])

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gr = run()
    ms = timed(g, 10)
    base = base or ms
    print(f"  batch {nb:3d}: {ms:8.2f} ms  = {ms / nb:7.2f} ms per row  ({ms / base:5.2f} x the first)", flush=True)
    del g, gr
