"""Are two IDENTICAL batch rows of a UNet pass bit-identical, and are two identical passes?  (development aid)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
tiny = "--tiny" in sys.argv
det = "--det" in sys.argv
bench = "--benchmark" in sys.argv
torch.backends.cudnn.deterministic = det
torch.backends.cudnn.benchmark = bench
dt = torch.bfloat16
p, _, _ = load_model(device="cuda:0", tiny=tiny, dtype=dt)
p.unet.set_attn_processor(VanillaAttentionProcessor())
torch.manual_seed(0)
S = 32 if tiny else 64
x1 = torch.randn(1, 4, S, S, device="cuda").to(dt); c1 = torch.randn(1, 77, 64 if tiny else 1024, device="cuda").to(dt)
x = x1.expand(2, -1, -1, -1).contiguous(); c = c1.expand(2, -1, -1).contiguous()
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
for grad in (False, True):
    outs = []
    for rep in range(3):
        if grad:
            xi = x.detach().float().requires_grad_(True)
            with torch.enable_grad():
                o = p.unet(xi, 500, encoder_hidden_states=c)["sample"]
                (g,) = torch.autograd.grad((o.float() ** 2).sum(), [xi])
            outs.append((o.detach(), g))
        else:
            with torch.no_grad():
                outs.append((p.unet(x, 500, encoder_hidden_states=c)["sample"], None))
    o, g = outs[-1]
    print(f"det={det} benchmark={bench} grad={grad}: rows of one pass identical: {bool(torch.equal(o[0], o[1]))} (rel {rel(o[0], o[1]):.1e}); "
          f"two passes identical: {bool(torch.equal(outs[1][0], outs[2][0]))} (rel {rel(outs[1][0], outs[2][0]):.1e})"
          + (f"; grad rows identical: {bool(torch.equal(g[0], g[1]))} (rel {rel(g[0], g[1]):.1e}); grad passes: {bool(torch.equal(outs[1][1], outs[2][1]))} (rel {rel(outs[1][1], outs[2][1]):.1e})" if grad else ""), flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        if grad:
            xi = x.detach().float().requires_grad_(True)
            with torch.enable_grad():
                o = p.unet(xi, 500, encoder_hidden_states=c)["sample"]
                torch.autograd.grad((o.float() ** 2).sum(), [xi])
        else:
            with torch.no_grad(): p.unet(x, 500, encoder_hidden_states=c)
    torch.cuda.synchronize(); print(f"   {1e3 * (time.perf_counter() - t0) / 10:.1f} ms per pass (eager)", flush=True)
