import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden", ROOT + "/oracle"): sys.path.insert(0, p)
from geodiffuser_amd.diffusion import load_model, diffusion_step
from geodiffuser_amd import editor
from geodiffuser_amd.synthetic import editor_kwargs, make_edit
pipe, tok, sched = load_model(device="cuda:0", tiny=True, dtype=torch.float16)
torch.manual_seed(0)
x = torch.randn(2, 4, 32, 32, device="cuda").half(); ctx = torch.randn(2, 77, 64, device="cuda").half()
t = torch.tensor([500], device="cuda")
with torch.no_grad():
    a = pipe.unet(x, t, encoder_hidden_states=ctx)["sample"].float()
    b = pipe.unet(x, t, encoder_hidden_states=ctx)["sample"].float()
print("unet fwd repeat max diff", float((a - b).abs().max()), "max", float(a.abs().max()))
# full edit with lr=0, capture intermediate latents per step
def run():
    image, depth, mask, T = make_edit(0, size=256, kind="translate")
    kw = editor_kwargs(); kw.update(lr=0.0, num_ddim_steps=6, ldm_stable_model=pipe, tokenizer_model=tok, scheduler_in=sched, return_latents=True)
    trace = []
    orig = editor.diffusion_step
    def spy(*a, **k):
        out = orig(*a, **k)
        o = out[0] if isinstance(out, tuple) else out
        trace.append(o.detach().float().cpu().clone())
        return out
    editor.diffusion_step = spy
    try:
        images, lat = editor.run_geodiffuser(image, depth, mask, T, **kw)
    finally:
        editor.diffusion_step = orig
    return trace, lat.float().cpu()
t1, l1 = run(); t2, l2 = run()
for i, (u, v) in enumerate(zip(t1, t2)):
    print(i, tuple(u.shape), "row0 diff", float((u[0] - v[0]).abs().max()), "row-1 diff", float((u[-1] - v[-1]).abs().max()), "mag", float(u.abs().max()))
