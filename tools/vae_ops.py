"""Kernel table of VAE encode + decode at 512^2 (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
torch.backends.cudnn.benchmark = True
from geodiffuser_amd.diffusion import load_model, image2latent, latent2image
from torch.profiler import profile, ProfilerActivity
import numpy as np, time
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
img = (np.random.rand(512, 512, 3) * 255).astype(np.uint8)
for _ in range(3):
    lat = image2latent(img, pipe); out = latent2image(pipe.vae, torch.cat([lat, lat]).to(torch.bfloat16))
torch.cuda.synchronize()
t0 = time.perf_counter(); lat = image2latent(img, pipe); torch.cuda.synchronize(); t1 = time.perf_counter()
out = latent2image(pipe.vae, torch.cat([lat, lat]).to(torch.bfloat16)); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"encode {1e3*(t1-t0):.1f} ms, decode(2) {1e3*(t2-t1):.1f} ms")
for name, fn in (("encode", lambda: image2latent(img, pipe)), ("decode", lambda: latent2image(pipe.vae, torch.cat([lat, lat]).to(torch.bfloat16)))):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    rows = sorted(((e.self_device_time_total, e.count, e.key) for e in prof.key_averages() if e.self_device_time_total > 0), reverse=True)
    print(name, "device total(double-counted) %.1f ms" % (sum(r[0] for r in rows) / 1e3))
    for t, n, k in rows[:16]:
        print(f"  {t/1e3:8.3f} ms {n:4d}  {k[:110]}")
