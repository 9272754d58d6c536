"""1x1 shortcut convolutions of the UNet: F.conv2d (MIOpen) against F.linear on channels_last views (hipBLASLt), from hipGraphs."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import miopen_cache, unet_sd21 as U
miopen_cache.configure(); torch.backends.cudnn.benchmark = True
dt = torch.bfloat16
def graph_time(fn, reps=20, replays=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays) * 1e3
tl = to = 0
for n in (1, 3):
    for (C, H, K) in ((320, 32, 640), (640, 16, 1280), (2560, 8, 1280), (2560, 16, 1280), (1920, 16, 1280), (1920, 32, 640), (1280, 32, 640), (960, 32, 640),
                      (960, 64, 320), (640, 64, 320)):
        x = torch.randn(n, C, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 1, 1, device="cuda") / C ** 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        a = F.conv2d(x, w); b = U.conv1x1(x, w)
        e = float((a.float() - b.float()).abs().max() / a.float().abs().max())
        t0 = graph_time(lambda: F.conv2d(x, w)); t1 = graph_time(lambda: U.conv1x1(x, w))
        tl += t0; to += t1
        print(f"n={n} C={C:4d} H={H:2d} K={K:4d}: conv2d {t0:6.1f}  linear {t1:6.1f}  diff {e:.1e}", flush=True)
print(f"sum: conv2d {tl:.0f} us, linear {to:.0f} us")
