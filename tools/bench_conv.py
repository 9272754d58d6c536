"""gd_conv3x3 (conv3x3.hip) on the UNet's 3x3 convolution shapes: correctness against F.conv2d in fp32 on the same 16-bit operands, then
timing against the library convolution (MIOpen, find mode, the committed find-db), both replayed from hipGraphs so that the host's
launch cost does not hide kernel time.  `--sweep` times every tile shape / reduction split per shape (to derive conv_plan's heuristic)."""
import os, sys, itertools, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib, miopen_cache
miopen_cache.configure()
torch.backends.cudnn.benchmark = True
lib = _lib.load()
dt = torch.float16 if "--fp16" in sys.argv else torch.bfloat16
SWEEP = "--sweep" in sys.argv

def mk(n, C, H, K):
    x = torch.randn(n, C, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device="cuda") / (3.0 * C ** 0.5)).to(dt).contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device="cuda").to(dt)
    return x, w, b

def ref(x, w, b, stride, up):
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    return F.conv2d(xi, w.float(), None if b is None else b.float(), stride=stride, padding=1)

print("== correctness (max abs err / max |ref|; bf16 rounding of the result is 4e-3) ==", flush=True)
bad = 0
for (n, C, H, K, stride, up, cfg) in [(1, 64, 8, 64, 1, 0, None), (2, 128, 9, 72, 1, 0, None), (3, 320, 16, 320, 1, 0, None), (1, 64, 7, 128, 2, 0, None),
                                      (2, 128, 8, 64, 2, 0, None), (2, 64, 5, 64, 1, 1, None), (1, 320, 64, 320, 1, 0, None), (3, 1280, 8, 1280, 1, 0, None),
                                      (2, 192, 12, 136, 1, 0, (2, 2, 3)), (2, 192, 12, 136, 1, 0, (1, 1, 5)), (2, 192, 12, 136, 1, 0, (2, 1, 1)),
                                      (2, 192, 12, 136, 1, 0, (1, 2, 27)), (1, 640, 32, 640, 1, 1, None), (1, 640, 32, 640, 2, 0, None)]:
    x, w, b = mk(n, C, H, K)
    for bias in (b, None):
        if cfg: lib.gd_conv3x3_set_config(*cfg)
        o = ops.conv3x3(x, w, bias, stride=stride, upsample=bool(up))
        lib.gd_conv3x3_set_config(0, 0, 0)
        r = ref(x, w, bias, stride, up)
        e = float((o.float() - r).abs().max() / r.abs().max())
        ok = e < (6e-3 if dt == torch.bfloat16 else 1e-3) and o.shape == r.shape and o.is_contiguous(memory_format=torch.channels_last)
        bad += not ok
        print(f"n={n} C={C} H={H} K={K} stride={stride} up={up} cfg={cfg} bias={'y' if bias is not None else 'n'}: rel {e:.2e} {'OK' if ok else 'FAIL'}", flush=True)
x, w, b = mk(2, 320, 32, 640)
o1 = ops.conv3x3(x, w, b); o2 = ops.conv3x3(x, w, b)
print("bit-reproducible:", bool(torch.equal(o1, o2)), flush=True)
if bad: print("CORRECTNESS FAILURES:", bad)

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

SHAPES = [(320, 64, 320), (640, 64, 320), (960, 64, 320), (320, 32, 640), (640, 32, 640), (960, 32, 640), (1280, 32, 640), (1920, 32, 640),
          (640, 16, 1280), (1280, 16, 1280), (1920, 16, 1280), (2560, 16, 1280), (1280, 8, 1280), (2560, 8, 1280)]
EXTRA = [(320, 64, 320, 2, 0), (640, 32, 640, 2, 0), (1280, 16, 1280, 2, 0), (1280, 8, 1280, 1, 1), (1280, 16, 1280, 1, 1), (640, 32, 640, 1, 1)]
print("== timing, us per call from a hipGraph (library = F.conv2d -> MIOpen) ==", flush=True)
tot_lib = tot_own = 0.0
ALL = []
for n in (1, 3):
    for (C, H, K, stride, up) in [(c, h, k, 1, 0) for (c, h, k) in SHAPES] + EXTRA:
        x, w, b = mk(n, C, H, K)
        xi = (lambda: F.interpolate(x, scale_factor=2.0, mode="nearest")) if up else (lambda: x)
        t_lib = graph_time(lambda: F.conv2d(xi(), w, None, stride=stride, padding=1))
        steps = 9 * C // 64
        P = n * (2 * H if up else (H - 1) // stride + 1) ** 2
        gflop = 2.0 * P * K * 9 * C / 1e9
        if SWEEP:
            best = None; rows = []
            for (pi, ki) in ((2, 2), (2, 1), (1, 2), (1, 1)):
                tiles = -(-P // (64 * pi)) * -(-K // (64 * ki))
                for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
                    if sp > 1 and (steps // sp < 3 or tiles * sp > 3072): continue
                    if sp * P * K * 4 > (1 << 30): continue
                    lib.gd_conv3x3_set_config(pi, ki, sp)
                    t = graph_time(lambda: ops.conv3x3(x, w, None, stride=stride, upsample=bool(up)), reps=10, replays=3)
                    rows.append((t, pi, ki, sp, tiles * sp))
            lib.gd_conv3x3_set_config(0, 0, 0)
            ALL.append(dict(n=n, C=C, H=H, K=K, stride=stride, up=up, P=P, steps=steps, lib=t_lib, rows=[list(r) for r in rows]))
            rows.sort()
            t_h = graph_time(lambda: ops.conv3x3(x, w, None, stride=stride, upsample=bool(up)))
            top = "  ".join(f"{pi}x{ki}/{sp}:{t:.1f}" for (t, pi, ki, sp, _) in rows[:4])
            print(f"n={n} C={C:4d} H={H:2d} K={K:4d} s={stride} up={up}: lib {t_lib:6.1f}  heur {t_h:6.1f}  best {top}   ({gflop / rows[0][0] * 1e3:.0f} TF/s)", flush=True)
            tot_own += t_h
        else:
            lib.gd_conv3x3_set_dma(0)
            t_own = graph_time(lambda: ops.conv3x3(x, w, None, stride=stride, upsample=bool(up)))
            lib.gd_conv3x3_set_dma(1)
            o_d = ops.conv3x3(x, w, None, stride=stride, upsample=bool(up))
            t_dma = graph_time(lambda: ops.conv3x3(x, w, None, stride=stride, upsample=bool(up)))
            lib.gd_conv3x3_set_dma(0)
            same = bool(torch.equal(o_d, ops.conv3x3(x, w, None, stride=stride, upsample=bool(up))))
            lib.gd_conv3x3_set_dma(1)
            tot_dma = globals().get("tot_dma", 0.0) + t_dma
            print(f"n={n} C={C:4d} H={H:2d} K={K:4d} s={stride} up={up}: lib {t_lib:6.1f}  own {t_own:6.1f}  x{t_lib / t_own:4.2f}  ({gflop / t_own * 1e3:.0f} TF/s)  dma {t_dma:6.1f} same={same}", flush=True)
            tot_own += t_own
        tot_lib += t_lib
print(f"sum over shapes: library {tot_lib:.0f} us, own {tot_own:.0f} us, dma {globals().get('tot_dma', 0.0):.0f} us", flush=True)
if SWEEP:
    import json
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(ALL, open(os.path.join(ROOT, "gpurun_out", "conv_sweep.json"), "w"))
