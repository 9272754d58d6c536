"""gd_conv3x3 at the batch sizes of multi-edit batching: every distinct 3x3 convolution call of one UNet pass at batch n — the launcher's
own plan against forced tile shapes / splits (ops.CONV3X3_CFG) and against the library (F.conv2d, MIOpen find mode).
    python tools/conv_batch_probe.py [n=12]"""
import sys, collections, torch, torch.nn.functional as F
sys.path.insert(0, ".")
torch.backends.cudnn.benchmark = True
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd import ops
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pipe, tok, _ = load_model(device="cuda:0", dtype=torch.bfloat16)
unet = pipe.unet; unet.set_attn_processor(VanillaAttentionProcessor())
calls = collections.Counter()
orig = ops.conv3x3
def rec(x, w, bias=None, stride=1, upsample=False, res=None):
    calls[(tuple(x.shape), w.shape[0], stride, bool(upsample), res is not None)] += 1
    return orig(x, w, bias, stride=stride, upsample=upsample, res=res)
ops.conv3x3 = rec
with torch.no_grad():
    emb = pipe.text_encoder(tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids.cuda())[0]
    unet(torch.randn(n, 4, 64, 64, device="cuda", dtype=torch.bfloat16), torch.tensor([500], device="cuda"), encoder_hidden_states=emb.expand(n, -1, -1).contiguous())
ops.conv3x3 = orig

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

tot_def = tot_best = tot_lib = 0.0
print(f"batch {n}: x shape, K, stride, up | calls | default us (TF/s) | best forced cfg us | library us")
for (xs, K, stride, up, has_res), cnt in sorted(calls.items(), key=lambda kv: -kv[1]):
    nb, C, H, W = xs
    x = torch.randn(xs, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device="cuda", dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device="cuda", dtype=torch.bfloat16)
    Ho = 2 * H if up else (H - 1) // stride + 1
    fl = 2.0 * nb * Ho * Ho * K * 9 * C
    ops.CONV3X3_CFG.update(pi=0, ki=0, ksplit=0)
    t_def = timeit(lambda: ops.conv3x3(x, w, b, stride=stride, upsample=up))
    best = (t_def, "default")
    for pi in (1, 2):
        for ki in (1, 2):
            if ki == 2 and K % 128: continue
            for ks in (1, 2, 3, 4):
                ops.CONV3X3_CFG.update(pi=pi, ki=ki, ksplit=ks)
                try:
                    t = timeit(lambda: ops.conv3x3(x, w, b, stride=stride, upsample=up), 10)
                except Exception:
                    continue
                if t < best[0]: best = (t, f"{pi}x{ki} split {ks}")
    ops.CONV3X3_CFG.update(pi=0, ki=0, ksplit=0)
    xl = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    t_lib = timeit(lambda: F.conv2d(xl, w, b, stride=stride, padding=1), 10)
    tot_def += cnt * t_def; tot_best += cnt * best[0]; tot_lib += cnt * min(t_lib, t_def)
    print(f"{xs} K={K} s={stride} up={int(up)} | {cnt:2d} | {t_def:7.1f} ({fl / t_def * 1e-6:6.0f}) | {best[0]:7.1f} {best[1]:14s} | {t_lib:7.1f}", flush=True)
print(f"per pass: default {tot_def / 1e3:.2f} ms, best forced {tot_best / 1e3:.2f} ms, best of (default, library) {tot_lib / 1e3:.2f} ms")
