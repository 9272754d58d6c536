"""Scratch GPU probe (development aid): checks the first kernels against torch on the GPU box."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
lib = ctypes.CDLL(os.path.join(ROOT, "geodiffuser_amd/csrc/libgeodiff_hip.so"))
lib.gd_last_error.restype = ctypes.c_char_p
lib.gd_rasterize_workspace_bytes.restype = ctypes.c_size_t
lib.gd_rasterize_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float]

class Seg(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("lse", ctypes.c_void_p), ("bh", ctypes.c_int32), ("pad_", ctypes.c_int32)]

dev = "cuda"
def P(t): return ctypes.c_void_p(t.data_ptr())

def attn(q, k, v, scale, dtype):
    BH, N, D = q.shape; M = k.shape[1]
    out = torch.empty_like(q); lse = torch.empty(BH, N, device=dev, dtype=torch.float32)
    seg = (Seg * 1)(Seg(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), BH, 0))
    rc = lib.gd_attn_fwd(seg, 1, N, M, D, ctypes.c_float(scale), 0 if dtype == torch.float16 else 1, None)
    assert rc == 0, lib.gd_last_error()
    torch.cuda.synchronize()
    return out, lse

def ref_attn(q, k, v, scale):
    s = torch.einsum("bnd,bmd->bnm", q.float(), k.float()) * scale
    p = torch.softmax(s, -1)
    return torch.einsum("bnm,bmd->bnd", p, v.float()), torch.logsumexp(s, -1)

torch.manual_seed(0)
for dtype in (torch.float16, torch.bfloat16):
    for (BH, N, M) in ((2, 128, 64), (3, 256, 256), (2, 1024, 77), (5, 4096, 4096), (2, 100, 77), (1, 64, 64)):
        q = torch.randn(BH, N, 64, device=dev).to(dtype); k = torch.randn(BH, M, 64, device=dev).to(dtype); v = torch.randn(BH, M, 64, device=dev).to(dtype)
        o, lse = attn(q, k, v, 0.125, dtype)
        ro, rl = ref_attn(q, k, v, 0.125)
        e = (o.float() - ro).abs().max().item() / ro.abs().max().item()
        el = (lse - rl).abs().max().item()
        print(f"attn {dtype} BH={BH} N={N} M={M}: rel_err={e:.2e} lse_err={el:.2e}", flush=True)

# probs
for (BH, N, M) in ((2, 256, 256), (2, 1024, 77)):
    dtype = torch.float16
    q = torch.randn(BH, N, 64, device=dev).to(dtype); k = torch.randn(BH, M, 64, device=dev).to(dtype); v = torch.randn(BH, M, 64, device=dev).to(dtype)
    o, lse = attn(q, k, v, 0.125, dtype)
    Mpad = (M + 7) // 8 * 8
    rows = torch.arange(5, N, 3, device=dev, dtype=torch.int32); R = rows.numel()
    Pm = torch.full((BH, R, Mpad), -1, device=dev, dtype=dtype)
    rc = lib.gd_attn_probs(P(q), P(k), P(lse), P(rows), None, BH, N, R, M, Mpad, 64, ctypes.c_float(0.125), P(Pm), 0, None)
    assert rc == 0, lib.gd_last_error()
    torch.cuda.synchronize()
    s = torch.einsum("bnd,bmd->bnm", q.float(), k.float()) * 0.125
    pr = torch.softmax(s, -1)[:, rows.long()]
    print(f"probs N={N} M={M}: err={(Pm[..., :M].float() - pr).abs().max().item():.2e} pad={(Pm[..., M:].float().abs().max().item() if Mpad > M else 0):.1e}", flush=True)

# raster vs oracle
import ref_cpu as O
for S, rpx, K in ((32, 1.3, 15), (64, 1.3, 15), (16, 2.7, 4)):
    rng = np.random.default_rng(S)
    Pn = S * S
    pts = rng.uniform(-1.1, 1.1, size=(1, Pn, 3)).astype(np.float32)
    pts[..., 2] = np.round(rng.uniform(-0.05, 1.0, size=(1, Pn)) * 16) / 16
    r = rpx / S * 2.0
    ri, rz, rd = O.rasterize_points(torch.from_numpy(pts), S, r, K)
    dp = torch.from_numpy(pts[0]).to(dev)
    idx = torch.empty(S, S, K, dtype=torch.int32, device=dev); zb = torch.empty(S, S, K, device=dev); d2 = torch.empty(S, S, K, device=dev)
    wsb = lib.gd_rasterize_workspace_bytes(Pn, S, ctypes.c_float(r))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = lib.gd_rasterize_points(P(dp), Pn, S, ctypes.c_float(r), K, P(idx), P(zb), P(d2), P(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, lib.gd_last_error()
    torch.cuda.synchronize()
    print(f"raster S={S}: idx_equal={torch.equal(idx.cpu(), ri[0])} z_equal={torch.equal(zb.cpu(), rz[0])} d2_equal={torch.equal(d2.cpu(), rd[0])}", flush=True)

# timing of the big attention
q = torch.randn(25, 4096, 64, device=dev).half(); k = torch.randn(25, 4096, 64, device=dev).half(); v = torch.randn(25, 4096, 64, device=dev).half()
for _ in range(3): attn(q, k, v, 0.125, torch.float16)
t0 = time.time(); n = 20
out = torch.empty_like(q); lse = torch.empty(25, 4096, device=dev)
seg = (Seg * 1)(Seg(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), 25, 0))
torch.cuda.synchronize(); t0 = time.time()
for _ in range(n): lib.gd_attn_fwd(seg, 1, 4096, 4096, 64, ctypes.c_float(0.125), 0, None)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
fl = 4 * 25 * 4096 * 4096 * 64
print(f"attn fwd 25x4096x4096x64: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
