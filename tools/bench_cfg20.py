"""The CFG pass's 64^2 self-attention launch of an edit (3 batch rows -> 4 token-major segments x 5 heads, fused query warp on one of them,
pre-scaled queries) under each kernel / split choice, with and without the warp tables (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))
dt = torch.bfloat16
N, heads, K = 4096, 5, 15
C = 64 * heads
g = torch.Generator(device="cuda").manual_seed(5)
q = torch.randn(3, N, C, device="cuda", generator=g).to(dt) * 0.2; k = torch.randn(3, N, C, device="cuda", generator=g).to(dt); v = torch.randn(3, N, C, device="cuda", generator=g).to(dt)
# realistic tables: an object of ~12 % of the pixels shifted by a few pixels; K = 15 slots, ~4 filled
S = 64
yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
obj = ((yy - 30) ** 2 + (xx - 28) ** 2 < 13 ** 2)
m = torch.zeros(S, S); m[obj.roll((3, 5), (0, 1))] = 1.0
idx = torch.full((N, K), -1, dtype=torch.int32); w = torch.zeros(N, K)
src = (yy.roll((3, 5), (0, 1)) * S + xx.roll((3, 5), (0, 1))).reshape(-1)
for j in range(4):
    idx[:, j] = torch.where(m.reshape(-1) > 0, (src + j) % N, torch.full((N,), -1)).int(); w[:, j] = 0.25
idx, w, m = idx.cuda(), w.cuda(), m.reshape(-1).cuda().contiguous()
o = [torch.empty_like(q[:1]) for _ in range(4)]

m0 = torch.zeros_like(m)
mfull = torch.ones_like(m)
def launch(warp):
    mm = {1: m, 2: m0, 3: mfull}.get(warp)
    segs = [(q[0:1], k[0:1], v[0:1], o[0], None), (q[1:2], k[1:2], v[1:2], o[1], None),
            (q[1:2], k[1:2], v[1:2], o[2], None, (idx, w, mm)) if warp else (q[1:2], k[1:2], v[1:2], o[2], None),
            (q[2:3], k[1:2], v[1:2], o[3], None)]
    ops.attn_fwd(segs, 0.125, heads=heads, nsplit=1, q_scaled=True)

def t(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

VARS = (("default", -1, 0, 1), ("r02 4x1", 4, 1, 0), ("w64 unsplit", 8, 1, 0), ("w64 split", 8, 1, 2))
res = {}
for rnd in range(5):
    for (nm, qb, ks, sk) in VARS:
        lib.gd_attn_fwd_set_config(qb, ks); lib.gd_attn_fwd_set_even_split(sk)
        for warp in (0, 1, 2, 3):
            res.setdefault((nm, warp), []).append(t(lambda: launch(warp)))
for (nm, _, _, _) in VARS:
    a, b, c, d = (sorted(res[(nm, i)])[2] for i in range(4))
    print(f"{nm:12s} plain {a:6.1f} us   fused warp {b:6.1f} us (+{b - a:4.1f})   tables only (mask all zero) {c:6.1f} (+{c - a:4.1f})   every pixel warped {d:6.1f} (+{d - a:4.1f})", flush=True)
