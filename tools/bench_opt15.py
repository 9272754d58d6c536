"""The optimisation pass's 64^2 self-attention launch (batch 2, 5 heads: reference rows with LSE, compact warped edit rows, edit-vs-reference
keys with LSE; head-major layout, exact scale) under each kernel choice (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))
dt = torch.bfloat16
N, f, K = 4096, 5, 15
g = torch.Generator(device="cuda").manual_seed(5)
q = (torch.randn(2 * f, N, 64, device="cuda", generator=g) * 1.5).to(dt); k = torch.randn(2 * f, N, 64, device="cuda", generator=g).to(dt); v = torch.randn(2 * f, N, 64, device="cuda", generator=g).to(dt)
S = 64
yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
R = int(os.environ.get("ROWS", 256))
obj = ((yy - 30) ** 2 + (xx - 28) ** 2 < (9 if R == 256 else 12) ** 2)
m = torch.zeros(S, S); m[obj.roll((3, 5), (0, 1))] = 1.0
idx = torch.full((N, K), -1, dtype=torch.int32); w = torch.zeros(N, K)
src = (yy.roll((3, 5), (0, 1)) * S + xx.roll((3, 5), (0, 1))).reshape(-1)
for j in range(4):
    idx[:, j] = torch.where(m.reshape(-1) > 0, (src + j) % N, torch.full((N,), -1)).int(); w[:, j] = 0.25
rows = torch.nonzero(m.reshape(-1) > 0).flatten().int()
nv = rows.numel(); assert nv <= R, nv
rows = torch.cat([rows, torch.zeros(R - nv, dtype=torch.int32)]).cuda()
nvt = torch.tensor([nv], dtype=torch.int32, device="cuda")
idx, w, m = idx.cuda(), w.cuda(), m.reshape(-1).cuda().contiguous()
o0 = torch.empty_like(q[:f]); o2 = torch.empty_like(q[:f]); oc = torch.empty(f, R, 64, device="cuda", dtype=dt); od = torch.empty_like(q[:f])
l0 = torch.empty(f, N, device="cuda"); l2 = torch.empty(f, N, device="cuda")

def launch(form):
    lse0, lse2 = (l0, l2) if form & 1 else (None, None)
    segs = [(q[:f], k[:f], v[:f], o0, lse0)]
    if form & 2: segs.append((q[:f], k[:f], v[:f], oc, None, (idx, w, m), (rows, nvt)))
    elif form & 4: segs.append((q[:f], k[:f], v[:f], od, None, (idx, w, m)))
    else: segs.append((q[:f], k[:f], v[:f], od, None))
    segs.append((q[f:], k[:f], v[:f], o2, lse2))
    ops.attn_fwd(segs, 0.125)

def t(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

VARS = (("default", -1, 0, 1), ("mp 4x1", 4, 1, 1), ("mp 4x1 nosk", 4, 1, 0), ("w64 unsplit", 8, 1, 0), ("w64 split", 8, 1, 2))
FORMS = ((0, "3 dense"), (1, "3 dense+lse"), (4, "dense warp"), (5, "dense warp+lse"), (2, "row list"), (3, "row list+lse"))
res = {}
for rnd in range(5):
    for (nm, qb, ks, sk) in VARS:
        lib.gd_attn_fwd_set_config(qb, ks); lib.gd_attn_fwd_set_even_split(sk)
        for fm, _ in FORMS:
            res.setdefault((nm, fm), []).append(t(lambda: launch(fm)))
for (nm, _, _, _) in VARS:
    print(f"{nm:12s} " + "  ".join(f"{lbl} {sorted(res[(nm, fm)])[2]:6.1f}" for fm, lbl in FORMS), flush=True)
