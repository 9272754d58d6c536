"""Row sums on the matrix pipe (k_attn_fwd_w64 LSUM) against the round-3 vector-pipe sums: run once with GD_ATTN_LSUM=0 and once with 1.
Launch forms of an edit at 64^2: the optimisation pass's (3 segments, row list, LSE; exact-scale rescue variant vs pre-scaled), the CFG pass's
(4 token-major rows x 5 heads, fused warp + row list), 15 plain heads, the 5-head inversion launch.  Also the error against an fp32 softmax."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dt = torch.bfloat16; dev = "cuda"
N, f, K = 4096, 5, 15
g = torch.Generator(device=dev).manual_seed(5)
C = 0.125 * 1.4426950408889634
q = (torch.randn(4 * f, N, 64, device=dev, generator=g) * 1.5); k = torch.randn(4 * f, N, 64, device=dev, generator=g); v = torch.randn(4 * f, N, 64, device=dev, generator=g)
qs = (q * C).to(dt); q16 = q.to(dt); k16 = k.to(dt); v16 = v.to(dt)
m = (torch.rand(N, device=dev, generator=g) < 0.08).float()
idx = torch.randint(-1, N, (N, K), device=dev, dtype=torch.int32, generator=g); w = torch.rand(N, K, device=dev, generator=g) * 0.2
rows = torch.nonzero(m > 0).reshape(-1).to(torch.int32); nv = rows.numel(); R = 512
rows = torch.cat([rows, torch.zeros(R - nv, dtype=torch.int32, device=dev)]).contiguous(); nvt = torch.tensor([nv], dtype=torch.int32, device=dev)
def t(fn, n=60):
    for _ in range(6): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
o = [torch.empty(f, N, 64, device=dev, dtype=dt) for _ in range(3)]; oc = torch.empty(f, R, 64, device=dev, dtype=dt)
l0 = torch.empty(f, N, device=dev); l2 = torch.empty(f, N, device=dev)
def opt(qq, scaled):
    segs = [(qq[:f], k16[:f], v16[:f], o[0], l0), (qq[:f], k16[:f], v16[:f], oc, None, (idx, w, m), (rows, nvt)), (qq[f:2 * f], k16[:f], v16[:f], o[2], l2)]
    ops.attn_fwd(segs, 0.6931471805599453 if scaled is not None else 0.125, q_scaled=bool(scaled))
# token-major CFG form
qt = qs[:4 * f].reshape(4, f, N, 64).permute(0, 2, 1, 3).reshape(4, N, f * 64).contiguous(); kt = k16[:4 * f].reshape(4, f, N, 64).permute(0, 2, 1, 3).reshape(4, N, f * 64).contiguous()
vt = v16[:4 * f].reshape(4, f, N, 64).permute(0, 2, 1, 3).reshape(4, N, f * 64).contiguous(); ot = torch.empty_like(qt); oct_ = torch.empty(1, R, f * 64, device=dev, dtype=dt)
def cfg():
    ops.attn_fwd([(qt[0:2], kt[0:2], vt[0:2], ot[0:2], None), (qt[1:2], kt[1:2], vt[1:2], oct_, None, (idx, w, m), (rows, nvt)),
                  (qt[2:3], kt[1:2], vt[1:2], ot[3:4], None)], 0.125, heads=f, q_scaled=True)
o15 = torch.empty(3 * f, N, 64, device=dev, dtype=dt); o5 = torch.empty(f, N, 64, device=dev, dtype=dt)
res = {}
for rnd in range(3):
    for nm, fn in (("opt pass, exact scale (rescue variant)", lambda: opt(q16, None)), ("opt pass, pre-scaled queries told to the kernel", lambda: opt(qs, True)),
                   ("CFG pass (4 rows x 5 heads, warp + row list)", cfg), ("15 plain heads, pre-scaled", lambda: ops.attn_fwd([(qs[:3 * f], k16[:3 * f], v16[:3 * f], o15, None)], 0.125, q_scaled=True)),
                   ("5 heads (inversion), pre-scaled", lambda: ops.attn_fwd([(qs[:f], k16[:f], v16[:f], o5, None)], 0.125, q_scaled=True))):
        res.setdefault(nm, []).append(t(fn))
print("GD_ATTN_LSUM =", os.environ.get("GD_ATTN_LSUM", "(default 1)"))
for nm, v_ in res.items():
    print(f"  {nm:52s} {sorted(v_)[1]:6.1f} us")
# accuracy of the pre-scaled launch (head 0) against an fp32 softmax of the same 16-bit inputs
ops.attn_fwd([(qs[:f], k16[:f], v16[:f], o5, l0)], 0.125, q_scaled=True)
s = (qs[0].float() @ k16[0].float().t()) * 0.6931471805599453
ref = torch.softmax(s, -1) @ v16[0].float()
print("  pre-scaled 5-head launch vs fp32: out max rel %.2e, lse max abs %.2e" % (float((o5[0].float() - ref).abs().max() / ref.abs().max()), float((l0[0] - torch.logsumexp(s, -1)).abs().max())))
