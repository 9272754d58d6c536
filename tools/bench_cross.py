"""The cross-attention launches of an edit (77 text keys, token-major, pre-scaled queries) per resolution and batch-row count, and the blend that
follows them (development aid): us per launch from back-to-back launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(5)

def t(fn, n=100):
    """us per launch inside a captured graph (as the edit runs them): host time of the binding out of the picture"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for S, heads in ((64, 5), (32, 10), (16, 20), (8, 20)):
    N, C, M = S * S, 64 * heads, 77
    for rows in (2, 3, 4):
        q = torch.randn(rows, N, C, device="cuda", generator=g).to(dt) * 0.2
        k = torch.randn(rows, M, C, device="cuda", generator=g).to(dt); v = torch.randn(rows, M, C, device="cuda", generator=g).to(dt)
        o = torch.empty_like(q)
        one = lambda: ops.attn_fwd([(q, k, v, o, None)], 0.125, heads=heads, q_scaled=True)
        segs = [(q[i:i + 1], k[i:i + 1], v[i:i + 1], o[i:i + 1], None) for i in range(rows)]
        many = lambda: ops.attn_fwd(segs, 0.125, heads=heads, q_scaled=True)
        r = sorted(t(one) for _ in range(5))[2]; r2 = sorted(t(many) for _ in range(5))[2]
        print(f"{S:2d}^2 x {heads:2d} heads x {rows} rows ({rows * heads * ((N + 127) // 128):4d} workgroups): one segment {r:5.1f} us, {rows} segments {r2:5.1f} us", flush=True)
    m = torch.rand(N, device="cuda")
    a = torch.randn(1, N, C, device="cuda", generator=g).to(dt); b = torch.randn(1, N, C, device="cuda", generator=g).to(dt); out = torch.empty_like(a)
    r = sorted(t(lambda: ops.blend_tokens(a, b, m, out=out)) for _ in range(5))[2]
    print(f"{S:2d}^2 blend_tokens {r:5.1f} us", flush=True)

print("\nthe CFG pass's cross-attention layer as an edit issues it: 3 plain batch rows + the edit rows (warped queries) + the replace rows, then the blend;")
print("against gd_attn_fwd_pair (both sides of the edit rows in one workgroup, blend inside)")
for S, heads in ((64, 5), (32, 10), (16, 20), (8, 20)):
    N, C, M, K = S * S, 64 * heads, 77, 15
    q = torch.randn(4, N, C, device="cuda", generator=g).to(dt) * 0.2
    k = torch.randn(4, M, C, device="cuda", generator=g).to(dt); v = torch.randn(4, M, C, device="cuda", generator=g).to(dt)
    m = torch.zeros(N, device="cuda"); m[: N // 8] = 1.0; m[N // 8: N // 6] = 0.5
    idx = torch.full((N, K), -1, dtype=torch.int32, device="cuda"); w = torch.zeros(N, K, device="cuda")
    for j in range(4):
        idx[:, j] = torch.where(m > 0, (torch.arange(N, device="cuda") + 3 * S + 5 + j) % N, torch.full((N,), -1, device="cuda")).int(); w[:, j] = 0.25
    o3, oe, orp, ob = torch.empty_like(q[:3]), torch.empty_like(q[:1]), torch.empty_like(q[:1]), torch.empty_like(q[:1])
    def two():
        ops.attn_fwd([(q[:3], k[:3], v[:3], o3, None), (q[2:3], k[2:3], v[2:3], oe, None, (idx, w, m)), (q[3:], k[3:], v[2:3], orp, None)], 0.125, heads=heads, q_scaled=True)
        ops.blend_tokens(oe, orp, m, out=ob)
    def pair():
        ops.attn_fwd_pair([(q[:3], k[:3], v[:3], o3, None), (q[2:3], k[2:3], v[2:3], ob, None, (idx, w, m))], (q[3:], k[3:], v[2:3]), m, 0.125, heads=heads, q_scaled=True)
    r2 = sorted(t(two, 50) for _ in range(5))[2]; rp = sorted(t(pair, 50) for _ in range(5))[2]
    print(f"{S:2d}^2 x {heads:2d} heads: attention (5 rows) + blend {r2:5.1f} us, pair launch {rp:5.1f} us", flush=True)
