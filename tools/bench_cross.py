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
