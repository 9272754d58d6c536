import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
_lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))      # GD_LIB: an A/B build of the library
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
LN2 = 0.6931471805599453
for (H, N, M) in ((5, 4096, 4096), (5, 4096, 77), (10, 1024, 1024), (10, 1024, 77)):
    q = (torch.randn(H, N, 64, device=dev) * 1.2).to(dt); k = (torch.randn(H, M, 64, device=dev) * 1.2).to(dt); v = torch.randn(H, M, 64, device=dev).to(dt)
    g = (torch.randn(H, N, 64, device=dev) * 0.1).to(dt)
    out = torch.empty_like(q); lse = torch.empty(H, N, device=dev)
    ops.attn_fwd([(q, k, v, out, lse)], 0.125)
    t = bench(lambda: ops.attn_bwd(q, k, v, out, lse, g, 0.125, M == 77))
    os.environ["GD_BWD_DQ"] = "1"
    t_old = bench(lambda: ops.attn_bwd(q, k, v, out, lse, g, 0.125, M == 77))
    os.environ.pop("GD_BWD_DQ")
    fl = 6.0 * H * N * M * 64
    if M == N:      # the optimisation pass's form: queries carry scale * log2 e, scale = ln 2 (c == 1: the score chain starts at -lse)
        qp = (q.float() * (0.125 * 1.4426950408889634)).to(dt); outp = torch.empty_like(q); lsep = torch.empty(H, N, device=dev)
        ops.attn_fwd([(qp, k, v, outp, lsep)], LN2)
        tp_ = bench(lambda: ops.attn_bwd(qp, k, v, outp, lsep, g, LN2, False))
        print(f"   pre-scaled queries (scale = ln 2): {tp_:8.1f} us ({fl / tp_ * 1e-6 / 2500:4.2f})")
    print(f"attn_bwd H={H} N={N} M={M} dk={M==77}: {t:8.1f} us = {fl / t * 1e-6:7.0f} TF/s ({fl / t * 1e-6 / 2500:4.2f} of the MFMA peak, dq + fold [+ dk]);  "
          f"k_attn_bwd_dq (GD_BWD_DQ=1): {t_old:8.1f} us ({fl / t_old * 1e-6 / 2500:4.2f})")
    R = N * 3 // 40
    rows = torch.arange(0, R, device=dev, dtype=torch.int32) * 7 % N
    rows = torch.unique(rows).to(torch.int32); R = rows.numel()
    m_inp = torch.zeros(N, device=dev); m_inp[rows.long()] = 1; m_wo = 1 - m_inp
    Pb = ops.attn_probs(q, k, lse, None, 0.125); Pe = ops.attn_probs(q, k, lse, rows, 0.125)
    tp = bench(lambda: ops.attn_probs(q, k, lse, None, 0.125))
    tf = bench(lambda: ops.removal_fwd(Pe, Pb, m_inp, m_wo, rows, int(N ** 0.5)))
    aux, rm = ops.removal_fwd(Pe, Pb, m_inp, m_wo, rows, int(N ** 0.5))
    dq32 = torch.zeros(H, N, 64, device=dev); dk32 = torch.zeros(H, M, 64, device=dev) if M == 77 else None
    tb = bench(lambda: ops.removal_bwd(Pe, Pb, q, k, rows, aux, m_inp, m_wo, 0.01, None, 0.125, dq32, dk32))
    print(f"   R={R}: probs(all rows) {tp:8.1f} us | removal_fwd (corr+reduce, incl. allocs) {tf:8.1f} us | removal_bwd {tb:8.1f} us")
