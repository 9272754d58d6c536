import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
BH, N, D = 4, 1024, 64
for M in (77, 1024):
    q = (torch.randn(BH, N, D, device="cuda") * 1.5).to(dt); k = torch.randn(BH, M, D, device="cuda").to(dt); v = torch.randn(BH, M, D, device="cuda").to(dt)
    scale = 0.125; c = scale * 1.4426950408889634; LN2 = 0.6931471805599453
    qp = (q.float() * c).to(dt)
    qe = qp.float() / c                        # the exact numbers the pre-scaled path works on
    s = torch.einsum("bnd,bmd->bnm", qe.double(), k.double()) * scale
    lse_ref = torch.logsumexp(s, -1); P_ref = torch.softmax(s, -1); o_ref = P_ref @ v.double()
    for nm, qq, sc, qs in (("ln2 scale, q_scaled=0", qp, LN2, False), ("q_scaled=1", qp, LN2, True)):
        o = torch.empty_like(qq); lse = torch.empty(BH, N, device="cuda")
        ops.attn_fwd([(qq, k, v, o, lse)], sc, q_scaled=qs)
        P = ops.attn_probs(qq, k, lse, None, sc)[:, :, :M].double()
        print(f"M={M} {nm}: out rel {float((o.double() - o_ref).abs().max() / o_ref.abs().max()):.2e}  lse abs {float((lse.double() - lse_ref).abs().max()):.2e}  "
              f"P rel-L2 {float((P - P_ref).norm() / P_ref.norm()):.2e}  P max-row-sum err {float((P.sum(-1) - 1).abs().max()):.2e}")
    s0 = torch.einsum("bnd,bmd->bnm", q.double(), k.double()) * scale
    o = torch.empty_like(q); lse = torch.empty(BH, N, device="cuda")
    ops.attn_fwd([(q, k, v, o, lse)], scale)
    P = ops.attn_probs(q, k, lse, None, scale)[:, :, :M].double()
    P0 = torch.softmax(s0, -1)
    print(f"M={M} unscaled: lse abs {float((lse.double() - torch.logsumexp(s0, -1)).abs().max()):.2e}  P rel-L2 {float((P - P0).norm() / P0.norm()):.2e}  P max-row-sum err {float((P.sum(-1) - 1).abs().max()):.2e}")
