"""Fused query warp vs the two-launch path on the launch shapes of an edit (development aid): CFG pass (token-major, 3 batch rows:
vanilla 2 rows + edit_out + replace = 20 heads) and optimisation pass (head-major: vanilla 5 + edit_out 5 + replace 5 heads)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
from geodiffuser_amd._lib import GD_TOKEN_MAJOR
dt = torch.bfloat16
N, K, H = 4096, 15, 5
torch.manual_seed(0)
def t(fn, n=30):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
idx = ((torch.arange(N, device="cuda")[:, None] + torch.randint(-70, 70, (N, K), device="cuda")) % N).to(torch.int32)
idx[:, 4:] = -1
w = (torch.rand(N, K, device="cuda") * 0.25).contiguous(); yy, xx = torch.meshgrid(torch.arange(64, device="cuda"), torch.arange(64, device="cuda"), indexing="ij")
m = ((((xx - 36) / 11.0) ** 2 + ((yy - 30) / 9.0) ** 2) <= 1.0).float().reshape(-1).contiguous()      # compact object mask, ~8 % of the map
for name, heads, qs in (("CFG pass, token-major, q_scaled", H, True), ("optimisation pass, head-major, exact", 0, False)):
    C = 64 * (heads if heads else 1)
    mk = lambda b: (torch.randn(b, N, C, device="cuda") * (0.18 if qs else 1.0)).to(dt)
    mkk = lambda b: torch.randn(b, N, C, device="cuda").to(dt)
    nb = 1 if heads else H                      # batch rows per "sample" in this layout
    q_van, k_van, v_van = mk(2 * nb if heads else nb), mkk(2 * nb if heads else nb), mkk(2 * nb if heads else nb)
    q_base, k_base, v_base, q_edit = mk(nb), mkk(nb), mkk(nb), mk(nb)
    o_van, o_e, o_r = torch.empty_like(q_van), torch.empty_like(q_base), torch.empty_like(q_base)
    def fused():
        ops.attn_fwd([(q_van, k_van, v_van, o_van, None), (q_base, k_base, v_base, o_e, None, (idx, w, m)), (q_edit, k_base, v_base, o_r, None)],
                     0.125, heads=heads, q_scaled=qs)
    def two():
        qw = ops.splat_composite(q_base, idx, w, m, GD_TOKEN_MAJOR)
        ops.attn_fwd([(q_van, k_van, v_van, o_van, None), (qw, k_base, v_base, o_e, None), (q_edit, k_base, v_base, o_r, None)],
                     0.125, heads=heads, q_scaled=qs)
    def plain():
        ops.attn_fwd([(q_van, k_van, v_van, o_van, None), (q_base, k_base, v_base, o_e, None), (q_edit, k_base, v_base, o_r, None)],
                     0.125, heads=heads, q_scaled=qs)
    comp = t(lambda: ops.splat_composite(q_base, idx, w, m, GD_TOKEN_MAJOR))
    r = [t(fused), t(two), t(plain)]; r2 = [t(fused), t(two), t(plain)]
    print(f"{name}: fused {min(r[0], r2[0]):6.1f} us | composite + attention {min(r[1], r2[1]):6.1f} us (composite alone {comp:5.1f}) | attention without warp {min(r[2], r2[2]):6.1f} us", flush=True)
