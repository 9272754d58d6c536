"""The HBM-bound kernels of the path, each launched a few times at the benchmark's shapes (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE and
--kernel-trace runs, tools/pmc_hbm.sh) — and, stand-alone, the table of their ALGORITHMIC bytes per launch:

    composite (k_composite_tok)   2 f N D 2 + N K 8 + N 4           the stand-alone query warp (normally fused into the attention prologue)
    rasterizer (4-6 kernels)      P 12 + S^2 K 12                    once per resolution per edit; 512^2: the pre-pass (P = 262,144)
    k_splat_weights               2 S^2 K 4 + S^2 K 4
    k_blend_merge / k_blend       (2 or 3) f N D 2 (+ list rows)     merge + blend in one pass
    k_losses_fused                2 f N D 2 (+ f N D 4 amodal target) + the tail's few KB
    k_losses_bwd_rowdot           3 f N D 2 (+ target) + 3 f R M 2 row-dot reads
    k_amodal_fused                f N D 2 + N 4 8 + f N D 4
    k_attn_probs2                 q, k, lse in; f N M 2 + f R M 2 out (write-bound)
    k_edit_dq_fold                kc f N D 4 in, f N D 2 out
    k_attn_fwd (77 keys)          Q + O: 2 B N C 2 (K / V: 77 rows per head)
"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests/golden"): sys.path.insert(0, p)
import cases
from geodiffuser_amd import ops
from geodiffuser_amd._lib import GD_TOKEN_MAJOR, GD_CHANNEL_MAJOR
import torch.nn.functional as F
dev, dt = "cuda", torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S, f, D, K = 64, 5, 64, 15
N = S * S
g = torch.Generator(device=dev).manual_seed(3)
mask = cases.ellipse_mask()
coords = torch.from_numpy(cases.make_coords("rotate", mask)).to(dev)
t64 = F.interpolate(coords.permute(0, 3, 1, 2), size=(S, S), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).half().float()[0].reshape(-1, 3).clone()
t64[:, :2] = -t64[:, :2]
t512 = coords.half().float()[0].reshape(-1, 3).clone(); t512[:, :2] = -t512[:, :2]
q = torch.randn(f, N, D, device=dev, generator=g).to(dt); ro = torch.randn(f, N, D, device=dev, generator=g).to(dt)
eo = torch.randn(f, N, D, device=dev, generator=g).to(dt)
m_edit = (torch.rand(N, device=dev, generator=g) < 0.12).float(); m_wo = 1.0 - m_edit; m_amo = (torch.rand(N, device=dev, generator=g) < 0.2).float()
rows_e = torch.nonzero(m_edit > 0).reshape(-1).to(torch.int32); R_e = 512
pos = torch.full((N,), -1, dtype=torch.int32, device=dev); pos[rows_e.long()] = torch.arange(rows_e.numel(), dtype=torch.int32, device=dev)
act = torch.randn(f, R_e, D, device=dev, generator=g).to(dt)
nn_idx, nn_w, w_dist = ops.nn_table(m_edit, S)
inv5 = torch.full((5,), 1e-5, device=dev); inv_rm = torch.full((1,), 1e-3, device=dev); wv = torch.ones(5, device=dev)
R = 512; nv = torch.tensor([307], dtype=torch.int32, device=dev)
rows = (torch.arange(R, device=dev, dtype=torch.int32) * 7 % N).contiguous()
m_inp = torch.zeros(N, device=dev); m_inp[rows[:307].long()] = 1
inp_pos = torch.full((N,), -1, dtype=torch.int32, device=dev); inp_pos[rows[:307].long()] = torch.arange(307, dtype=torch.int32, device=dev)
k = torch.randn(f, N, D, device=dev, generator=g).to(dt); v = torch.randn(f, N, D, device=dev, generator=g).to(dt)
out = torch.empty_like(q); lse = torch.empty(f, N, device=dev)
ops.attn_fwd([(q, k, v, out, lse)], 0.125)
# cross-attention (77 keys): the CFG pass's form (4 token-major rows x 5 heads at 64^2) and the inversion's (1 row)
C = f * D
qx = torch.randn(4, N, C, device=dev, generator=g).to(dt); kx = torch.randn(4, 77, C, device=dev, generator=g).to(dt); vx = torch.randn(4, 77, C, device=dev, generator=g).to(dt)
ox = torch.empty_like(qx)
idx64, d2 = ops.rasterize_points(t64.contiguous(), S, 1.3 / S * 2.0, K); w64 = ops.splat_weights(idx64, d2, 1.3 / S * 2.0, 2.0, 1.0)
ALG = {
    "k_composite_tok": 2 * f * N * D * 2 + N * K * 8 + N * 4,
    "rasterizer_64 (all its kernels)": N * 12 + N * K * 12,
    "rasterizer_512 (all its kernels)": 262144 * 12 + 262144 * K * 12,
    "k_splat_weights (512^2)": 262144 * K * 12,
    "k_blend_merge (row list 512, eo + out written)": (2 + 2) * f * N * D * 2 + f * R_e * D * 2 + 2 * N * 4,
    "k_blend": 3 * f * N * D * 2 + N * 4,
    "k_amodal_fused": f * N * D * 2 + N * 32 + f * N * D * 4,
    "k_losses_fused (with amodal target)": 2 * f * N * D * 2 + f * N * D * 4 + 4 * N * 4,
    "k_losses_bwd_rowdot": 3 * f * N * D * 2 + f * N * D * 4 + 3 * f * 307 * N * 2,
    "k_attn_probs2 (base map + 512-slot list, 307 live)": f * N * N * 2 + f * 384 * N * 2 + 2 * (f * N * D * 2) * 2,
    "k_edit_dq_fold (3 runs)": 3 * f * N * D * 4 + f * N * D * 2,
    "k_attn_fwd 77 keys, CFG form (4 rows x 5 heads)": 2 * 4 * N * C * 2 + 2 * 4 * 77 * C * 2,
    "k_attn_fwd 77 keys, inversion form (1 row x 5 heads)": 2 * 1 * N * C * 2 + 2 * 77 * C * 2,
}
if len(sys.argv) > 2 and sys.argv[2] == "alg":
    print(json.dumps(ALG, indent=1)); sys.exit(0)
for _ in range(reps):
    ops.splat_composite(q, idx64, w64, m_edit, GD_TOKEN_MAJOR)
    i64, d64 = ops.rasterize_points(t64.contiguous(), S, 1.3 / S * 2.0, K)
    i512, d512 = ops.rasterize_points(t512.contiguous(), 512, 1.3 / 512 * 2.0, K)
    ops.splat_weights(i512, d512, 1.3 / 512 * 2.0, 2.0, 1.0)
    eo2 = torch.empty_like(q); o2 = torch.empty_like(q)
    ops.blend_merge(q, act, pos, ro, m_edit, eo_out=eo2, out=o2)
    ops.blend_tokens(eo, ro, m_edit)
    tgt = ops.amodal_target(eo, nn_idx, nn_w, m_edit, S)
    best = torch.empty(f, R, 2, dtype=torch.int64, device=dev)
    Pb, Pe = ops.attn_probs_pair(q, k, lse, q, k, lse, rows, nv, 0.125, zero=best)
    ops.removal_corr_max_nz(Pe, Pb, m_inp, m_wo, nv, best)
    terms, loss, coefs, rm_coef, aux = ops.edit_losses_fused(eo, ro, tgt, m_wo, m_edit, w_dist, m_amo, S, best, rows, nv, inv5, inv_rm, wv, inv5, True)
    gs = torch.ones(1, device=dev)
    rm_args, rm_ws = ops.removal_bwd_args(Pe, Pb, q, k, rows, aux, m_inp, m_wo, 1.0, gs, rm_coef, 0.125, nv, False)
    dro = ops.edit_losses_bwd_rowdot(eo, ro, tgt, m_wo, m_edit, w_dist, m_amo, None, coefs, gs, True, S, rm_args)
    dq = torch.empty_like(q)
    dk32, kc, part, ws = ops.attn_bwd_nofold(q, k, v, out, lse, dro, 0.125, False, dq)
    ops.removal_bwd_nofold(rm_args, dt)
    ops.edit_dq_fold(part if kc > 1 else None, kc, f, N, D, rm_ws, N, R, inp_pos, aux["wgt"], dq)
    ops.attn_fwd([(qx[0:2], kx[0:2], vx[0:2], ox[0:2], None), (qx[2:3], kx[2:3], vx[2:3], ox[2:3], None, (idx64, w64, m_edit)),
                  (qx[3:4], kx[3:4], vx[2:3], ox[3:4], None)], 0.125, heads=f)
    ops.attn_fwd([(qx[0:1], kx[0:1], vx[0:1], ox[0:1], None)], 0.125, heads=f)
torch.cuda.synchronize()
