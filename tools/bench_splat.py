"""Micro-benchmark of the warp/splat kernels: achieved HBM GB/s against algorithmic bytes."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests/golden"): sys.path.insert(0, p)
import cases
from geodiffuser_amd import ops
from geodiffuser_amd._lib import GD_TOKEN_MAJOR, GD_CHANNEL_MAJOR
import torch.nn.functional as F
dev = "cuda"
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
mask = cases.ellipse_mask()
coords = torch.from_numpy(cases.make_coords("rotate", mask)).to(dev)
res = {}
for S, f in ((64, 5), (32, 10), (16, 20)):
    N = S * S
    t = F.interpolate(coords.permute(0, 3, 1, 2), size=(S, S), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).half().float()[0].reshape(-1, 3).clone()
    t[:, :2] = -t[:, :2]
    r = 1.3 / S * 2.0
    tr = bench(lambda: ops.rasterize_points(t.contiguous(), S, r, 15), n=20)
    idx, d2 = ops.rasterize_points(t.contiguous(), S, r, 15)
    w = ops.splat_weights(idx, d2, r, 2.0, 1.0)
    q = torch.randn(f, N, 64, device=dev).bfloat16(); m = torch.rand(N, device=dev)
    out = torch.empty_like(q)
    tc = bench(lambda: ops.splat_composite(q, idx, w, m, GD_TOKEN_MAJOR, out=out))
    bytes_alg = 2 * f * N * 64 * 2 + N * 15 * 8 + N * 4
    res[f"composite_{S}"] = dict(us=tc * 1e6, alg_bytes=bytes_alg, GBps=bytes_alg / tc / 1e9)
    res[f"rasterize_{S}"] = dict(us=tr * 1e6, alg_bytes=N * 12 + N * 15 * 12)
    print(f"S={S} f={f}: composite+blend {tc*1e6:7.2f} us  {bytes_alg/1e6:6.2f} MB -> {bytes_alg/tc/1e9:8.1f} GB/s ({bytes_alg/tc/8e12*100:4.1f}% of 8 TB/s);  rasterize(4 kernels + alloc) {tr*1e6:7.1f} us", flush=True)
# 512^2 one-off warps
t = coords.half().float()[0].reshape(-1, 3).clone(); t[:, :2] = -t[:, :2]
r = 1.3 / 512 * 2.0
tr = bench(lambda: ops.rasterize_points(t.contiguous(), 512, r, 15), n=10)
idx, d2 = ops.rasterize_points(t.contiguous(), 512, r, 15); w = ops.splat_weights(idx, d2, r, 2.0, 1.0)
img = torch.rand(1, 3, 512 * 512, device=dev)
tc = bench(lambda: ops.splat_composite(img, idx, w, None, GD_CHANNEL_MAJOR), n=20)
b_r = 262144 * 12 + 262144 * 15 * 12
b_c = 2 * 3 * 262144 * 4 + 262144 * 15 * 8
print(f"512^2: rasterize {tr*1e6:8.1f} us ({b_r/1e6:.1f} MB alg -> {b_r/tr/1e9:7.1f} GB/s);  composite 3ch f32 {tc*1e6:7.1f} us ({b_c/1e6:.1f} MB -> {b_c/tc/1e9:7.1f} GB/s)")
res["rasterize_512"] = dict(us=tr * 1e6, alg_bytes=b_r, GBps=b_r / tr / 1e9); res["composite_512"] = dict(us=tc * 1e6, alg_bytes=b_c, GBps=b_c / tc / 1e9)
print(json.dumps(res))

# ---- the HBM-bound kernels around the hooked layers at the 64^2 level (f = 5 heads x 64, and the token-major 320-channel row): achieved GB/s
print("\n== loss / blend / merge kernels, 64^2, bf16 ==")
S, f, D = 64, 5, 64
N = S * S
eo = torch.randn(f, N, D, device=dev).bfloat16(); ro = torch.randn(f, N, D, device=dev).bfloat16()
m_edit = (torch.rand(N, device=dev) < 0.12).float(); m_wo = 1.0 - m_edit; m_amo = (torch.rand(N, device=dev) < 0.2).float()
t_nn = bench(lambda: ops.nn_table(m_edit, S), n=20)
nn_idx, nn_w, w_dist = ops.nn_table(m_edit, S)
print(f"nn_table (once per edit per resolution): {t_nn*1e6:7.1f} us")
t_am = bench(lambda: ops.amodal_target(eo, nn_idx, nn_w, m_edit, S))
tgt = ops.amodal_target(eo, nn_idx, nn_w, m_edit, S)
b_am = f * N * D * 2 * 2 + N * 4 * 8
print(f"amodal_target (4-NN interpolation + 5x5 gauss): {t_am*1e6:7.1f} us  {b_am/1e6:5.2f} MB -> {b_am/t_am/1e9:7.1f} GB/s")
t_lf = bench(lambda: ops.edit_losses_fwd(eo, ro, tgt, m_wo, m_edit, w_dist, m_amo, S))
b_lf = 3 * f * N * D * 2
print(f"edit_losses_fwd (5 sums over eo / ro / tgt): {t_lf*1e6:7.1f} us  {b_lf/1e6:5.2f} MB -> {b_lf/t_lf/1e9:7.1f} GB/s ({b_lf/t_lf/8e12*100:4.1f} % of 8 TB/s)")
gout = torch.randn(f, N, D, device=dev).bfloat16(); coefs = torch.rand(5, device=dev); gs = torch.ones(1, device=dev)
t_lb = bench(lambda: ops.edit_losses_bwd(eo, ro, tgt, m_wo, m_edit, w_dist, m_amo, gout, coefs, gs, blend=True, S=S))
b_lb = 5 * f * N * D * 2
print(f"edit_losses_bwd (eo, ro, tgt, gout -> dro): {t_lb*1e6:7.1f} us  {b_lb/1e6:5.2f} MB -> {b_lb/t_lb/1e9:7.1f} GB/s ({b_lb/t_lb/8e12*100:4.1f} %)")
out = torch.empty_like(eo)
t_b = bench(lambda: ops.blend_tokens(eo, ro, m_edit, out=out))
b_b = 3 * f * N * D * 2
print(f"blend_tokens: {t_b*1e6:7.1f} us  {b_b/1e6:5.2f} MB -> {b_b/t_b/1e9:7.1f} GB/s ({b_b/t_b/8e12*100:4.1f} %)")
rows = torch.nonzero(m_edit > 0).reshape(-1).to(torch.int32); R = rows.numel(); Rp = -(-R // 256) * 256
pos = torch.full((N,), -1, dtype=torch.int32, device=dev); pos[rows.long()] = torch.arange(R, dtype=torch.int32, device=dev)
act = torch.randn(f, Rp, D, device=dev).bfloat16()
t_m = bench(lambda: ops.rows_merge(eo, act, pos, out=out))
b_m = 2 * f * N * D * 2
print(f"rows_merge: {t_m*1e6:7.1f} us  {b_m/1e6:5.2f} MB -> {b_m/t_m/1e9:7.1f} GB/s ({b_m/t_m/8e12*100:4.1f} %)")
print("(5.2 MB working sets: these launches sit on the ~4-5 us launch floor, not on the HBM roof; an edit issues them inside hipGraphs)")
