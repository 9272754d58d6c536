import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load()
DEV = "cuda"; dtype = torch.bfloat16
BH, N, M = 5, 4096, 4096
torch.manual_seed(BH + M)
q = (torch.randn(BH, N, 64, device=DEV) * 1.5).to(dtype); k = (torch.randn(BH, M, 64, device=DEV) * 1.5).to(dtype)
v = torch.randn(BH, M, 64, device=DEV).to(dtype)
k[0, :, 0] += torch.linspace(-40, 40, M, device=DEV).to(dtype); q[0, :, 0] = 8.0
k[1, M // 2, :] *= 6.0
lib.gd_attn_fwd_set_config(8, 1)
res = []
for split in (0, 2, 2, 2, 0):
    lib.gd_attn_fwd_set_even_split(split)
    out = torch.zeros_like(q); lse = torch.zeros(BH, N, device=DEV)
    ops.attn_fwd([(q, k, v, out, lse)], 0.125, nsplit=1)
    torch.cuda.synchronize()
    res.append((out.float(), lse))
for (i, j) in ((1, 2), (2, 3), (0, 4)):
    do = (res[i][0] - res[j][0]).abs(); dl = (res[i][1] - res[j][1]).abs()
    print(f"runs {i},{j}: out equal {bool((do == 0).all())} (max {float(do.max()):.3e}, heads differing {(do.amax(dim=(1,2)) > 0).int().tolist()}), lse equal {bool((dl == 0).all())} (max {float(dl.max()):.3e}, heads {(dl.amax(1) > 0).int().tolist()})")
    if float(do.max()) > 0:
        h = int((do.amax(dim=(1, 2)) > 0).nonzero()[0]); rows = (do[h].amax(1) > 0).nonzero().reshape(-1)
        print("   head", h, "rows differing:", rows.numel(), "first", rows[:12].tolist(), "per 256-row unit:", torch.bincount(rows // 256, minlength=16).tolist())
o1 = res[1][0]
nan = torch.isnan(o1).any(-1)
print("NaN rows per head:", nan.sum(1).tolist())
for h in range(BH):
    if nan[h].any():
        rows = nan[h].nonzero().reshape(-1)
        print(" head", h, "NaN rows per 256-row unit:", torch.bincount(rows // 256, minlength=16).tolist(), "first rows", rows[:8].tolist(), "rows mod 64:", sorted(set((rows % 64).tolist()))[:20])
