import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load()
dt = torch.bfloat16
lib.gd_attn_fwd_set_config(8, 1)
BH, N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 5, 4096, 4096
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt); k = torch.randn(BH, M, 64, device="cuda", generator=g).to(dt); v = torch.randn(BH, M, 64, device="cuda", generator=g).to(dt)
if "--adv" in sys.argv:
    k[0, :, 0] += torch.linspace(-40, 40, M, device="cuda").to(dt); q[0, :, 0] = 8.0
    k[1, M // 2, :] *= 6.0
outs = []
for sk in (0, 2, 2, 0):
    lib.gd_attn_fwd_set_even_split(sk)
    o = torch.zeros_like(q); l = torch.zeros(BH, N, device="cuda")
    ops.attn_fwd([(q, k, v, o, l)], 0.125, nsplit=1)
    torch.cuda.synchronize()
    outs.append((o.float(), l))
d = (outs[1][0] - outs[0][0]).abs()
per_unit = d.reshape(BH, N // 256, 256, 64).amax(dim=(2, 3))
print("units wrong:", int((per_unit > 0.05).sum()), "of", per_unit.numel())
print((per_unit > 0.05).int())
# inside a wrong unit: which rows / columns
bad = (per_unit > 0.05).nonzero()
if len(bad):
    b, t = bad[0].tolist()
    blk = d[b, t * 256:(t + 1) * 256]
    print("unit", b, t, "rows wrong per 32-row block:", (blk.amax(1) > 0.05).reshape(8, 32).sum(1).tolist())
    print("cols wrong:", (blk.amax(0) > 0.05).int().tolist())
    r = (blk.amax(1) > 0.05).nonzero()[0].item()
    print("row", r, "split:", outs[1][0][b, t * 256 + r, :8].tolist(), "unsplit:", outs[0][0][b, t * 256 + r, :8].tolist())
print("split run-to-run identical:", torch.equal(outs[1][0], outs[2][0]), " unsplit run-to-run identical:", torch.equal(outs[0][0], outs[3][0]))
d2 = (outs[1][0] - outs[2][0]).abs()
pu = d2.reshape(BH, N // 256, 256, 64).amax(dim=(2, 3))
print("units differing run to run:", int((pu > 0).sum()), "max diff", float(d2.max()))
if float(d2.max()) > 0:
    b, t = (pu > 0).nonzero()[0].tolist()
    blk = d2[b, t * 256:(t + 1) * 256]
    print("unit", b, t, "rows differing per 32-row block:", (blk.amax(1) > 0).reshape(8, 32).sum(1).tolist(), "cols:", (blk.amax(0) > 0).int().tolist())
