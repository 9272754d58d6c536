import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get('GD_LIB', _lib.LIB_PATH))
dt = torch.bfloat16
for (BH, N, M, qs_, ks_) in ((8, 1024, 1024, 1.5, 1.5), (8, 1024, 1024, 3.0, 3.0), (4, 4096, 4096, 3.0, 3.0), (8, 1024, 1024, 0.5, 0.5)):
    g = torch.Generator(device="cuda").manual_seed(1)
    q = (torch.randn(BH, N, 64, device="cuda", generator=g) * qs_).to(dt); k = (torch.randn(BH, M, 64, device="cuda", generator=g) * ks_).to(dt)
    v = torch.randn(BH, M, 64, device="cuda", generator=g).to(dt)
    s = torch.einsum("bnd,bmd->bnm", q.double(), k.double()) * 0.125
    ref = torch.einsum("bnm,bmd->bnd", torch.softmax(s, -1), v.double())
    tile0 = s[:, :, :64].amax(-1); gap = (s.amax(-1) - tile0) / 0.6931
    line = f"BH={BH} N={N} q*{qs_}: max-vs-tile0 gap (log2) mean {float(gap.mean()):.1f} max {float(gap.max()):.1f} |"
    for cfg in ((4, 1), (8, 1)):
        lib.gd_attn_fwd_set_config(*cfg)
        o = torch.zeros_like(q); ops.attn_fwd([(q, k, v, o, None)], 0.125, nsplit=1); torch.cuda.synchronize()
        e = (o.double() - ref).abs()
        line += f" {cfg}: max {float(e.max()):.4f} mean {float(e.mean()):.6f} rms {float((e**2).mean().sqrt()):.6f} |"
    print(line)
