"""k_corr_max2's tile variants per inpaint-row count at the 64^2 self-attention shape (5 heads x 4096 x 4096): 24 = 128 x 256 tiles
(eight waves), 22 = 128 x 128 tiles (four waves).  usage: python tools/corr_variants.py [bf16|fp16]"""
import sys, torch
sys.path.insert(0, ".")
from geodiffuser_amd import ops
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
H, N = 5, 4096
g = torch.Generator(device="cuda").manual_seed(0)
Pb = torch.softmax(torch.randn(H, N, N, device="cuda", generator=g) * 2, -1).to(dt)
m_inp = (torch.rand(N, device="cuda", generator=g) < 0.1).float(); m_wo = 1 - m_inp
for R, nv in ((256, 200), (256, 256), (512, 300), (512, 384), (512, 450), (512, 512), (768, 640), (768, 768)):
    Pe = torch.softmax(torch.randn(H, R, N, device="cuda", generator=g) * 2, -1).to(dt)
    rows = torch.randperm(N, device="cuda", generator=g)[:R].to(torch.int32).contiguous()
    nvt = torch.tensor([nv], dtype=torch.int32, device="cuda")
    line = f"R_pad {R:4d} valid {nv:4d}:"
    ref = None
    for var in (24, 22, 1):
        for _ in range(5):
            aux, loss = ops.removal_fwd(Pe, Pb, m_inp, m_wo, rows, 64, n_valid=nvt, variant=var)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            aux, loss = ops.removal_fwd(Pe, Pb, m_inp, m_wo, rows, 64, n_valid=nvt, variant=var)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 30
        fl = 2.0 * H * nv * N * N
        if ref is None:
            ref = aux["j_wo"][:, :nv].clone()
        same = torch.equal(ref, aux["j_wo"][:, :nv])
        line += f"   variant {var:2d}: {us:6.1f} us (corr + reduce) = {fl / us * 1e-6 / 2500:.2f} of peak{'' if same else ' MISMATCH'}"
    print(line, flush=True)
