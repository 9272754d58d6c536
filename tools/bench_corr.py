"""k_corr_max variants on the 64^2 / 32^2 self-attention shapes of an optimisation pass: bit-identity against the register-staged kernel
and achieved FLOP rate (algorithmic = 2 H n_valid N M; executed = 2 H ceil64(n_valid) N M).  Development aid (run on the GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dev = "cuda"
def bench(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for dtype in (torch.bfloat16, torch.float16):
    for H, N, R, nvs in ((5, 4096, 256, (200, 256)), (5, 4096, 512, (307, 384, 450, 512)), (5, 4096, 768, (640,)), (10, 1024, 64, (40, 64)), (10, 1024, 128, (100,))):
        g = torch.Generator(device=dev).manual_seed(1)
        Pb = torch.softmax(torch.randn(H, N, N, device=dev, generator=g), -1).to(dtype)
        Pe = torch.softmax(torch.randn(H, R, N, device=dev, generator=g), -1).to(dtype)
        m_inp = (torch.rand(N, device=dev, generator=g) < 0.1).float(); m_wo = 1 - m_inp
        rows = torch.randperm(N, device=dev, generator=g)[:R].to(torch.int32).contiguous()
        S = int(N ** 0.5)
        for nv in nvs:
            nvt = torch.tensor([nv], dtype=torch.int32, device=dev)
            ref = None
            line = f"{str(dtype)[6:]:9s} H={H} N={N} R_pad={R} n_valid={nv}:"
            for var in ("0", "42", "24", "22"):
                os.environ["GD_CORR_MAX"] = var
                aux, loss = ops.removal_fwd(Pe, Pb, m_inp, m_wo, rows, S, n_valid=nvt)
                cur = {k: v[:, :nv].clone() for k, v in aux.items()}
                if ref is None: ref = cur
                same = all(torch.equal(cur[k], ref[k]) for k in cur)
                # time the correlation launch alone (zero fill + k_corr_max*)
                import ctypes
                from geodiffuser_amd import _lib
                lib = _lib.load(); best = torch.empty(H, R, 2, dtype=torch.int64, device=dev)
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                dt = 1 if dtype == torch.float16 else 2
                from geodiffuser_amd._lib import GD_F16, GD_BF16
                dt = GD_F16 if dtype == torch.float16 else GD_BF16
                fn = lambda: lib.gd_removal_corr_max(ctypes.c_void_p(Pe.data_ptr()), ctypes.c_void_p(Pb.data_ptr()), ctypes.c_void_p(m_inp.data_ptr()),
                                                     ctypes.c_void_p(m_wo.data_ptr()), ctypes.c_void_p(nvt.data_ptr()), H, R, N, N, ctypes.c_void_p(best.data_ptr()), dt, st)
                t = bench(fn)
                alg = 2.0 * H * nv * N * N
                line += f"  [{var}] {t*1e6:6.1f} us {alg/t/1e12:6.0f} TF/s ({alg/t/2.5e15:4.2f}){'' if same else ' MISMATCH'}"
            print(line, flush=True)
os.environ.pop("GD_CORR_MAX", None)
