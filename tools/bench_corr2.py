"""Time of the correlation launch against the number of live workgroups (H x r-tiles sweep) — where does a round end?  (development aid)"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
from geodiffuser_amd._lib import GD_BF16
dev = "cuda"; lib = _lib.load()
def bench(fn, n=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
N = 4096
g = torch.Generator(device=dev).manual_seed(1)
Pb_all = torch.softmax(torch.randn(10, N, N, device=dev, generator=g), -1).bfloat16()
m_inp = (torch.rand(N, device=dev, generator=g) < 0.1).float(); m_wo = 1 - m_inp
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for var in sys.argv[1:] or ["42", "24"]:
    os.environ["GD_CORR_MAX"] = var
    for R, nv in ((128, 128), (256, 256), (384, 384), (512, 307), (512, 384), (512, 512)):
        Pe = torch.softmax(torch.randn(10, R, N, device=dev, generator=g), -1).bfloat16()
        nvt = torch.tensor([nv], dtype=torch.int32, device=dev)
        line = f"[{var}] R_pad={R} n_valid={nv}: "
        for H in (1, 2, 3, 4, 5, 6, 8, 10):
            best = torch.empty(H, R, 2, dtype=torch.int64, device=dev)
            Pe_h = Pe[:H].contiguous(); Pb = Pb_all[:H]
            fn = lambda: lib.gd_removal_corr_max(ctypes.c_void_p(Pe_h.data_ptr()), ctypes.c_void_p(Pb.data_ptr()), ctypes.c_void_p(m_inp.data_ptr()),
                                                 ctypes.c_void_p(m_wo.data_ptr()), ctypes.c_void_p(nvt.data_ptr()), H, R, N, N, ctypes.c_void_p(best.data_ptr()), GD_BF16, st)
            t = bench(fn)
            wgs = (N // 256) * ((nv + 127) // 128) * H
            line += f" H={H}({wgs}wg) {t*1e6:5.1f}"
        print(line, flush=True)
