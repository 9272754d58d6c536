"""Times ONE library variant (GD_LIB=path; tools/build_dbg.sh) of the pipelined attention forward at the benchmark's 64^2 launch
sizes: python tools/bench_dbg.py [8x1]  ->  one line per head count (us, unsplit / even split)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))
dt = torch.bfloat16
cfg = [a for a in sys.argv[1:] if "x" in a]
if cfg: lib.gd_attn_fwd_set_config(*[int(x) for x in cfg[0].split("x")])

def t(fn, n=40):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

row = []
for BH in (5, 15, 20, 32):
    N = 4096
    q = (torch.randn(BH, N, 64, device="cuda") * 0.3).to(dt); k = (torch.randn(BH, N, 64, device="cuda") * 1.2).to(dt); v = torch.randn(BH, N, 64, device="cuda").to(dt)
    o = torch.empty_like(q)
    r = []
    for sk in (0, 11):
        lib.gd_attn_fwd_set_even_split(sk)
        xs = sorted(t(lambda: ops.attn_fwd([(q, k, v, o, None)], 0.125, nsplit=1, q_scaled=True)) for _ in range(5))
        r.append(xs[2])
    row.append(f"BH={BH}: {r[0]:6.1f} / {r[1]:6.1f}")
print(os.path.basename(os.environ.get("GD_LIB", "product")), cfg, " | ".join(row), flush=True)
