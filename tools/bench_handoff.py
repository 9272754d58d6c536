"""What the even split's hand-off costs: forced split with and without the merge (mode 12: partial tiles written, no ticket / merge —
wrong results, timing only) against the unsplit launch, 5 and 20 heads at 64^2, pre-scaled queries (development aid)."""
import os as _os; _os.environ.setdefault('GD_ATTN_DEV_MODES', '1')  # development hand-off modes 10-12 of gd_attn_fwd_set_even_split
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))
dt = torch.bfloat16
N = 4096
def t(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for BH in (5, 10, 20):
    g = torch.Generator(device="cuda").manual_seed(BH)
    q = (torch.randn(BH, N, 64, device="cuda", generator=g) * 0.18).to(dt); k = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt); v = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt)
    o = torch.empty_like(q)
    res = {}
    VARS = [("mp 4x1", 4, 1), ("w64", 8, 1)]
    for rnd in range(5):
        for nm, qb, ks in VARS:
            for mode in (0, 2, 12):
                lib.gd_attn_fwd_set_config(qb, ks); lib.gd_attn_fwd_set_even_split(mode)
                res.setdefault((nm, mode), []).append(t(lambda: ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True)))
    for nm, _, _ in VARS:
        a, b, c = (sorted(res[(nm, m)])[2] for m in (0, 2, 12))
        print(f"BH={BH:2d} {nm:8s} unsplit {a:6.1f}   even split {b:6.1f}   even split without ticket/merge {c:6.1f}", flush=True)
lib.gd_attn_fwd_set_config(-1, 0); lib.gd_attn_fwd_set_even_split(1)
