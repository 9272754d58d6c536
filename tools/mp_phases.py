"""Phase timeline of k_attn_fwd_mp (timing build with debug bit 32): 0 segment start, 1 loop start, 2 loop end, 3 key ranges merged,
4 hand-off done, 5 output written; per workgroup, wave 0 (development aid)."""
import os as _os; _os.environ.setdefault('GD_ATTN_DEV_MODES', '1')  # development hand-off modes 10-12 of gd_attn_fwd_set_even_split
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))
dt = torch.bfloat16
N = 4096
for BH, qb, ks, mode in ((5, 4, 2, 1), (5, 4, 1, 2), (15, 4, 1, 1)):
    g = torch.Generator(device="cuda").manual_seed(BH)
    q = (torch.randn(BH, N, 64, device="cuda", generator=g) * 0.18).to(dt); k = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt); v = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt)
    o = torch.empty_like(q)
    lib.gd_attn_fwd_set_config(qb, ks); lib.gd_attn_fwd_set_even_split(mode)
    for _ in range(20): ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True)
    torch.cuda.synchronize()
    ws = ops._SK_WS[0]
    OFF = 2 * 512 * 4 * 9 * 64 * 16 + (1 << 20)
    ws[OFF:OFF + 1024 * 4 * 8 * 8].zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(10): ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True)
    e0.record(); ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True); e1.record()
    torch.cuda.synchronize()
    st = ws[OFF:OFF + 1024 * 4 * 8 * 8].view(torch.int64).view(1024, 4, 8).cpu().double()
    st = st[st[:, 0, 0] > 0]
    t0 = st[:, 0, 0].min()
    st = (st - t0) * 0.01
    st[st < 0] = float("nan")
    end = torch.nan_to_num(st[:, :, 5], nan=0.0).max(1).values
    print(f"== {BH} heads, QB={qb} KS={ks}, split mode {mode}: {st.shape[0]} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (event bracket, single launch)")
    print(f"   workgroup start: median {st[:, 0, 0].median():.1f} max {st[:, 0, 0].max():.1f};  workgroup end: median {end.median():.1f} max {end.max():.1f}")
    for sgi in range(3):
        s = st[:, sgi]
        ok = ~torch.isnan(s[:, 5]) & (s[:, 5] > 0)
        if ok.sum() == 0: continue
        s = s[ok]
        f = lambda a, b: f"{(s[:, b] - s[:, a]).median():.1f} (max {(s[:, b] - s[:, a]).max():.1f})"
        print(f"   segment {sgi}: n={int(ok.sum())}  prologue {f(0, 1)}  loop {f(1, 2)}  last PV + key-range merge {f(2, 3)}  hand-off {f(3, 4)}  output {f(4, 5)}")
lib.gd_attn_fwd_set_config(-1, 0); lib.gd_attn_fwd_set_even_split(1)
