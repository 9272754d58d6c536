"""Phase timeline of k_attn_fwd_w64 from a timing build with debug bit 32 (tools/build_dbg.sh 32; GD_LIB=tools/ub/build/libgd_dbg32.so):
per workgroup and segment the 100 MHz timestamps 0 segment start, 1 loop start, 2 loop end, 3 ticket done, 4 merge done, 5 output written
(development aid)."""
import os as _os; _os.environ.setdefault('GD_ATTN_DEV_MODES', '1')  # development hand-off modes 10-12 of gd_attn_fwd_set_even_split
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops, _lib
lib = _lib.load(os.environ.get("GD_LIB", _lib.LIB_PATH))
dt = torch.bfloat16
N = 4096
for BH, mode in ((15, 1), (20, 1), (10, 2), (5, 2)):
    g = torch.Generator(device="cuda").manual_seed(BH)
    q = (torch.randn(BH, N, 64, device="cuda", generator=g) * 0.18).to(dt); k = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt); v = torch.randn(BH, N, 64, device="cuda", generator=g).to(dt)
    o = torch.empty_like(q)
    lib.gd_attn_fwd_set_config(8, 1); lib.gd_attn_fwd_set_even_split(mode)
    for _ in range(20): ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True)
    torch.cuda.synchronize()
    ws = ops._SK_WS[0]
    OFF = 2 * 512 * 4 * 9 * 64 * 16 + (1 << 20)      # GD_SK_SLOT_BYTES + 1 MB
    ws[OFF:OFF + 1024 * 4 * 8 * 8].zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(10): ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True)
    e0.record(); ops.attn_fwd([(q, k, v, o, None)], 0.125, q_scaled=True); e1.record()
    torch.cuda.synchronize()
    st = ws[OFF:OFF + 1024 * 4 * 8 * 8].view(torch.int64).view(1024, 4, 8).cpu().double()
    used = st[:, 0, 0] > 0
    st = st[used]
    t0 = st[:, 0, 0].min()
    st = (st - t0) * 0.01                      # us since the first workgroup's start
    st[st < 0] = float("nan")
    nwg = st.shape[0]
    print(f"== {BH} heads, split mode {mode}: {nwg} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (event bracket, single launch)")
    end = torch.nan_to_num(st[:, :, 5], nan=0.0).max(1).values
    print(f"   workgroup start: median {st[:, 0, 0].median():.1f} max {st[:, 0, 0].max():.1f};  workgroup end: median {end.median():.1f} max {end.max():.1f}")
    for sgi in range(3):
        s = st[:, sgi]
        ok = ~torch.isnan(s[:, 5]) & (s[:, 5] > 0)
        if ok.sum() == 0: continue
        s = s[ok]
        d = lambda a, b: (s[:, b] - s[:, a])
        pro, loop = d(0, 1), d(1, 2)
        part = ~torch.isnan(s[:, 3]) & (s[:, 3] > s[:, 2])
        line = f"   segment {sgi}: n={int(ok.sum())}  prologue {pro.median():.1f} (max {pro.max():.1f})  loop {loop.median():.1f} (max {loop.max():.1f})"
        if part.any():
            tk = (s[part, 3] - s[part, 2]); mg = (s[part, 4] - s[part, 3])
            line += f"  partial store+ticket {tk.median():.1f} (max {tk.max():.1f}) n={int(part.sum())}  merge {mg.median():.1f} (max {mg.max():.1f})"
        ep = (s[:, 5] - torch.where(part, s[:, 4], s[:, 2]))
        line += f"  output {ep.median():.1f} (max {ep.max():.1f})"
        print(line)
    if os.environ.get("RAW"):
        print(st[:2, 0]); break
lib.gd_attn_fwd_set_config(-1, 0); lib.gd_attn_fwd_set_even_split(1)
