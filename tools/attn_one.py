"""Launch the 64^2 forward attention a few times (for rocprofv3 --pmc runs).  BH = heads (plain head-major launch); FORM=cfg: the CFG
pass's launch of an edit instead (4 token-major segments x 5 heads that share K / V, fused query warp + row list on one of them)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
BH, N, M = int(os.environ.get("BH", "32")), 4096, 4096
torch.manual_seed(0)
if "HS" in os.environ:       # hand-off of split units: 0 = release / acquire fences, 1 = cache-policy bits only (default)
    ops.ATTN_CFG.update(handoff=int(os.environ["HS"]))
if "ES" in os.environ:       # even split of the key tiles: 0 never, 1 where the launcher's cost model says so (default), 2 every launch that can be split
    ops.ATTN_CFG.update(even_split=int(os.environ["ES"]))
QS = os.environ.get("QS", "0") == "1"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
DT = torch.float16 if os.environ.get("DT", "bf16") == "fp16" else torch.bfloat16      # DT=fp16: the same launches in the reference's autocast dtype
if os.environ.get("FORM") == "cfg":
    heads, K, C = 5, 15, 320
    q = (torch.randn(3, N, C, device="cuda") * 0.2).to(DT); k = torch.randn(3, N, C, device="cuda").to(DT); v = torch.randn(3, N, C, device="cuda").to(DT)
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
    m = (((yy - 33) ** 2 + (xx - 33) ** 2) < 13 ** 2).float().reshape(-1).cuda()
    idx = torch.full((N, K), -1, dtype=torch.int32, device="cuda"); w = torch.zeros(N, K, device="cuda")
    for j in range(4):
        idx[:, j] = torch.where(m > 0, (torch.arange(N, device="cuda") + 64 * 3 + 5 + j) % N, torch.full((N,), -1, device="cuda")).int(); w[:, j] = 0.25
    rows = torch.nonzero(m > 0).reshape(-1).int(); R = rows.numel(); Rp = -(-R // 256) * 256
    rows_p = torch.cat([rows, torch.zeros(Rp - R, dtype=torch.int32, device="cuda")]).contiguous(); n_dev = torch.tensor([R], dtype=torch.int32, device="cuda")
    o = [torch.empty_like(q[:1]) for _ in range(3)]; act = torch.empty(1, Rp, C, device="cuda", dtype=DT)
    segs = [(q[0:1], k[0:1], v[0:1], o[0], None), (q[1:2], k[1:2], v[1:2], o[1], None),
            (q[1:2], k[1:2], v[1:2], act, None, (idx, w, m), (rows_p, n_dev)), (q[2:3], k[1:2], v[1:2], o[2], None)]
    run = lambda: ops.attn_fwd(segs, 0.125, heads=heads, q_scaled=True)        # the product path: even-split workspace, parts for the row-list units
elif os.environ.get("FORM") in ("cfg4n", "cfg4n_split"):
    # the CFG pass's launch when it carries the next step's reference row (editor.REF_AHEAD): 3 vanilla rows + the row-list segment + the
    # replace segment = 25 heads; cfg4n_split: the carried row as its own 5-head launch next to the 20-head launch of FORM=cfg
    heads, K, C = 5, 15, 320
    q = (torch.randn(4, N, C, device="cuda") * 0.2).to(DT); k = torch.randn(4, N, C, device="cuda").to(DT); v = torch.randn(4, N, C, device="cuda").to(DT)
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
    m = (((yy - 33) ** 2 + (xx - 33) ** 2) < 13 ** 2).float().reshape(-1).cuda()
    idx = torch.full((N, K), -1, dtype=torch.int32, device="cuda"); w = torch.zeros(N, K, device="cuda")
    for j in range(4):
        idx[:, j] = torch.where(m > 0, (torch.arange(N, device="cuda") + 64 * 3 + 5 + j) % N, torch.full((N,), -1, device="cuda")).int(); w[:, j] = 0.25
    rows = torch.nonzero(m > 0).reshape(-1).int(); R = rows.numel(); Rp = -(-R // 256) * 256
    rows_p = torch.cat([rows, torch.zeros(Rp - R, dtype=torch.int32, device="cuda")]).contiguous(); n_dev = torch.tensor([R], dtype=torch.int32, device="cuda")
    o = torch.empty_like(q); act = torch.empty(1, Rp, C, device="cuda", dtype=DT)
    rl = (q[2:3], k[2:3], v[2:3], act, None, (idx, w, m), (rows_p, n_dev)); rep = (q[3:4], k[2:3], v[2:3], o[3:4], None)
    if os.environ.get("FORM") == "cfg4n":
        segs = [(q[0:3], k[0:3], v[0:3], o[0:3], None), rl, rep]
        run = lambda: ops.attn_fwd(segs, 0.125, heads=heads, q_scaled=True)
    else:
        s1 = [(q[0:1], k[0:1], v[0:1], o[0:1], None)]; s2 = [(q[1:3], k[1:3], v[1:3], o[1:3], None), rl, rep]
        def run():
            ops.attn_fwd(s1, 0.125, heads=heads, q_scaled=True); ops.attn_fwd(s2, 0.125, heads=heads, q_scaled=True)
elif os.environ.get("FORM") == "cfgb4":
    # the CFG pass's 64^2 launch of a BATCH of 4 edits (geodiffuser_amd/batch.py): the vanilla rows of all edits as one segment (8 rows x 5
    # heads), one warped / row-list segment per edit (own tables), the replace attention of all edits as one segment: 80 heads, 6 segments
    heads, K, C, B = 5, 15, 320, 4
    q = (torch.randn(3 * B, N, C, device="cuda") * 0.2).to(DT); k = torch.randn(3 * B, N, C, device="cuda").to(DT); v = torch.randn(3 * B, N, C, device="cuda").to(DT)
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
    out = torch.empty(2 * B, N, C, device="cuda", dtype=DT); rep = torch.empty(B, N, C, device="cuda", dtype=DT)
    segs = [(q[:2 * B], k[:2 * B], v[:2 * B], out, None)]
    keep = []
    for j in range(B):
        m = (((yy - 25 - 5 * j) ** 2 + (xx - 30 + 3 * j) ** 2) < (11 + j) ** 2).float().reshape(-1).cuda()
        idx = torch.full((N, K), -1, dtype=torch.int32, device="cuda"); w = torch.zeros(N, K, device="cuda")
        for jj in range(4):
            idx[:, jj] = torch.where(m > 0, (torch.arange(N, device="cuda") + 64 * 3 + 5 + jj) % N, torch.full((N,), -1, device="cuda")).int(); w[:, jj] = 0.25
        rows = torch.nonzero(m > 0).reshape(-1).int(); R = rows.numel(); Rp = -(-R // 256) * 256
        rows_p = torch.cat([rows, torch.zeros(Rp - R, dtype=torch.int32, device="cuda")]).contiguous(); n_dev = torch.tensor([R], dtype=torch.int32, device="cuda")
        act = torch.empty(1, Rp, C, device="cuda", dtype=DT)
        keep.append((m, idx, w, rows_p, n_dev, act))
        segs.append((q[B + j:B + j + 1], k[B + j:B + j + 1], v[B + j:B + j + 1], act, None, (idx, w, m), (rows_p, n_dev)))
    segs.append((q[2 * B:], k[B:2 * B], v[B:2 * B], rep, None))
    run = lambda: ops.attn_fwd(segs, 0.125, heads=heads, q_scaled=True)
elif os.environ.get("FORM") == "opt":
    # the optimisation pass's launch (head-major, 5 heads: reference rows + LSE, row-list edit rows with the fused warp, replace rows + LSE;
    # pre-scaled queries with row sums over the rounded probabilities: q_scaled = 2, k_attn_fwd_w64 LSUM)
    f, K = 5, 15
    q = (torch.randn(2 * f, N, 64, device="cuda") * 0.2).to(DT); k = torch.randn(f, N, 64, device="cuda").to(DT); v = torch.randn(f, N, 64, device="cuda").to(DT)
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
    m = (((yy - 33) ** 2 + (xx - 33) ** 2) < 13 ** 2).float().reshape(-1).cuda()
    idx = torch.full((N, K), -1, dtype=torch.int32, device="cuda"); w = torch.zeros(N, K, device="cuda")
    for j in range(4):
        idx[:, j] = torch.where(m > 0, (torch.arange(N, device="cuda") + 64 * 3 + 5 + j) % N, torch.full((N,), -1, device="cuda")).int(); w[:, j] = 0.25
    rows = torch.nonzero(m > 0).reshape(-1).int(); R = rows.numel(); Rp = -(-R // 256) * 256
    rows_p = torch.cat([rows, torch.zeros(Rp - R, dtype=torch.int32, device="cuda")]).contiguous(); n_dev = torch.tensor([R], dtype=torch.int32, device="cuda")
    o0 = torch.empty_like(q[:f]); o2 = torch.empty_like(q[:f]); act = torch.empty(f, Rp, 64, device="cuda", dtype=DT)
    l0 = torch.empty(f, N, device="cuda"); l2 = torch.empty(f, N, device="cuda")
    segs = [(q[:f], k, v, o0, l0), (q[:f], k, v, act, None, (idx, w, m), (rows_p, n_dev)), (q[f:], k, v, o2, l2)]
    run = lambda: ops.attn_fwd(segs, 0.6931471805599453, q_scaled=2 if os.environ.get("LSUM", "1") == "1" else 0)
else:
    q = (torch.randn(BH, N, 64, device="cuda") * 1.2).to(DT); k = (torch.randn(BH, M, 64, device="cuda") * 1.2).to(DT); v = torch.randn(BH, M, 64, device="cuda").to(DT)
    out = torch.empty_like(q); lse = torch.empty(BH, N, device="cuda")
    if QS:
        q = (q.float() * (0.125 * 1.4426950408889634)).to(DT)
    run = lambda: ops.attn_fwd([(q, k, v, out, lse if not QS else None)], 0.125, q_scaled=QS)
if os.environ.get("TIME") == "1":        # back-to-back launches inside one event bracket (what bench.py's replay does): us per launch
    for _ in range(20):
        run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"FORM={os.environ.get('FORM', 'plain')} BH={BH} QS={int(QS)} DT={os.environ.get('DT', 'bf16')}: {1e3 * e0.elapsed_time(e1) / reps:.1f} us per launch ({reps} launches)")
else:
    for _ in range(reps):
        run()
torch.cuda.synchronize()
