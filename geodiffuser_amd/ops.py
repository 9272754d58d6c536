"""Thin torch <-> C-ABI glue: every function takes/returns CUDA(HIP) tensors and launches one entry point of
libgeodiff_hip.so on torch's current stream.  No arithmetic happens here; there is no CPU fallback — a CPU
tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes
import threading
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GD_BF16, GD_CHANNEL_MAJOR, GD_F16, GD_F32, GD_TOKEN_MAJOR, GdAttnSeg, GdEditLosses, GdHeadsMerge, GdHeadsSplit, GdProbs, GdRemovalBwd, check

_DT = {torch.float16: GD_F16, torch.bfloat16: GD_BF16, torch.float32: GD_F32}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ZeroPool:
    """The few-float f32 accumulators of the loss kernels (5 terms, one scalar ...) come out of one pre-zeroed chunk instead of one
    ``torch.zeros`` fill launch each (~160 per optimisation pass).  A slice is handed out once and never reused; a chunk never spans a
    hipGraph capture boundary — the key holds the runtime's capture id, and graphs.py also calls ``zero_pool_reset`` around its captures —
    because a slice zeroed by a fill OUTSIDE a graph would not be re-zeroed by the graph's replays."""
    CHUNK = 2048

    def __init__(self):
        self.buf, self.pos, self.key, self.gen = None, 0, None, 0

    @staticmethod
    def _capture_id() -> int:
        """0 outside a capture, otherwise 1 + the runtime's id of the capture sequence the current stream is recording."""
        if not torch.cuda.is_current_stream_capturing():
            return 0
        cid = ctypes.c_ulonglong(0)
        check(_lib.load().gd_stream_capture_id(_stream(), ctypes.byref(cid)), "gd_stream_capture_id")
        return int(cid.value)

    def take(self, n: int, device) -> torch.Tensor:
        if n > self.CHUNK // 4:                               # not a "few floats" request: its own buffer
            return torch.zeros(n, dtype=torch.float32, device=device)
        key = (torch.device(device), self._capture_id(), self.gen)
        if self.buf is None or self.key != key or self.pos + n > self.CHUNK:
            self.buf = torch.zeros(self.CHUNK, dtype=torch.float32, device=device)
            self.pos, self.key = 0, key
        v = self.buf[self.pos:self.pos + n]
        self.pos += (n + 3) // 4 * 4                      # keep every slice 16-byte aligned
        return v


# One pool / workspace / ticket PER LAUNCHING THREAD: the library's contract is one stream at a time per caller, and
# editor.start_ahead adds a second launching thread on a side stream beside the caller's.  A chunk zero-filled on one stream must not be
# sliced and consumed on the other before the fill has run there, and the even split's arrival counters / part slots and the loss tail's
# ticket must not be shared by two launches in flight on different streams.  Captures and replays all happen on the caller's thread.
_TLS = threading.local()
_POOL_GEN = [0]


def _zeros_pool() -> _ZeroPool:
    pool = getattr(_TLS, "zeros", None)
    if pool is None:
        pool = _TLS.zeros = _ZeroPool()
    pool.gen = _POOL_GEN[0]
    return pool


def zeros_f32(n: int, device) -> torch.Tensor:
    """n zeroed floats (a slice of the calling thread's current pre-zeroed chunk)."""
    return _zeros_pool().take(n, device)


def zero_pool_reset() -> None:
    """Called around every hipGraph capture (graphs.py): the next request (of any thread) starts a fresh chunk."""
    _POOL_GEN[0] += 1


def _need(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise _lib.GeodiffError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback), got {t.device}")
    if not t.is_contiguous():
        raise _lib.GeodiffError(f"{name}: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _lib.GeodiffError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _dt16(t: torch.Tensor, name: str) -> int:
    if t.dtype not in (torch.float16, torch.bfloat16):
        raise _lib.GeodiffError(f"{name}: expected float16/bfloat16, got {t.dtype}")
    return _DT[t.dtype]


# ---------------------------------------------------------------------------------------------------
# R3 splat
# ---------------------------------------------------------------------------------------------------
def rasterize_points(pts: torch.Tensor, S: int, radius_ndc: float, K: int, want_zbuf: bool = False):
    """pts [P,3] f32 (x,y negated already) -> idx [S,S,K] i32, dist2 [S,S,K] f32 (, zbuf)."""
    lib = _lib.load()
    _need(pts, "pts", torch.float32)
    P = pts.shape[0]
    idx = torch.empty(S, S, K, dtype=torch.int32, device=pts.device)
    dist2 = torch.empty(S, S, K, dtype=torch.float32, device=pts.device)
    zbuf = torch.empty(S, S, K, dtype=torch.float32, device=pts.device) if want_zbuf else None
    nbytes = lib.gd_rasterize_workspace_bytes(P, S, radius_ndc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    check(lib.gd_rasterize_points(_p(pts), P, S, radius_ndc, K, _p(idx), _p(zbuf), _p(dist2), _p(ws), nbytes, _stream()),
          "gd_rasterize_points")
    return (idx, dist2, zbuf) if want_zbuf else (idx, dist2)


def splat_weights(idx: torch.Tensor, dist2: torch.Tensor, radius_ndc: float, rad_pow: float = 2.0, tau: float = 1.0):
    lib = _lib.load()
    _need(idx, "idx", torch.int32); _need(dist2, "dist2", torch.float32)
    K = idx.shape[-1]
    npix = idx.numel() // K
    w = torch.empty(npix, K, dtype=torch.float32, device=idx.device)
    check(lib.gd_splat_weights(_p(idx), _p(dist2), npix, K, radius_ndc, rad_pow, tau, _p(w), _stream()), "gd_splat_weights")
    return w


def splat_composite(src: torch.Tensor, idx: torch.Tensor, w: torch.Tensor, m: Optional[torch.Tensor], layout: int,
                    out: Optional[torch.Tensor] = None):
    """src [B,P,C] (token-major) or [B,C,P] (channel-major); returns the same shape with P -> npix."""
    lib = _lib.load()
    _need(src, "src"); _need(idx, "idx", torch.int32); _need(w, "w", torch.float32)
    if m is not None:
        _need(m, "m", torch.float32)
    K = idx.shape[-1]
    npix = idx.numel() // K
    if layout == GD_TOKEN_MAJOR:
        B, P, C = src.shape
        shape = (B, npix, C)
    else:
        B, C, P = src.shape
        shape = (B, C, npix)
    if out is None:
        out = torch.empty(shape, dtype=src.dtype, device=src.device)
    check(lib.gd_splat_composite(_p(src), _p(idx), _p(w), _p(m), B, P, C, npix, K, layout, _p(out), _DT[src.dtype], _stream()),
          "gd_splat_composite")
    return out


def mesh_coverage(verts, faces, S: int):
    """verts [V,3] f32, faces [F,3] i32 -> coverage [S,S] f32 in {0,1}."""
    lib = _lib.load()
    _need(verts, "verts", torch.float32)
    if faces.numel():
        _need(faces, "faces", torch.int32)
    out = torch.empty(S, S, dtype=torch.float32, device=verts.device)
    if verts.numel() == 0 or faces.numel() == 0:               # empty mesh: nothing is covered (no launch needed, no null pointers)
        return out.zero_()
    check(lib.gd_mesh_coverage(_p(verts), _p(faces), verts.shape[0], faces.shape[0], S, _p(out), _stream()), "gd_mesh_coverage")
    return out


# ---------------------------------------------------------------------------------------------------
# R5-R7 attention
# ---------------------------------------------------------------------------------------------------
SPLIT_KV = os.environ.get("GD_ATTN_SPLIT_KV", "1") == "1"
_PLAN_CACHE = {}


def _env_attn_cfg():
    qb, ks = -1, 0
    e = os.environ.get("GD_ATTN_CFG")               # "QBxKS" forces one pipelined kernel, "0" the plain kernel (development / benchmarks)
    if e == "0":
        qb = 0
    elif e and "x" in e:
        qb, ks = (int(x) for x in e.split("x"))
    return dict(even_split=int(os.environ.get("GD_ATTN_EVEN_SPLIT", "-1")), qb=qb, ks=ks, handoff=1)


# Launch-configuration DEFAULTS of this process's Python layer.  The C ABI is stateless (ABI 5: every call carries its gd_attn_cfg_t /
# gd_conv3x3_cfg_t / single_launch flag); these dictionaries are what ops.* fills those per-call arguments from when the caller passes
# none.  Tests and benchmarks edit them (and restore them); the product path never does.
ATTN_CFG = _env_attn_cfg()
CONV3X3_CFG = dict(pi=0, ki=0, ksplit=0, dma=0 if os.environ.get("GD_CONV_DMA") == "0" else -1)
GN_SINGLE_LAUNCH = -1


def _attn_cfg(cfg: Optional[dict], nsplit: int = 0):
    c = dict(ATTN_CFG)
    if cfg:
        c.update(cfg)
    return _lib.GdAttnCfg(int(c["even_split"]), int(c["qb"]), int(c["ks"]), int(nsplit), int(c["handoff"]))


_SK_WS = {}          # (thread, device index) -> zero-initialised workspace of the even split (persistent: captured graphs hold its address)
_SK_WS_RETIRED = []  # buffers replaced by a larger one (never freed: see _attn_ws)


def _attn_ws(lib, dev: torch.device, tot_bh: int, N: int, M: int):
    """Workspace of the even split (gd_attn_fwd): arrival counters (zero before the first launch, every launch leaves them zero) + part slots.  One
    buffer per launching thread and device, grown on demand — all launches of a thread run on one stream at a time (the library's contract)."""
    need = int(lib.gd_attn_fwd_workspace_bytes(tot_bh, N, M))
    wkey = (threading.get_ident(), dev.index)
    ws = _SK_WS.get(wkey)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            # (a buffer allocated during a capture lives in that graph's private pool and would be zero-filled again by every replay)
            raise _lib.GeodiffError("attn_fwd: the even-split workspace must exist before a graph capture (run one eager pass first)")
        if ws is not None:
            _SK_WS_RETIRED.append(ws)                      # captured passes may still address the smaller buffer: it stays alive
        ws = _SK_WS[wkey] = torch.zeros(max(need, 40 << 20), dtype=torch.uint8, device=dev)
    return ws


def _attn_plan(lib, tot_bh: int, N: int, M: int):
    key = (tot_bh, N, M)
    p = _PLAN_CACHE.get(key)
    if p is None:
        nb = ctypes.c_size_t(0)
        ns = int(lib.gd_attn_fwd_plan(tot_bh, N, M, ctypes.byref(nb)))
        p = _PLAN_CACHE[key] = (ns, int(nb.value))
    return p


def _attn_seg_array(segs: Sequence[tuple], heads: int, q_scaled, what: str = "attn_fwd"):
    """-> (GdAttnSeg array, n, N, M, D, dtype code, tot_bh) for a list of (q, k, v, out | None, lse | None[, warp[, q_rows]]) tuples."""
    n = len(segs)
    arr = (GdAttnSeg * n)()
    q0, k0 = segs[0][0], segs[0][1]
    D = 64 if heads else q0.shape[2]
    N, M = q0.shape[1], k0.shape[1]
    dt = _dt16(q0, "q")
    tot_bh = 0
    for i, seg in enumerate(segs):
        q, k, v, o, lse = seg[:5]
        warp = seg[5] if len(seg) > 5 else None
        qrows = seg[6] if len(seg) > 6 else None          # (rows i32 [R], n_valid i32 [1] on the device): see gd_attn_seg_t.q_rows
        for t, nm in ((q, "q"), (k, "k"), (v, "v")) + (((o, "out"),) if o is not None else ()):
            _need(t, nm, q0.dtype)
        No = N if qrows is None else int(qrows[0].numel())
        if heads:
            ok = q.shape[1:] == (N, heads * D) and k.shape[1:] == (M, heads * D) and v.shape == k.shape and k.shape[0] == q.shape[0] \
                and (o is None or tuple(o.shape) == (q.shape[0], No, heads * D))
        else:
            ok = q.shape[1:] == (N, D) and k.shape[1:] == (M, D) and v.shape == k.shape and k.shape[0] == q.shape[0] \
                and (o is None or tuple(o.shape) == (q.shape[0], No, D))
        if not ok:
            raise _lib.GeodiffError(f"{what}: segment shapes disagree")
        if lse is not None:
            _need(lse, "lse", torch.float32)
        rl = rn = 0
        if qrows is not None:
            _need(qrows[0], "q_rows", torch.int32); _need(qrows[1], "q_rows_n", torch.int32)
            rl, rn = qrows[0].data_ptr(), qrows[1].data_ptr()
        bh = q.shape[0] * (heads if heads else 1)
        tot_bh += bh
        widx = ww = wm = 0
        wK = 0
        if warp is not None:
            t_idx, t_w, t_m = warp
            _need(t_idx, "warp idx", torch.int32); _need(t_w, "warp w", torch.float32)
            wK = t_idx.shape[-1]
            if t_idx.numel() != N * wK or t_w.numel() != N * wK or (t_m is not None and t_m.numel() != N):
                raise _lib.GeodiffError(f"{what}: warp tables must have N rows")
            if t_m is not None:
                _need(t_m, "warp m", torch.float32)
            widx, ww, wm = t_idx.data_ptr(), t_w.data_ptr(), 0 if t_m is None else t_m.data_ptr()
        arr[i] = GdAttnSeg(q.data_ptr(), k.data_ptr(), v.data_ptr(), 0 if o is None else o.data_ptr(), 0 if lse is None else lse.data_ptr(), bh, heads,
                           widx, ww, wm, wK, int(q_scaled) if q_scaled in (0, 1, 2) else int(bool(q_scaled)), rl, rn, 0 if qrows is None else No)
    return arr, n, N, M, D, dt, tot_bh


def attn_fwd_pair(segs: Sequence[tuple], side_b, m, scale: float, heads: int = 0, q_scaled: bool = False) -> None:
    """Short-key launches (at most 128 keys) with the blend inside: ``segs`` as attn_fwd (no LSE, no row lists).  One pair: the LAST segment
    is side A, ``side_b`` = (q, k, v[, warp]) its side B and ``m`` [N] f32 the mask: segs[-1]'s out = A*m + B*(1-m) as blend_tokens computes
    it from the two attention outputs.  P pairs (one per edit of a batch): ``side_b`` a LIST of P such tuples, ``m`` a list of P masks and
    the LAST P segments their A sides (gd_attn_fwd_pair)."""
    lib = _lib.load()
    many = isinstance(side_b, list)
    bs, ms = (side_b, m) if many else ([side_b], [m])
    P = len(bs)
    if len(ms) != P or P > len(segs):
        raise _lib.GeodiffError("attn_fwd_pair: one mask and one A side per B side")
    arr, n, N, M, D, dt, _ = _attn_seg_array(segs, heads, q_scaled, "attn_fwd_pair")
    arr_b, _, Nb, Mb, _, _, _ = _attn_seg_array([(b[0], b[1], b[2], None, None) + tuple(b[3:4]) for b in bs], heads, q_scaled, "attn_fwd_pair")
    mp = (ctypes.c_void_p * P)()
    for i, (b, mm) in enumerate(zip(bs, ms)):
        _need(mm, "m", torch.float32)
        if mm.numel() != N or b[0].shape != segs[len(segs) - P + i][0].shape:
            raise _lib.GeodiffError("attn_fwd_pair: side B / mask do not match side A")
        mp[i] = mm.data_ptr()
    if Nb != N or Mb != M:
        raise _lib.GeodiffError("attn_fwd_pair: side B / mask do not match side A")
    check(lib.gd_attn_fwd_pair(arr, n, arr_b, mp, P, N, M, D, scale, dt, _stream()), "gd_attn_fwd_pair")


def attn_fwd(segs: Sequence[tuple], scale: float, heads: int = 0, nsplit: Optional[int] = None, q_scaled: bool = False,
             cfg: Optional[dict] = None) -> None:
    """segs: list of (q, k, v, out, lse | None[, warp]); one launch.
    heads == 0: q/out [bh,N,D], k/v [bh,M,D] (head-major).  heads > 0: token-major q/out [B,N,heads*D], k/v [B,M,heads*D]
    exactly as to_q/to_k/to_v produce them (no head_to_batch_dim copies); lse [B*heads, N].
    warp = (idx [N,K] i32, w [N,K] f32, m [N] f32 | None): the segment attends with the warped, blended queries
    q*(1-m) + m*half(sum_k w*q[idx]) built inside the kernel (U/attention_processors.py:424-428,544-549) — the fused
    attention-warp launch; bit-identical to passing splat_composite(q, idx, w, m) as q.
    q_scaled: every q already carries scale*log2(e) (applied by the projection GEMM before its rounding, see attention_processors
    ``_project_qkv``); ``scale`` is then ignored.  2: additionally row sums over the rounded probabilities (gd_attn_seg_t.q_scaled).
    cfg: overrides of ATTN_CFG for this call (gd_attn_cfg_t fields: even_split, qb, ks, handoff)."""
    lib = _lib.load()
    arr, n, N, M, D, dt, tot_bh = _attn_seg_array(segs, heads, q_scaled)
    q0 = segs[0][0]
    if nsplit is None:
        nsplit, ws_bytes = _attn_plan(lib, tot_bh, N, M) if (SPLIT_KV and ATTN_CFG["qb"] < 0 and not (cfg and cfg.get("qb", -1) >= 0)) else (1, 0)
    else:
        ws_bytes = nsplit * tot_bh * N * (D + 2) * 4 if nsplit > 1 else 0
    if nsplit > 1:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q0.device)
        c = _attn_cfg(cfg, nsplit)
        check(lib.gd_attn_fwd(arr, n, N, M, D, scale, ctypes.byref(c), _p(ws), ws_bytes, dt, _stream()), "gd_attn_fwd (split-KV)")
        return
    c = _attn_cfg(cfg)
    if D == 64 and M % 256 == 0 and M >= 1024 and c.even_split != 0:
        ws = _attn_ws(lib, q0.device, tot_bh, N, M)
        check(lib.gd_attn_fwd(arr, n, N, M, D, scale, ctypes.byref(c), _p(ws), ws.numel(), dt, _stream()), "gd_attn_fwd (even split)")
    else:
        check(lib.gd_attn_fwd(arr, n, N, M, D, scale, ctypes.byref(c), None, 0, dt, _stream()), "gd_attn_fwd")


class RowCopyTable:
    """``gd_copy_rows`` table: for every (src [rows, ...], dst [1, ...]) pair, dst <- src[row] in ONE launch (``copy(row)``).  The (src, dst, bytes)
    triples live in device memory; built once per set of tensors (one upload)."""

    def __init__(self, pairs):
        ent, self.keep, self.max_bytes = [], [], 0
        for src, dst in pairs:
            _need(src, "copy_rows src"); _need(dst, "copy_rows dst", src.dtype)
            nb = dst.numel() * dst.element_size()
            if src.dim() < 1 or src.numel() % max(1, src.shape[0]) or src.numel() // src.shape[0] != dst.numel() or nb % 16 or src.data_ptr() % 16 or dst.data_ptr() % 16:
                raise _lib.GeodiffError("copy_rows: dst must hold exactly one row of src, 16-byte multiples, 16-byte aligned")
            ent.append((src.data_ptr(), dst.data_ptr(), nb))
            self.keep.append((src, dst))
            self.max_bytes = max(self.max_bytes, nb)
        self.rows = min(src.shape[0] for src, _ in pairs)
        self.n = len(ent)
        self.table = torch.tensor(ent, dtype=torch.int64).to(pairs[0][0].device)

    def copy(self, row: int) -> None:
        if not 0 <= row < self.rows:
            raise _lib.GeodiffError(f"copy_rows: row {row} outside 0..{self.rows - 1}")
        check(_lib.load().gd_copy_rows(_p(self.table), self.n, int(row), self.max_bytes, _stream()), "gd_copy_rows")


def rows_merge(base: torch.Tensor, act: torch.Tensor, pos: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[h, n] = act[h, pos[n]] where pos[n] >= 0, else base[h, n]  (base / out [H, N, D], act [H, R, D] 16-bit, pos [N] i32):
    gd_blend_merge without the blend."""
    if out is None:
        out = torch.empty_like(base)
    blend_merge(base, act, pos, None, None, eo_out=out, out=None)
    return out


def fp8_quantize(q, k, v, scale: float, heads: int = 0):
    """Per-head e4m3 quantisation for ``attn_fwd_fp8`` (``scale``: the softmax scale, whose mantissa is folded into q8): q, k, v 16-bit, head-major [BH, N | M, 64] (heads = 0) or token-major
    [B, N | M, heads*64].  -> dict(q8 [BH,N,64] u8, k8 [BH,M,64] u8, vt8 [BH, M/64, 64, 64] u8, aq, ak, av [BH] f32)."""
    lib = _lib.load()
    dt = _dt16(q, "q")
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, nm, q.dtype)
    B = q.shape[0]
    BH = B * heads if heads else B
    N, M = q.shape[1], k.shape[1]
    if (q.shape[2] != (heads or 1) * 64) or k.shape[2] != q.shape[2] or v.shape != k.shape:
        raise _lib.GeodiffError("fp8_quantize: head dim must be 64")
    dev = q.device
    amax3 = torch.empty(3, BH, dtype=torch.float32, device=dev)
    aq, ak, av = amax3[0], amax3[1], amax3[2]
    q8 = torch.empty(BH, N, 64, dtype=torch.uint8, device=dev)
    k8 = torch.empty(BH, M, 64, dtype=torch.uint8, device=dev)
    vt8 = torch.empty(BH, (M + 63) // 64, 64, 64, dtype=torch.uint8, device=dev)
    check(lib.gd_fp8_quantize_qkv(_p(q), _p(k), _p(v), BH, heads, N, M, scale, _p(amax3), _p(q8), _p(k8), _p(vt8), dt, _stream()),
          "gd_fp8_quantize_qkv")
    return dict(q8=q8, k8=k8, vt8=vt8, aq=aq, ak=ak, av=av, scale=float(scale))


def attn_fwd_fp8(qz: dict, scale: float, out: torch.Tensor, lse: Optional[torch.Tensor] = None, heads: int = 0) -> None:
    """out = softmax(scale q k^T) v on the fp8 matrix instruction from ``fp8_quantize`` output; out 16-bit [BH,N,64] (heads = 0) or
    [B,N,heads*64].  Opt-in (parity is the oracle's, not the 1e-3 of the 16-bit path)."""
    lib = _lib.load()
    dt = _dt16(out, "out")
    _need(out, "out")
    BH, N, _ = qz["q8"].shape
    M = qz["k8"].shape[1]
    if float(scale) != qz["scale"]:
        raise _lib.GeodiffError("attn_fwd_fp8: scale differs from the one folded into q8 by fp8_quantize")
    if lse is not None:
        _need(lse, "lse", torch.float32)
    check(lib.gd_attn_fwd_fp8(_p(qz["q8"]), _p(qz["k8"]), _p(qz["vt8"]), _p(qz["aq"]), _p(qz["ak"]), _p(qz["av"]), BH, heads, N, M, 64, scale,
                              _p(out), _p(lse), dt, _stream()), "gd_attn_fwd_fp8")


def attn_bwd(q, k, v, out, lse, dout, scale: float, need_dk: bool, dq_out=None, variant: int = 0):
    """-> (dq 16-bit [BH,N,D], dk f32 [BH,M,D] | None).  dq_out: write dq into this contiguous tensor (e.g. a row slice of a larger gradient)."""
    lib = _lib.load()
    dt = _dt16(q, "q")
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout")):
        _need(t, nm, q.dtype)
    _need(lse, "lse", torch.float32)
    BH, N, D = q.shape
    M = k.shape[1]
    if dq_out is not None:
        _need(dq_out, "dq_out", q.dtype)
        if dq_out.shape != q.shape:
            raise _lib.GeodiffError("attn_bwd: dq_out must have q's shape")
    dq = dq_out if dq_out is not None else torch.empty_like(q)
    dk = torch.zeros(BH, M, D, dtype=torch.float32, device=q.device) if need_dk else None
    nbytes = lib.gd_attn_bwd_workspace_bytes(BH, N, M, D, int(need_dk), variant)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device) if nbytes else None
    check(lib.gd_attn_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), BH, N, M, D, scale, _p(dq), _p(dk), _p(ws), nbytes, None, None, variant, dt,
                          _stream()), "gd_attn_bwd")
    return dq, dk


def attn_bwd_dkv(q, k, v, out, lse, dout, scale: float):
    """-> (dk, dv) f32 [BH, M, D]: the key / value gradients of out = softmax(scale q k^T) v for any key count."""
    lib = _lib.load()
    dt = _dt16(q, "q")
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout")):
        _need(t, nm, q.dtype)
    _need(lse, "lse", torch.float32)
    BH, N, D = q.shape
    M = k.shape[1]
    dk = torch.zeros(BH, M, D, dtype=torch.float32, device=q.device)
    dv = torch.zeros_like(dk)
    nbytes = lib.gd_attn_bwd_dkv_workspace_bytes(BH, N, M, D)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    check(lib.gd_attn_bwd_dkv(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), BH, N, M, D, scale, _p(dk), _p(dv), _p(ws), nbytes, dt,
                              _stream()), "gd_attn_bwd_dkv")
    return dk, dv


def _probs_problem(q, k, lse, rows, n_valid):
    """-> (GdProbs, P [BH, R, Mpad] 16-bit, Mpad = ceil8(M))."""
    _need(q, "q"); _need(k, "k", q.dtype); _need(lse, "lse", torch.float32)
    BH, N, D = q.shape
    M = k.shape[1]
    Mpad = (M + 7) // 8 * 8
    R = N if rows is None else rows.numel()
    if rows is not None:
        _need(rows, "rows", torch.int32)
    if n_valid is not None:
        _need(n_valid, "n_valid", torch.int32)
    P = torch.empty(BH, R, Mpad, dtype=q.dtype, device=q.device)
    return _lib.GdProbs(q.data_ptr(), k.data_ptr(), lse.data_ptr(), _ip(rows), _ip(n_valid), P.data_ptr(), BH, N, R, M, Mpad), P


def attn_probs(q, k, lse, rows: Optional[torch.Tensor], scale: float, n_valid: Optional[torch.Tensor] = None):
    """-> P [BH, R, Mpad] 16-bit, Mpad = ceil8(M).  n_valid (device int32[1]): slots [n_valid, R) of ``rows`` are padding; whole
    128-row tiles of padding are not computed (those rows of P stay uninitialised)."""
    lib = _lib.load()
    a, P = _probs_problem(q, k, lse, rows, n_valid)
    check(lib.gd_attn_probs(ctypes.byref(a), None, q.shape[2], scale, None, 0, _dt16(q, "q"), _stream()), "gd_attn_probs")
    return P


# ---------------------------------------------------------------------------------------------------
# R8 losses
# ---------------------------------------------------------------------------------------------------
def removal_fwd(Pe, Pb, m_inp, m_wo, rows, S: int, n_valid=None, variant: int = 0):
    """-> dict(p_in, j_in, p_wo, j_wo, wgt [H,R]) and loss_sum [1] f32 (un-normalised)."""
    lib = _lib.load()
    dt = _dt16(Pe, "Pe")
    _need(Pe, "Pe"); _need(Pb, "Pb", Pe.dtype); _need(m_inp, "m_inp", torch.float32); _need(m_wo, "m_wo", torch.float32)
    _need(rows, "rows", torch.int32)
    H, R, Mpad = Pe.shape
    N = Pb.shape[1]
    dev = Pe.device
    best = torch.empty(H, R, 2, dtype=torch.int64, device=dev)
    check(lib.gd_removal_corr_max(_p(Pe), _p(Pb), _p(m_inp), _p(m_wo), _p(n_valid), H, R, N, Mpad, _p(best), 1, variant, dt, _stream()), "gd_removal_corr_max")
    p_in = torch.empty(H, R, dtype=torch.float32, device=dev); p_wo = torch.empty_like(p_in); wgt = torch.empty_like(p_in)
    j_in = torch.empty(H, R, dtype=torch.int32, device=dev); j_wo = torch.empty_like(j_in)
    loss = zeros_f32(1, dev)
    check(lib.gd_removal_loss_reduce(_p(best), _p(rows), _p(n_valid), H, R, S, _p(p_in), _p(j_in), _p(p_wo), _p(j_wo), _p(wgt), _p(loss), _stream()),
          "gd_removal_loss_reduce")
    return dict(p_in=p_in, j_in=j_in, p_wo=p_wo, j_wo=j_wo, wgt=wgt), loss


def removal_bwd(Pe, Pb, q, k, rows, aux, m_inp, m_wo, coef: float, gscale, scale: float, dq_f32, dk_f32, n_valid=None, dq16=None,
                variant: int = 0):
    """The complete removal-loss backward: dq_f32 (f32, accumulated) and / or dq16 (16-bit, added in place with one rounding) receive the
    query gradient."""
    lib = _lib.load()
    if dq_f32 is None and dq16 is None:
        raise _lib.GeodiffError("removal_bwd: dq_f32 or dq16 is required (the products-only form is removal_bwd_nofold)")
    if dq_f32 is not None:
        _need(dq_f32, "dq_f32", torch.float32)
    if dq16 is not None:
        _need(dq16, "dq16", q.dtype)
    a, ws = removal_bwd_args(Pe, Pb, q, k, rows, aux, m_inp, m_wo, coef, gscale, None, scale, n_valid, dk_f32 is not None)
    a.dk_f32 = _ip(dk_f32)
    a.variant = variant
    check(lib.gd_removal_bwd(ctypes.byref(a), _p(dq_f32), _p(dq16), _dt16(Pe, "Pe"), _stream()), "gd_removal_bwd")


def nn_table(fg, S: int):
    """fg [N] f32 -> nn_idx [N,4] i32, nn_w [N,4] f32, w_dist [N] f32 (deterministic 4-nearest-foreground table)."""
    lib = _lib.load()
    _need(fg, "fg", torch.float32)
    N = S * S
    nn_idx = torch.empty(N, 4, dtype=torch.int32, device=fg.device)
    nn_w = torch.empty(N, 4, dtype=torch.float32, device=fg.device)
    w_dist = torch.empty(N, dtype=torch.float32, device=fg.device)
    check(lib.gd_nn_table(_p(fg), S, _p(nn_idx), _p(nn_w), _p(w_dist), _stream()), "gd_nn_table")
    return nn_idx, nn_w, w_dist


def amodal_target(eo, nn_idx, nn_w, fg, S: int):
    lib = _lib.load()
    dt = _dt16(eo, "eo")
    _need(eo, "eo"); _need(nn_idx, "nn_idx", torch.int32); _need(nn_w, "nn_w", torch.float32); _need(fg, "fg", torch.float32)
    H, N, D = eo.shape
    tgt = torch.empty(H, N, D, dtype=torch.float32, device=eo.device)
    # (scratch of the two-launch form only: head dims above 64)
    tmp = torch.empty_like(tgt) if D > 64 else tgt
    check(lib.gd_amodal_target(_p(eo), _p(nn_idx), _p(nn_w), _p(fg), H, S, D, _p(tmp), _p(tgt), dt, _stream()), "gd_amodal_target")
    return tgt


def loss_assemble(sums, rm, inv5, inv_rm, wv, inv5_bwd, use_amodal: bool):
    """-> (terms [5], loss (), coefs [5], rm_coef [1]): the per-layer scalar arithmetic of the edit losses in one launch
    (gd_loss_assemble); all operands device f32."""
    lib = _lib.load()
    for t, nm, n in ((sums, "sums", 5), (rm, "rm", 1), (inv5, "inv5", 5), (inv_rm, "inv_rm", 1), (wv, "wv", 5), (inv5_bwd, "inv5_bwd", 5)):
        _need(t, nm, torch.float32)
        if t.numel() != n:
            raise _lib.GeodiffError(f"loss_assemble: {nm} must have {n} entries")
    out = torch.empty(12, dtype=torch.float32, device=sums.device)
    check(lib.gd_loss_assemble(_p(sums), _p(rm), _p(inv5), _p(inv_rm), _p(wv), _p(inv5_bwd), int(bool(use_amodal)), _p(out), _stream()),
          "gd_loss_assemble")
    return out[0:5], out[5], out[6:11], out[11:12]


def edit_losses_fwd(eo, ro, tgt, m_wo, m_edit, w_am, m_amodal, S: int):
    """The five loss reductions only (gd_edit_losses_fwd with wv == NULL) -> sums [5] f32."""
    lib = _lib.load()
    dt = _dt16(eo, "eo")
    _need(eo, "eo"); _need(ro, "ro", eo.dtype)
    H, N, D = eo.shape
    sums = zeros_f32(5, eo.device)
    ws = torch.empty(lib.gd_edit_losses_fwd_workspace_bytes(H, S, D) // 4, dtype=torch.float32, device=eo.device)
    a = GdEditLosses()
    a.eo, a.ro, a.tgt, a.m_wo, a.m_edit, a.w_am, a.m_amodal = eo.data_ptr(), ro.data_ptr(), _ip(tgt), m_wo.data_ptr(), m_edit.data_ptr(), _ip(w_am), _ip(m_amodal)
    a.out12, a.workspace, a.H, a.S, a.D = sums.data_ptr(), ws.data_ptr(), H, S, D
    check(lib.gd_edit_losses_fwd(ctypes.byref(a), dt, _stream()), "gd_edit_losses_fwd")
    return sums


def _gout_flag(gout, eo, blend, gout_tok: bool) -> int:
    """blend bit 0 + the layout bit of gout (bit 1: the token-major row [1, N, H*D] of the layer's output gradient)."""
    H, N, D = eo.shape
    if gout is not None:
        _need(gout, "gout", eo.dtype)
        if gout.numel() != eo.numel() or (gout_tok and tuple(gout.shape[-2:]) != (N, H * D)) or ((not gout_tok) and gout.shape != eo.shape):
            raise _lib.GeodiffError("edit_losses_bwd: gout must be [H, N, D] (or [1, N, H*D] with gout_tok)")
    return int(bool(blend)) | (2 if (gout_tok and gout is not None) else 0)


def edit_losses_bwd(eo, ro, tgt, m_wo, m_edit, w_am, m_amodal, gout, coefs, gscale, blend: bool, S: int, gout_tok: bool = False, out=None):
    """coefs: device f32 [5] tensor (or a host sequence, uploaded here — not capture-safe)."""
    if not isinstance(coefs, torch.Tensor):
        coefs = torch.tensor([float(x) for x in coefs], dtype=torch.float32, device=eo.device)
    return edit_losses_bwd_rowdot(eo, ro, tgt, m_wo, m_edit, w_am, m_amodal, gout, coefs, gscale, blend, S, None, gout_tok, out=out)


def blend_tokens(a, b, m, out=None):
    """out = a*m + b*(1-m) per token (U/attention_processors.py:504,619), op by op in the tensor dtype: gd_blend_merge without a row list."""
    if out is None:
        out = torch.empty_like(a)
    blend_merge(a, None, None, b, m, eo_out=None, out=out)
    return out


# ---------------------------------------------------------------------------------------------------
# the launches of one hooked optimisation-pass layer (include/geodiff_hip.h R8: the optional fields of the loss / backward entry points)
# ---------------------------------------------------------------------------------------------------
def _ip(t):
    return 0 if t is None else t.data_ptr()


_TICKETS = {}        # (thread, device index) -> one zeroed int32 (the arrival ticket of gd_edit_losses_fwd's tail: zero before every launch, left zero by it)


def _ticket(dev: torch.device) -> torch.Tensor:
    tkey = (threading.get_ident(), dev.index)
    t = _TICKETS.get(tkey)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.GeodiffError("edit_losses_fused: the arrival ticket must exist before a graph capture (run one eager pass first)")
        t = _TICKETS[tkey] = torch.zeros(4, dtype=torch.int32, device=dev)
    return t


def blend_merge(base, act, pos, ro, m, eo_out=None, out=None):
    """gd_blend_merge: e = act[h, pos[n]] where pos[n] >= 0 (act given) else base[h, n]; eo_out = e;
    out = e*m + ro*(1-m).  Either output may be None."""
    lib = _lib.load()
    dt = _dt16(base, "base")
    _need(base, "base")
    H, N, D = base.shape
    R = 0
    if act is not None:
        _need(act, "act", base.dtype); _need(pos, "pos", torch.int32)
        R = act.shape[1]
        if act.shape[0] != H or act.shape[2] != D or pos.numel() != N:
            raise _lib.GeodiffError("blend_merge: shapes disagree")
    for t, nm in ((ro, "ro"), (eo_out, "eo_out"), (out, "out")):
        if t is not None:
            _need(t, nm, base.dtype)
            if t.shape != base.shape:
                raise _lib.GeodiffError(f"blend_merge: {nm} must have base's shape")
    if m is not None:
        _need(m, "m", torch.float32)
    check(lib.gd_blend_merge(_p(base), _p(act), _p(pos) if act is not None else None, _p(ro), _p(m), H, N, R, D, _p(eo_out), _p(out), dt, _stream()),
          "gd_blend_merge")


def heads_split(tensors: Sequence[torch.Tensor], heads: int) -> List[torch.Tensor]:
    """Token-major [B, rows_i, heads*D] -> head-major [B*heads, rows_i, D] for up to six tensors of one batch size in ONE launch
    (head_to_batch_dim of q, k and v: U/attention_processors.py:118-120,201-203)."""
    lib = _lib.load()
    t0 = tensors[0]
    dt = _dt16(t0, "tensor 0")
    if not 1 <= len(tensors) <= 6:
        raise _lib.GeodiffError("heads_split: 1..6 tensors")
    B, _, C = t0.shape
    if C % heads or (C // heads) % 8:
        raise _lib.GeodiffError("heads_split: channels must be heads x (a multiple of 8)")
    D = C // heads
    a = GdHeadsSplit()
    outs = []
    for i, t in enumerate(tensors):
        _need(t, f"tensor {i}", t0.dtype)
        if t.dim() != 3 or t.shape[0] != B or t.shape[2] != C:
            raise _lib.GeodiffError("heads_split: tensors must be [B, rows, heads*D] with one B and one width")
        o = torch.empty(B * heads, t.shape[1], D, dtype=t.dtype, device=t.device)
        a.src[i], a.dst[i], a.rows[i] = t.data_ptr(), o.data_ptr(), t.shape[1]
        outs.append(o)
    a.n, a.B, a.heads, a.D = len(tensors), B, heads, D
    check(lib.gd_heads_split(ctypes.byref(a), dt, _stream()), "gd_heads_split")
    return outs


def heads_merge(srcs: Sequence[Optional[torch.Tensor]], heads: int, rows: int, D: int, dtype: torch.dtype, device, blend=None, out=None):
    """-> token-major [B, rows, heads*D]: batch row b from the head-major srcs[b] [heads, rows, D] (16-bit, or f32 for all: rounded once),
    zeros where srcs[b] is None.  blend = (row, b, m) or a list of such triples: that row is srcs[row]*m + b*(1-m) as blend_tokens
    computes it (every edit of a batch blends with its own mask).  One launch (batch_to_head_dim of the output, the blend before it, and
    the autograd of head_to_batch_dim with its zero rows); 16 rows per launch."""
    lib = _lib.load()
    B = len(srcs)
    if B < 1:
        raise _lib.GeodiffError("heads_merge: at least one batch row")
    if out is None:
        out = torch.empty(B, rows, heads * D, dtype=dtype, device=device)
    elif tuple(out.shape) != (B, rows, heads * D) or out.dtype != dtype:
        raise _lib.GeodiffError("heads_merge: out must be [B, rows, heads*D] of the tensor dtype")
    if B > 16:                                             # GD_HEADS_MERGE_MAX_ROWS per launch: 16 rows at a time into slices of `out`
        triples = [] if blend is None else ([blend] if isinstance(blend, tuple) else list(blend))
        for i in range(0, B, 16):
            part = [(row - i, bb, m) for row, bb, m in triples if i <= row < i + 16]
            heads_merge(srcs[i:i + 16], heads, rows, D, dtype, device, blend=part or None, out=out[i:i + 16])
        return out
    dt = _dt16(out, "out")
    a = GdHeadsMerge()
    f32 = None
    for b, t in enumerate(srcs):
        if t is None:
            a.src[b] = None
            continue
        is32 = t.dtype == torch.float32
        if f32 is None:
            f32 = is32
        _need(t, f"source {b}", torch.float32 if f32 else dtype)
        if tuple(t.shape) != (heads, rows, D):
            raise _lib.GeodiffError("heads_merge: sources must be [heads, rows, D]")
        a.src[b] = t.data_ptr()
    if blend is not None:
        for row, bb, m in ([blend] if isinstance(blend, tuple) else blend):
            _need(bb, "blend b", dtype); _need(m, "blend m", torch.float32)
            if tuple(bb.shape) != (heads, rows, D) or m.numel() != rows or f32 or srcs[row] is None:
                raise _lib.GeodiffError("heads_merge: blend operands disagree")
            a.blend_b[row], a.m[row] = bb.data_ptr(), m.data_ptr()
    a.out, a.src_f32, a.B, a.rows, a.heads, a.D = out.data_ptr(), int(bool(f32)), B, rows, heads, D
    check(lib.gd_heads_merge(ctypes.byref(a), dt, _stream()), "gd_heads_merge")
    return out


def attn_probs_pair(qb, kb, lse_b, qe, ke, lse_e, rows, n_valid, scale: float, zero: Optional[torch.Tensor] = None):
    """-> (Pb [BH, N, Mpad_b] over all rows of (qb, kb), Pe [BH, R, Mpad_e] over rows[...] of (qe, ke)) in ONE launch, which also clears
    ``zero`` (the `best` scratch of removal_corr_max_nz)."""
    lib = _lib.load()
    dt = _dt16(qb, "qb")
    for t, nm in ((qb, "qb"), (kb, "kb"), (qe, "qe"), (ke, "ke")):
        _need(t, nm, qb.dtype)
    _need(lse_b, "lse_b", torch.float32); _need(lse_e, "lse_e", torch.float32); _need(rows, "rows", torch.int32)
    if n_valid is not None:
        _need(n_valid, "n_valid", torch.int32)
    BH, N, D = qb.shape
    Mb, Me = kb.shape[1], ke.shape[1]
    R = rows.numel()
    Pb = torch.empty(BH, N, (Mb + 7) // 8 * 8, dtype=qb.dtype, device=qb.device)
    Pe = torch.empty(BH, R, (Me + 7) // 8 * 8, dtype=qb.dtype, device=qb.device)
    a = GdProbs(qb.data_ptr(), kb.data_ptr(), lse_b.data_ptr(), 0, 0, Pb.data_ptr(), BH, N, N, Mb, Pb.shape[2])
    b = GdProbs(qe.data_ptr(), ke.data_ptr(), lse_e.data_ptr(), rows.data_ptr(), _ip(n_valid), Pe.data_ptr(), BH, N, R, Me, Pe.shape[2])
    zb = 0
    if zero is not None:
        _need(zero, "zero")
        zb = zero.numel() * zero.element_size()
    check(lib.gd_attn_probs(ctypes.byref(a), ctypes.byref(b), D, scale, _p(zero), zb, dt, _stream()), "gd_attn_probs")
    return Pb, Pe


def removal_corr_max_nz(Pe, Pb, m_inp, m_wo, n_valid, best):
    """The correlation + masked arg-max into a ``best`` [H, R, 2] int64 scratch the CALLER cleared (attn_probs_pair's ``zero``)."""
    lib = _lib.load()
    dt = _dt16(Pe, "Pe")
    _need(Pe, "Pe"); _need(Pb, "Pb", Pe.dtype); _need(m_inp, "m_inp", torch.float32); _need(m_wo, "m_wo", torch.float32)
    _need(best, "best", torch.int64)
    H, R, Mpad = Pe.shape
    check(lib.gd_removal_corr_max(_p(Pe), _p(Pb), _p(m_inp), _p(m_wo), _p(n_valid), H, R, Pb.shape[1], Mpad, _p(best), 0, 0, dt, _stream()),
          "gd_removal_corr_max")


def edit_losses_fused(eo, ro, tgt, m_wo, m_edit, w_am, m_amodal, S: int, best, rows, n_valid, inv5, inv_rm, wv, inv5_bwd, use_amodal: bool,
                      log_acc=None, running=False, loss_out=None):
    """edit_losses_fwd + removal_loss_reduce + the fold + loss_assemble in one launch.
    -> (terms [5], loss (), coefs [5], rm_coef [1], aux | None) — what loss_assemble and removal_fwd return.
    log_acc (f32 [>= 4], updated in place): += the four logged terms.  running: False = no running loss; None or a 0-d / [1] f32 tensor =
    the controller's loss so far (None: 0) — a sixth result, running + loss, is appended (written to ``loss_out`` [1] f32 if given)."""
    lib = _lib.load()
    dt = _dt16(eo, "eo")
    _need(eo, "eo"); _need(ro, "ro", eo.dtype)
    H, N, D = eo.shape
    dev = eo.device
    ws = torch.empty(lib.gd_edit_losses_fwd_workspace_bytes(H, S, D) // 4, dtype=torch.float32, device=dev)
    out = torch.empty(13 if running is not False else 12, dtype=torch.float32, device=dev)
    if log_acc is not None:
        _need(log_acc, "log_acc", torch.float32)
    if running is not None and running is not False:
        _need(running, "running", torch.float32)
    aux = None
    R = 0
    if best is not None:
        _need(best, "best", torch.int64); _need(rows, "rows", torch.int32)
        Hr, R = best.shape[0], best.shape[1]
        if Hr != H:
            raise _lib.GeodiffError("edit_losses_fused: best must have eo's head count")
        f32 = torch.empty(3, H, R, dtype=torch.float32, device=dev)
        i32 = torch.empty(2, H, R, dtype=torch.int32, device=dev)
        aux = dict(p_in=f32[0], p_wo=f32[1], wgt=f32[2], j_in=i32[0], j_wo=i32[1])
    a = GdEditLosses(eo.data_ptr(), ro.data_ptr(), _ip(tgt), m_wo.data_ptr(), m_edit.data_ptr(), _ip(w_am), _ip(m_amodal),
                     _ip(best), _ip(rows) if best is not None else 0, _ip(n_valid) if best is not None else 0,
                     _ip(aux["p_in"]) if aux else 0, _ip(aux["j_in"]) if aux else 0, _ip(aux["p_wo"]) if aux else 0,
                     _ip(aux["j_wo"]) if aux else 0, _ip(aux["wgt"]) if aux else 0,
                     inv5.data_ptr(), inv_rm.data_ptr(), wv.data_ptr(), inv5_bwd.data_ptr(),
                     out.data_ptr(), ws.data_ptr(), _ticket(dev).data_ptr(),
                     _ip(log_acc), 0 if running is False else _ip(running),
                     0 if running is False else (loss_out.data_ptr() if loss_out is not None else out[12:].data_ptr()),
                     H, S, D, R, int(bool(use_amodal)))
    check(lib.gd_edit_losses_fwd(ctypes.byref(a), dt, _stream()), "gd_edit_losses_fwd")
    if running is not False:
        return out[0:5], out[5], out[6:11], out[11:12], aux, (loss_out if loss_out is not None else out[12])
    return out[0:5], out[5], out[6:11], out[11:12], aux


def removal_bwd_args(Pe, Pb, q, k, rows, aux, m_inp, m_wo, coef: float, gscale, gscale2, scale: float, n_valid, need_dk: bool):
    """-> (GdRemovalBwd, workspace tensor): the argument block the three removal-backward pieces share."""
    lib = _lib.load()
    H, R, Mpad = Pe.shape
    N, D = q.shape[1], q.shape[2]
    M = k.shape[1]
    ws = torch.empty(lib.gd_removal_bwd_workspace_bytes(H, R, M, Mpad, D, int(need_dk)) // 4, dtype=torch.float32, device=Pe.device)
    a = GdRemovalBwd(Pe.data_ptr(), Pb.data_ptr(), q.data_ptr(), k.data_ptr(), rows.data_ptr(),
                     aux["p_in"].data_ptr(), aux["j_in"].data_ptr(), aux["p_wo"].data_ptr(), aux["j_wo"].data_ptr(), aux["wgt"].data_ptr(),
                     m_inp.data_ptr(), m_wo.data_ptr(), _ip(gscale), _ip(gscale2), _ip(n_valid), 0, ws.data_ptr(),
                     float(coef), float(scale), H, R, N, M, Mpad, D, 0)
    return a, ws


def edit_losses_bwd_rowdot(eo, ro, tgt, m_wo, m_edit, w_am, m_amodal, gout, coefs, gscale, blend: bool, S: int, rm, gout_tok: bool = False, out=None):
    """edit_losses_bwd's grid + the removal backward's row dots (into rm's workspace) in one launch; rm None: plain edit_losses_bwd.
    out: write d(loss)/d(ro) there (ro's shape and dtype, contiguous: e.g. one edit's slice of a batch's gradient)."""
    lib = _lib.load()
    dt = _dt16(eo, "eo")
    H, N, D = eo.shape
    blend = _gout_flag(gout, eo, blend, gout_tok)
    if out is not None:
        _need(out, "out", ro.dtype)
        if out.shape != ro.shape:
            raise _lib.GeodiffError("edit_losses_bwd: out must have ro's shape")
    dro = out if out is not None else torch.empty_like(ro)
    _need(coefs, "coefs", torch.float32)
    check(lib.gd_edit_losses_bwd(_p(eo), _p(ro), _p(tgt), _p(m_wo), _p(m_edit), _p(w_am), _p(m_amodal), _p(gout), _p(coefs), _p(gscale),
                                 int(blend), H, S, D, _p(dro), ctypes.byref(rm) if rm is not None else None, dt, _stream()),
          "gd_edit_losses_bwd")
    return dro


def attn_bwd_nofold(q, k, v, out, lse, dout, scale: float, need_dk: bool, dq_out):
    """attn_bwd that leaves the per-key-run dq partials of a split launch to edit_dq_fold.
    -> (dk f32 | None, kchunks, address of the partials, workspace tensor that owns them)."""
    lib = _lib.load()
    dt = _dt16(q, "q")
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout"), (dq_out, "dq_out")):
        _need(t, nm, q.dtype)
    _need(lse, "lse", torch.float32)
    BH, N, D = q.shape
    M = k.shape[1]
    dk = torch.empty(BH, M, D, dtype=torch.float32, device=q.device) if need_dk else None
    nbytes = lib.gd_attn_bwd_workspace_bytes(BH, N, M, D, int(need_dk), 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device) if nbytes else None
    kc = ctypes.c_int(0)
    part = ctypes.c_void_p(0)
    check(lib.gd_attn_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), BH, N, M, D, scale, _p(dq_out), _p(dk), _p(ws), nbytes,
                          ctypes.byref(kc), ctypes.byref(part), 0, dt, _stream()), "gd_attn_bwd (partials left)")
    return dk, int(kc.value), part.value, ws


def removal_bwd_nofold(rm, dtype: torch.dtype):
    """The dS K products of the removal backward (row dots: edit_losses_bwd_rowdot; partials folded by edit_dq_fold)."""
    lib = _lib.load()
    check(lib.gd_removal_bwd(ctypes.byref(rm), None, None, _DT[dtype], _stream()), "gd_removal_bwd (products only)")


def edit_dq_fold(dq_part_ptr, kchunks: int, BH: int, N: int, D: int, rm_ws, M: int, R: int, inp_pos, wgt, dq16, head0: int = 0, nheads: int = 0):
    """dq16 = T(sum of the attention partials (dq_part_ptr None / 0: dq16 itself) + the removal partials of live inpaint rows): one rounding.
    head0 / nheads: fold only heads [head0, head0 + nheads) of the BH-head backward (one edit of a batch, whose removal partials rm_ws are
    its own): dq16 is still the full [BH, N, D] tensor."""
    lib = _lib.load()
    dt = _dt16(dq16, "dq16")
    _need(dq16, "dq16")
    if rm_ws is not None:
        _need(inp_pos, "inp_pos", torch.int32); _need(wgt, "wgt", torch.float32)
    nh = nheads if nheads else BH
    off = head0 * N * D
    part = ctypes.c_void_p(dq_part_ptr + off * 4) if dq_part_ptr else None
    dq_ptr = ctypes.c_void_p(dq16.data_ptr() + off * dq16.element_size())
    check(lib.gd_edit_dq_fold(part, kchunks, BH * N * D if nheads else 0, nh, N, D, _p(rm_ws), M, R, _p(inp_pos) if rm_ws is not None else None,
                              _p(wgt) if rm_ws is not None else None, dq_ptr, dt, _stream()), "gd_edit_dq_fold")


def softsplat_fwd(tenIn, tenFlow):
    """in [N,C,H,W] f32, flow [N,2,H,W] f32 -> summation splat [N,C,H,W] f32."""
    lib = _lib.load()
    _need(tenIn, "tenIn", torch.float32); _need(tenFlow, "tenFlow", torch.float32)
    N, C, H, W = tenIn.shape
    if tuple(tenFlow.shape) != (N, 2, H, W):
        raise _lib.GeodiffError("softsplat: flow must be [N, 2, H, W]")
    out = torch.empty_like(tenIn)
    check(lib.gd_softsplat_fwd(_p(tenIn), _p(tenFlow), N, C, H, W, _p(out), _stream()), "gd_softsplat_fwd")
    return out


def softsplat_bwd(tenIn, tenFlow, outgrad, need_in: bool, need_flow: bool):
    lib = _lib.load()
    _need(outgrad, "outgrad", torch.float32)
    N, C, H, W = tenIn.shape
    gi = torch.empty_like(tenIn) if need_in else None
    gf = torch.empty_like(tenFlow) if need_flow else None
    if gi is None and gf is None:
        return None, None
    check(lib.gd_softsplat_bwd(_p(tenIn), _p(tenFlow), _p(outgrad), N, C, H, W, _p(gi), _p(gf), _stream()), "gd_softsplat_bwd")
    return gi, gf


def hist_match(src, tmpl, m_src, m_tmpl):
    """src, tmpl [npix, C] uint8; m_src, m_tmpl [npix] uint8 -> (out [npix, C] f64, lut [C,256] f64, counts [2,C,256] i32)."""
    lib = _lib.load()
    for t, nm in ((src, "src"), (tmpl, "tmpl"), (m_src, "m_src"), (m_tmpl, "m_tmpl")):
        _need(t, nm, torch.uint8)
    npix, C = src.shape
    if tmpl.shape != src.shape or m_src.numel() != npix or m_tmpl.numel() != npix:
        raise _lib.GeodiffError("hist_match: shapes disagree")
    counts = torch.empty(2, C, 256, dtype=torch.int32, device=src.device)
    lut = torch.empty(C, 256, dtype=torch.float64, device=src.device)
    out = torch.empty(npix, C, dtype=torch.float64, device=src.device)
    check(lib.gd_hist_match(_p(src), _p(tmpl), _p(m_src), _p(m_tmpl), npix, C, _p(counts), _p(lut), _p(out), _stream()), "gd_hist_match")
    return out, lut, counts


# ---------------------------------------------------------------------------------------------------
# R10-R12 scheduler / latent arithmetic
# ---------------------------------------------------------------------------------------------------
def ddim_step(x, eps_u, eps_c, guidance: float, a_t: float, a_to: float, out=None, v_prediction: bool = False):
    """CFG combine + DDIM closed form; ``v_prediction``: eps_u / eps_c hold the v-prediction of an SD2.1-768-style UNet."""
    lib = _lib.load()
    _need(x, "x"); _need(eps_u, "eps_u", x.dtype)
    if eps_c is not None:
        _need(eps_c, "eps_c", x.dtype)
    if out is None:
        out = torch.empty_like(x)
    fn = lib.gd_ddim_step_v if v_prediction else lib.gd_ddim_step
    check(fn(_p(x), _p(eps_u), _p(eps_c), guidance, a_t, a_to, _p(out), x.numel(), _DT[x.dtype], _stream()), "gd_ddim_step")
    return out


def masked_latent_update(x, g, m, step: float):
    """x, g [C,h,w] f32; m [h*w] f32."""
    lib = _lib.load()
    _need(x, "x", torch.float32); _need(g, "g", torch.float32); _need(m, "m", torch.float32)
    C = x.shape[-3]
    hw = x.shape[-1] * x.shape[-2]
    out = torch.empty_like(x)
    check(lib.gd_masked_latent_update(_p(x), _p(g), _p(m), step, C * (x.numel() // (C * hw)), hw, _p(out), _stream()),
          "gd_masked_latent_update")
    return out


def sumsq(x):
    lib = _lib.load()
    _need(x, "x", torch.float32)
    acc = zeros_f32(1, x.device)
    check(lib.gd_sumsq(_p(x), x.numel(), _p(acc), _stream()), "gd_sumsq")
    return acc


def norm_rescale(x, num_sumsq, den_sumsq):
    lib = _lib.load()
    _need(x, "x", torch.float32)
    out = torch.empty_like(x)
    check(lib.gd_norm_rescale(_p(x), _p(num_sumsq), _p(den_sumsq), x.numel(), _p(out), _stream()), "gd_norm_rescale")
    return out


# ---------------------------------------------------------------------------------------------------
# UNet plumbing: fused GroupNorm (+SiLU), channels-last, no-grad
# ---------------------------------------------------------------------------------------------------
def _add_ld(add_bc, x, B, C):
    if add_bc is None:
        return 0
    # [B, C], rows may be strided (a column slice of a wider [B, sum C] matrix)
    if (add_bc.dtype != x.dtype or not add_bc.is_cuda or tuple(add_bc.shape) != (B, C) or add_bc.stride(1) != 1
            or add_bc.stride(0) % 8 or add_bc.storage_offset() % 8):
        raise _lib.GeodiffError("group_norm_nhwc: add_bc must be a 16-byte aligned [B, C] matrix of x's dtype with unit column stride")
    return add_bc.stride(0) if B > 1 else C


def group_norm_nhwc_bwd(x, add_bc, gamma, beta, dy, groups: int, eps: float, silu: bool, fwd_scratch):
    """dx of group_norm_nhwc for frozen gamma / beta; fwd_scratch = the scratch the forward returned."""
    lib = _lib.load()
    dt = _dt16(x, "x")
    B, C, H, W = x.shape
    if not (x.is_contiguous(memory_format=torch.channels_last) and dy.is_contiguous(memory_format=torch.channels_last)) or dy.dtype != x.dtype:
        raise _lib.GeodiffError("group_norm_nhwc_bwd: expected channels_last x and dy of one dtype")
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    scratch = torch.empty_like(fwd_scratch)
    check(lib.gd_group_norm_nhwc_bwd(_p(x), _p(add_bc), _add_ld(add_bc, x, B, C), _p(gamma), _p(beta), _p(dy), B, H * W, C, groups, eps,
                                     int(silu), GN_SINGLE_LAUNCH, _p(fwd_scratch), _p(scratch), _p(dx), dt, _stream()), "gd_group_norm_nhwc_bwd")
    return dx


def group_norm_nhwc(x, gamma, beta, groups: int, eps: float, silu: bool, add_bc=None, return_scratch: bool = False):
    """x [B,C,H,W] in channels_last memory format (16-bit) -> same shape / format; add_bc [B,C]: norm of x + add_bc[:, :, None, None]."""
    lib = _lib.load()
    dt = _dt16(x, "x")
    if not x.is_cuda or not x.is_contiguous(memory_format=torch.channels_last):
        raise _lib.GeodiffError("group_norm_nhwc: expected a channels_last GPU tensor")
    B, C, H, W = x.shape
    add_ld = _add_ld(add_bc, x, B, C)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    scratch = torch.empty(int(lib.gd_group_norm_nhwc_scratch_floats(B, H * W, groups)), dtype=torch.float32, device=x.device)
    check(lib.gd_group_norm_nhwc(_p(x), _p(add_bc), add_ld, _p(gamma), _p(beta), B, H * W, C, groups, eps, int(silu), GN_SINGLE_LAUNCH, _p(scratch), _p(y),
                                 dt, _stream()), "gd_group_norm_nhwc")
    return (y, scratch) if return_scratch else y


def _rows_c(t, what):
    _dt16(t, what)
    if not t.is_cuda:
        raise _lib.GeodiffError(f"{what}: expected a GPU tensor")
    if t.dim() == 4:                       # [B,C,H,W] channels_last == [B*H*W, C] rows
        if not t.is_contiguous(memory_format=torch.channels_last):
            raise _lib.GeodiffError(f"{what}: expected channels_last memory")
        return t.shape[0] * t.shape[2] * t.shape[3], t.shape[1]
    if not t.is_contiguous():
        raise _lib.GeodiffError(f"{what}: expected a contiguous tensor")
    return t.numel() // t.shape[-1], t.shape[-1]


def bias_residual(x, bias, res=None):
    """y = x + bias[c] (+ res); x / res either [B,C,H,W] channels_last or [..., C] contiguous."""
    lib = _lib.load()
    rows, C = _rows_c(x, "bias_residual x")
    _need(bias, "bias", x.dtype)
    if res is not None and (_rows_c(res, "bias_residual res") != (rows, C) or res.dtype != x.dtype):
        raise _lib.GeodiffError("bias_residual: res must match x")
    y = torch.empty_like(x, memory_format=torch.channels_last) if x.dim() == 4 else torch.empty_like(x)
    check(lib.gd_bias_residual(_p(x), _p(bias), _p(res), rows, C, _p(y), _DT[x.dtype], _stream()), "gd_bias_residual")
    return y


def geglu(x):
    """x [..., 2C] contiguous -> x[..., :C] * gelu(x[..., C:])."""
    lib = _lib.load()
    rows, C2 = _rows_c(x, "geglu x")
    if x.dim() == 4 or C2 % 16:
        raise _lib.GeodiffError("geglu: expected [..., 2C] with C % 8 == 0")
    y = torch.empty(*x.shape[:-1], C2 // 2, dtype=x.dtype, device=x.device)
    check(lib.gd_geglu(_p(x), rows, C2 // 2, _p(y), _DT[x.dtype], _stream()), "gd_geglu")
    return y


def layer_norm_bwd(s, gamma, gy, gs, eps: float):
    """ds = dLayerNorm(gy | s) (+ gs): gradient of ``add_layer_norm`` w.r.t. its sum s (frozen gamma / beta)."""
    lib = _lib.load()
    rows, C = _rows_c(s, "layer_norm_bwd s")
    _need(gamma, "gamma", s.dtype); _need(gy, "gy", s.dtype)
    if gy.shape != s.shape or (gs is not None and (gs.shape != s.shape or gs.dtype != s.dtype or not gs.is_contiguous())):
        raise _lib.GeodiffError("layer_norm_bwd: gy / gs must match s")
    ds = torch.empty_like(s)
    check(lib.gd_layer_norm_bwd(_p(s), _p(gamma), _p(gy), _p(gs), rows, C, eps, _p(ds), _DT[s.dtype], _stream()), "gd_layer_norm_bwd")
    return ds


def geglu_bwd(x, dy):
    """Gradient of ``geglu`` w.r.t. x [..., 2C] for dy [..., C]."""
    lib = _lib.load()
    rows, C2 = _rows_c(x, "geglu_bwd x")
    _need(dy, "dy", x.dtype)
    if x.dim() == 4 or C2 % 16 or dy.shape != x.shape[:-1] + (C2 // 2,):
        raise _lib.GeodiffError("geglu_bwd: expected x [..., 2C] (C % 8 == 0) and dy [..., C]")
    dx = torch.empty_like(x)
    check(lib.gd_geglu_bwd(_p(x), _p(dy), rows, C2 // 2, _p(dx), _DT[x.dtype], _stream()), "gd_geglu_bwd")
    return dx


def add_layer_norm(a, b, gamma, beta, eps: float):
    """-> (s, y): s = a + b (a itself when b is None), y = LayerNorm(s)."""
    lib = _lib.load()
    rows, C = _rows_c(a, "add_layer_norm a")
    _need(gamma, "gamma", a.dtype); _need(beta, "beta", a.dtype)
    if b is not None and (b.shape != a.shape or b.dtype != a.dtype or not b.is_contiguous()):
        raise _lib.GeodiffError("add_layer_norm: b must match a")
    y = torch.empty_like(a)
    s = torch.empty_like(a) if b is not None else a
    check(lib.gd_add_layer_norm(_p(a), _p(b), _p(gamma), _p(beta), rows, C, eps, _p(s) if b is not None else None, _p(y), _DT[a.dtype],
                                _stream()), "gd_add_layer_norm")
    return s, y


# ---------------------------------------------------------------------------------------------------
# UNet harness: 3x3 convolution on the matrix cores (conv3x3.hip)
# ---------------------------------------------------------------------------------------------------
def conv3x3_supported(x, w, stride=1) -> bool:
    """True when ``conv3x3`` takes this call: 16-bit channels_last activations [n, C, H, W] and weight [K, C, 3, 3], C % 64 == 0,
    K % 8 == 0, stride 1 / 2."""
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and w.dtype == x.dtype and x.dim() == 4 and w.dim() == 4
            and tuple(w.shape[2:]) == (3, 3) and w.shape[1] == x.shape[1] and x.shape[1] % 64 == 0 and w.shape[0] % 8 == 0
            and stride in (1, 2) and x.is_contiguous(memory_format=torch.channels_last)
            and w.is_contiguous(memory_format=torch.channels_last) and x.numel() * 2 < 0x7FFFFFFF and w.numel() * 2 < 0x7FFFFFFF)


def conv3x3(x, w, bias=None, stride: int = 1, upsample: bool = False, res=None):
    """3x3 convolution, padding 1 (F.conv2d(x, w, bias, stride, 1); with ``upsample`` over F.interpolate(x, 2.0, 'nearest') without
    building it) [+ res, added before the result is rounded].  x [n, C, H, W], res and the result [n, K, Ho, Wo] are channels_last
    (NHWC memory)."""
    lib = _lib.load()
    if not conv3x3_supported(x, w, stride):
        raise _lib.GeodiffError("conv3x3: needs 16-bit channels_last x [n,C,H,W] / w [K,C,3,3] with C % 64 == 0, K % 8 == 0")
    n, C, H, W = x.shape
    K = w.shape[0]
    if upsample:
        if stride != 1:
            raise _lib.GeodiffError("conv3x3: upsample needs stride 1")
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if bias is not None:
        _need(bias, "bias", x.dtype)
    if res is not None and (tuple(res.shape) != (n, K, Ho, Wo) or res.dtype != x.dtype or not res.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.GeodiffError("conv3x3: res must be a channels_last tensor of the output's shape and dtype")
    out = torch.empty((n, K, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    c = CONV3X3_CFG
    cfg = _lib.GdConv3x3Cfg(int(c["pi"]), int(c["ki"]), int(c["ksplit"]), int(c["dma"]))
    nb = int(lib.gd_conv3x3_workspace_bytes(n, Ho, Wo, C, K, ctypes.byref(cfg)))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
    check(lib.gd_conv3x3(x.data_ptr(), w.data_ptr(), _p(bias), _p(res), out.data_ptr(), n, H, W, C, K, stride, int(bool(upsample)),
                         ctypes.byref(cfg), _p(ws), nb, _DT[x.dtype], _stream()), "gd_conv3x3")
    return out
