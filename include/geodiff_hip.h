/*
 * geodiff_hip.h — C ABI of libgeodiff_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * Drop-in boundary for GeoDiffuser's geometry-guided attention-sharing hot path.  The reference has
 * no FFI of its own (it is pure Python on torch + pytorch3d); each entry point below replaces the
 * torch / pytorch3d op sequence cited next to it (paths relative to the reference root,
 * U/ = GeoDiffuser/utils/).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns GD_OK (0) or a negative GD_E* code; nothing throws across the ABI;
 *     gd_last_error() gives a thread-local message for the last failure on this thread.
 *   - all buffers are CALLER-OWNED DEVICE pointers (e.g. torch Tensor.data_ptr()), contiguous in the
 *     documented layout; the library never allocates, never frees and never synchronises.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Calls are asynchronous,
 *     stateless and re-entrant per stream, and are safe to capture in a hipGraph.  STATELESS means
 *     it: there is no process-wide setting (ABI 5 removed the gd_*_set_* tuning hooks of ABI <= 4)
 *     and the library reads no environment variable; whatever selects a kernel variant travels in
 *     the call (gd_attn_cfg_t, gd_conv3x3_cfg_t, an int flag), NULL / -1 meaning "the launcher's
 *     own choice".  Two controllers on two streams of one process can therefore differ.
 *   - one entry point per operation: the fused launches of ABI 4 are the optional fields of the
 *     same call (e.g. gd_edit_losses_fwd's tail, gd_attn_bwd's partials left for gd_edit_dq_fold).
 *   - `dtype` selects the 16-bit storage type of feature tensors: GD_F16 or GD_BF16 (GD_F32 where
 *     noted).  Accumulation is always binary32.
 */
#ifndef GEODIFF_HIP_H
#define GEODIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_ABI_VERSION 6

enum { GD_F16 = 0, GD_BF16 = 1, GD_F32 = 2 };
enum { GD_TOKEN_MAJOR = 0 /* [B, P, C] */, GD_CHANNEL_MAJOR = 1 /* [B, C, P] */ };

enum {
    GD_OK = 0,
    GD_EINVAL = -1,      /* bad argument (null pointer, unsupported size / dtype) */
    GD_EWORKSPACE = -2,  /* workspace too small */
    GD_ELAUNCH = -3,     /* HIP launch failure (message in gd_last_error) */
    GD_EUNSUPPORTED = -4
};

int gd_version(void);
const char* gd_last_error(void);
const char* gd_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * R3  point splat.   Replaces pytorch3d rasterize_points + compositing.alpha_composite as called by
 *     RasterizePointsXYsBlending.forward (U/warp_utils.py:72-176) / warp_grid_edit (:798-825).
 * ---------------------------------------------------------------------------------------------- */

/* Bytes of scratch gd_rasterize_points needs for P points on an S x S grid. */
size_t gd_rasterize_workspace_bytes(int P, int S, float radius_ndc);

/*
 * U/warp_utils.py:111  rasterize_points(Pointclouds(pts), S, radius_ndc, K) for ONE cloud.
 *   pts   [P,3] f32  (x,y,z) in the rasterizer's convention (x,y already negated, :90-91)
 *   idx   [S,S,K] i32  point index or -1       (the bit-exact "integer warp-index grid")
 *   zbuf  [S,S,K] f32  (may be NULL)           dist2 [S,S,K] f32
 * Semantics: pixel (row,col) centre at NDC (1-(2col+1)/S, 1-(2row+1)/S); hit iff dist2 < r^2 and
 * z >= 0; K nearest in z, ascending, ties as documented in DESIGN.md ("Splat semantics" A1-A7).
 */
int gd_rasterize_points(const float* pts, int P, int S, float radius_ndc, int K,
                        int32_t* idx, float* zbuf, float* dist2,
                        void* workspace, size_t workspace_bytes, void* stream);

/*
 * U/warp_utils.py:131-140 + the compositor's transmittance:  alpha_k = (1-clamp(d2/r^rad_pow,1e-3,1)^.5)^tau,
 *   w[pix,k] = alpha_k * prod_{j<k}(1-alpha_j)   (0 for empty slots).   w [npix,K] f32.
 */
int gd_splat_weights(const int32_t* idx, const float* dist2, int npix, int K,
                     float radius_ndc, float rad_pow, float tau, float* w, void* stream);

/*
 * U/warp_utils.py:156-176 (alpha_composite, result rounded through fp16) fused with the blend of
 * U/attention_processors.py:424,544:
 *     out = src * (1 - m) + m * half(sum_k w[pix,k] * src[idx[pix,k]])        (m == NULL: out = half(sum))
 *   src/out: B clouds that share ONE idx/w table; layout GD_TOKEN_MAJOR [B,P,C] (attention queries,
 *   P == npix) or GD_CHANNEL_MAJOR [B,C,P] (latents / masks / images).  m [npix] f32.
 */
int gd_splat_composite(const void* src, const int32_t* idx, const float* w, const float* m,
                       int B, int P, int C, int npix, int K, int layout, void* out, int dtype, void* stream);

/*
 * Setup (once per edit), U/warp_utils.py:235-298 splatter_mesh: coverage of the transformed object surface mesh.
 *   verts [V,3] f32 (x,y in the rasterizer's NDC convention, z depth), faces [F,3] i32, out [S,S] f32 in {0,1}.
 */
int gd_mesh_coverage(const float* verts, const int32_t* faces, int V, int F, int S, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * R5/R6/R7  attention.  Replaces compute_attention (baddbmm + softmax, U/attention_sharing.py:30-47)
 *     followed by torch.bmm(attn, v) (U/attention_processors.py:428,433,549,557,644,647) without ever
 *     materialising the [BH,N,M] map.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const void* q;   /* [bh, N, D]            (heads == 0)   or   [B, N, heads*D]  token-major (heads > 0, bh = B*heads) */
    const void* k;   /* [bh, M, D]                           or   [B, M, heads*D] */
    const void* v;   /* [bh, M, D]                           or   [B, M, heads*D] */
    void* out;       /* [bh, N, D]                           or   [B, N, heads*D] */
    float* lse;      /* [bh, N]  natural-log sum-exp of the scaled scores (may be NULL) */
    int32_t bh;      /* batch*heads entries in this segment */
    int32_t heads;   /* 0: head-major (the reference's head_to_batch_dim layout); > 0: token-major as produced by to_q/to_k/to_v,
                        which saves the four head_to_batch_dim / batch_to_head_dim copies per attention layer */
    /* Fused geometric warp of the queries (the north star's "fused attention-warp kernel"), U/attention_processors.py:424-428,
     * 544-549: when warp_idx != NULL the segment attends with
     *     q_warp[n] = q[n] * (1 - m[n]) + m[n] * half(sum_k warp_w[n,k] * q[warp_idx[n,k]])
     * built in the kernel's prologue from the per-resolution splat tables (gd_rasterize_points / gd_splat_weights) instead of
     * reading a q_warp tensor written by gd_splat_composite: bit-identical to that two-launch path (same code, splat_common.hpp),
     * one launch and one HBM round trip of the queries less.  warp_m == NULL: no blend.  Needs N == number of table rows. */
    const int32_t* warp_idx;   /* [N, warp_K] i32 (packed point index, -1 = empty slot) or NULL */
    const float* warp_w;       /* [N, warp_K] f32 composite weights */
    const float* warp_m;       /* [N] f32 soft edit mask or NULL */
    int32_t warp_K;
    /* Nonzero: q already holds scale*log2(e)*q — the query projection applied that factor in its fp32 GEMM epilogue, BEFORE the one
     * rounding to 16 bits.  The kernels then take the scores as exponents of 2 directly (p = exp2(q.k - mu): one vector instruction
     * per probability instead of two on the vector-issue-bound D = 64 path) and `scale` is ignored.  All segments of a launch must
     * agree.  lse stays the natural-log sum-exp of the scaled scores.
     * 2 (as 1, and): the row sums are taken over the ROUNDED 16-bit probabilities the P.V products consume (on the matrix pipe, one extra
     * MFMA per 16-key step), so out = sum p~ v / sum p~ exactly: a dominant probability reproduces its value row whatever the softmax
     * reference is.  For launches whose outputs are compared with each other by L1 losses (the optimisation pass); lse then carries the
     * rounding of the probabilities (~1e-3 relative).  Honoured by the 64-query kernel (64^2-token launches); elsewhere the same as 1. */
    int32_t q_scaled;
    /* Query ROW LIST (at most GD_ATTN_MAX_ROWLIST_SEGS segments of a launch — one per edit of a batch; NULL: all N rows).  The segment attends only with rows q_rows[0 .. *q_rows_n) of q
     * (taken from the full [.., N, ..] tensor; the warp tables, if any, are indexed by the same row ids) and writes a DENSE result:
     * out [bh, q_rows_len, D] / [B, q_rows_len, heads*D], lse [bh, q_rows_len]; list slots >= *q_rows_n are padding (computed on row
     * q_rows[i] but not stored; the list is padded so that launch dimensions repeat from edit to edit).  Use: the edit attention
     * with warped queries (U/attention_processors.py:424-428,544-549) differs from the reference row's attention only where the soft edit
     * mask is non-zero (q*(1-m) + m*splat(q) == q for m == 0) — ~10 % of the rows; gd_blend_merge puts the two together. */
    const int32_t* q_rows;     /* [q_rows_len] i32 row ids, or NULL */
    const int32_t* q_rows_n;   /* DEVICE int32[1]: number of wanted list entries */
    int32_t q_rows_len;
} gd_attn_seg_t;

#define GD_ATTN_MAX_SEGS 12      /* vanilla rows + replace rows of B edits as two segments + one warped / row-list segment per edit (B <= 8) */
#define GD_ATTN_MAX_ROWLIST_SEGS 8

/* Per-call launch configuration of gd_attn_fwd (NULL = all defaults).  Replaces the process-wide gd_attn_fwd_set_even_split /
 * gd_attn_fwd_set_config hooks of ABI <= 4 and the GD_ATTN_* environment variables the library used to read. */
typedef struct gd_attn_cfg {
    int32_t even_split;  /* even split of the key tiles over the resident workgroups (needs `workspace`): -1 / 1 = where the launcher's
                            cost model says it pays (default), 0 = never, 2 = every launch that can be split */
    int32_t qb, ks;      /* which software-pipelined kernel serves launches with full key tiles (M % 128 == 0): a workgroup of QB query
                            blocks (32 rows each) x KS key ranges merged through LDS; (4,1), (2,2), (4,2), (2,4) exist, and (8,1) = the
                            64-queries-per-wave kernel (256-query workgroups, one wave per SIMD, direct-to-LDS staging; M % 256 == 0).
                            qb < 0: automatic (default), qb == 0: always the plain kernel */
    int32_t nsplit;      /* > 1: split-KV — the keys cut into nsplit ranges handled by separate workgroups whose un-normalised partial
                            results (O, reference max, row sum; f32) go through `workspace` (gd_attn_fwd_plan's size) and are merged by a
                            second small kernel; same result up to f32 summation order.  0 / 1: off (default) */
    int32_t handoff;     /* how the parts of an even split are handed over: 0 = agent-scope release / acquire fences around the ticket,
                            1 = device-scope stores and loads (default).  (Timing builds compiled with GD_MP_DBG also take 2 = no merge;
                            a release library returns GD_EINVAL for it: it produces wrong outputs by construction.) */
} gd_attn_cfg_t;
#define GD_ATTN_CFG_DEFAULT {-1, -1, 0, 0, 1}

/* out = softmax(scale * q k^T) v for up to GD_ATTN_MAX_SEGS independent segments in ONE launch (vanilla rows, edit_out with warped
 * queries, replace_out) that share N, M, D; replaces U/attention_sharing.py:30-47 + torch.bmm at
 * U/attention_processors.py:428,433,549,557,644,647.  D: 64, 128 or 192 (row lists, the even split and split-KV: 64 only).
 *
 * workspace (may be NULL): with cfg->nsplit <= 1 the EVEN SPLIT's scratch — gd_attn_fwd_workspace_bytes(sum of segment bh, N, M) bytes,
 * 256-byte aligned, ZERO before its first use (arrival counters; every launch leaves them zero), private to one stream at a time.  A
 * 64^2 launch of 20 heads is 640 units (head x 128-query tile) for 512 resident workgroups: 1.57 rounds; 5 heads fill 160 of 256 CUs.
 * With the workspace the launch's units x key tiles are dealt out as one linear range, the same number of key tiles to every workgroup
 * (launches of at most 128 units of 256 queries — the 5-head inversion pass — instead cut every unit into 2-4 parts, one per
 * workgroup); a unit that ends up in several workgroups is merged (un-normalised O, reference, row sum in f32, fixed part order:
 * bit-reproducible) by the workgroup that finishes last.  workspace == NULL, head dims other than 64, key counts that are not a
 * multiple of 256 and launches too short to split run unsplit.  With cfg->nsplit > 1 the workspace is split-KV's (gd_attn_fwd_plan). */
size_t gd_attn_fwd_workspace_bytes(int tot_bh, int N, int M);
/* the nsplit worth using for tot_bh = sum of segment bh (1 = do not split) and the workspace size split-KV needs for it */
int gd_attn_fwd_plan(int tot_bh, int N, int M, size_t* workspace_bytes);
int gd_attn_fwd(const gd_attn_seg_t* segs, int nseg, int N, int M, int D, float scale, const gd_attn_cfg_t* cfg, void* workspace,
                size_t workspace_bytes, int dtype, void* stream);

/* Short-key launches (M <= 128: the 77-key text context of every cross-attention layer, the 8^2 self-attention layer) with the blend
 * of U/attention_processors.py:502-508,617-622 (remover: :831-834) inside the launch.  segs[0 .. nseg-npair-1]: plain segments as
 * gd_attn_fwd; segs[nseg-npair+p] is side A of pair p and side_b[p] its side B (same bh / heads / layout for every pair; lse, row lists
 * unsupported; side B's `out` is ignored): side A's out = A*m + B*(1-m), A = attention(side A), B = attention(side B), both rounded to the
 * tensor dtype first and blended op by op like gd_blend_merge — bit-identical to gd_attn_fwd over nseg + npair segments followed by
 * gd_blend_merge.  blend_m[p]: [N] f32 mask of pair p (one pair per edit of a batch: npair <= GD_ATTN_MAX_PAIRS).  D = 64 only. */
#define GD_ATTN_MAX_PAIRS 8
int gd_attn_fwd_pair(const gd_attn_seg_t* segs, int nseg, const gd_attn_seg_t* side_b, const float* const* blend_m, int npair, int N, int M,
                     int D, float scale, int dtype, void* stream);

/*
 * Backward of out = softmax(scale q k^T) v w.r.t. q (always) and k (dk_f32 != NULL).
 *   dout [BH,N,D] 16-bit; lse from the forward; dq [BH,N,D] 16-bit (overwritten);
 *   dk_f32 [BH,M,D] f32 — used by cross-attention where k_edit carries gradient (U/attention_processors.py:432).  v never receives
 *   gradient on this path (v_base.detach(), :433,557).  workspace: gd_attn_bwd_workspace_bytes() bytes (may be 0 -> NULL): per-chunk dK
 *   partials when dk_f32 != NULL, and f32 dQ partials per key range when a launch of few workgroups is split over the keys to fill the
 *   chip; both are summed in a fixed order, without atomics.  D: 64, 128 or 192.
 * kchunks_out == NULL (and dq_part_out == NULL): the call is complete — dq is folded and rounded here, dk_f32 is ACCUMULATED into (the
 *   caller zeroes it).
 * kchunks_out != NULL (with dq_part_out): the dq partials of a split key range are LEFT in the workspace for gd_edit_dq_fold, which adds
 *   the removal loss's contribution before the ONE rounding: *kchunks_out = number of runs, *dq_part_out = their address inside
 *   `workspace` ([kchunks, BH, N, D] f32); kchunks == 1: dq (16-bit) was written directly.  dk_f32 is then OVERWRITTEN (0 + the sum: the
 *   caller needs no fill launch).
 * variant: 0 = the launcher's choice; 1 forces the register-staging dq kernel where the direct-to-LDS one would run (tests / benchmarks).
 */
size_t gd_attn_bwd_workspace_bytes(int BH, int N, int M, int D, int need_dk, int variant);
int gd_attn_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                const void* dout, int BH, int N, int M, int D, float scale,
                void* dq, float* dk_f32, void* workspace, size_t workspace_bytes, int* kchunks_out, float** dq_part_out,
                int variant, int dtype, void* stream);

/* dK and dV of out = softmax(scale q k^T) v for ANY key count (with gd_attn_bwd's dq the full backward of vanilla attention).
 *   dk_f32, dv_f32 [BH,M,D] f32, ACCUMULATED into (caller zeroes); per-chunk partials go through `workspace`
 *   (gd_attn_bwd_dkv_workspace_bytes) and are summed in a fixed order — no atomics.
 * Used by the autograd of the vanilla attention op: null-text optimisation (U/inversion.py:213-259) differentiates the UNet w.r.t.
 * the text context, i.e. through k and v of every cross-attention layer and q/k/v of every self-attention layer. */
size_t gd_attn_bwd_dkv_workspace_bytes(int BH, int N, int M, int D);
int gd_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* out, const float* lse, const void* dout,
                    int BH, int N, int M, int D, float scale, float* dk_f32, float* dv_f32,
                    void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* P[bh, r, m] = exp(scale * q[bh, rows[r]] . k[bh, m] - lse[bh, rows[r]])   (rows == NULL: r = row)
 * The opt-pass materialisation of base_att / replace_att rows that removal_loss_geodiff consumes
 * (U/attention_processors.py:250, 307-317).  P is 16-bit [BH, R, Mpad], Mpad = multiple of 8 >= M,
 * padding columns are written as 0.  n_valid (DEVICE int32[1], may be NULL = R): only rows[0 .. n_valid) are wanted — the rest of
 * the list is padding that keeps launch dimensions identical across edits; 128-row tiles made of padding only are skipped (their P
 * rows stay unwritten and are not read by gd_removal_corr_max / gd_removal_bwd given the same n_valid).  D: 64, 128 or 192.
 * ONE or TWO problems per launch (b may be NULL; a hooked layer needs the base map and the inpaint rows of the edit map) + an optional
 * clear of zero_bytes bytes (multiple of 16) at zero_ptr — the `best` scratch of gd_removal_corr_max, which this launch precedes. */
typedef struct gd_probs {
    const void* q; const void* k; const float* lse; const int32_t* rows; const int32_t* n_valid; void* P;
    int32_t BH, N, R, M, Mpad;
} gd_probs_t;
int gd_attn_probs(const gd_probs_t* a, const gd_probs_t* b, int D, float scale, void* zero_ptr, size_t zero_bytes, int dtype,
                  void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) attention forward on the block-scaled K = 64 matrix instruction — opt-in, no-grad passes only (BASELINE configs[4]).
 * The reference has no fp8 path (attention is fp16 compute_attention + torch.bmm, U/attention_sharing.py:30-47; its SDXL line is
 * commented out, U/diffusion.py:106); parity is defined by oracle/ref_cpu.py:attention_fp8_oracle.
 *   gd_fp8_absmax_heads : amax[bh] = max |x[bh]|                       (x 16-bit, [BH,N,64] head-major or [B,N,heads*64] with heads > 0)
 *   gd_fp8_quant_rows   : out8[bh,n,:] = e4m3(clamp(x * 448 / amax[bh], +-448))     [BH,N,64] bytes, head-major.  For the QUERY tensor pass
 *                         amax_k (the keys' absmax) and the softmax scale: c = scale log2(e) (aq/448) (ak/448) is split by frexp into
 *                         mant 2^ex, the multiplier becomes mant * 448 / aq and the attention kernel applies 2^ex as the instruction's
 *                         block scale — the score product then yields exponents of 2 directly.  Keys: amax_k = NULL.
 *   gd_fp8_quant_vt     : V transposed per 64-key tile in the slot order the second product needs: vt8 [BH, ceil(M/64), 64 d, 64 slots]
 *   gd_attn_fwd_fp8     : out = softmax(scale q k^T) v from those, M % 64 == 0, D = 64; out 16-bit head-major (heads = 0) or token-major;
 *                         lse [BH,N] f32 or NULL.  Probabilities are re-quantised to e4m3 (x 64) in registers.
 * ---------------------------------------------------------------------------------------------- */
int gd_fp8_absmax_heads(const void* x, int BH, int heads, int N, float* amax, int dtype, void* stream);
int gd_fp8_quant_rows(const void* x, int BH, int heads, int N, const float* amax, const float* amax_k, float scale, void* out8,
                      int dtype, void* stream);
int gd_fp8_quant_vt(const void* v, int BH, int heads, int M, const float* amax, void* vt8, int dtype, void* stream);
/* gd_fp8_absmax_heads x 3 + gd_fp8_quant_rows (q, with the keys' absmax and the scale) + gd_fp8_quant_rows (k) + gd_fp8_quant_vt (v) in three
 * launches.  amax3: [3, BH] f32 (q | k | v), written. */
int gd_fp8_quantize_qkv(const void* q, const void* k, const void* v, int BH, int heads, int N, int M, float scale, float* amax3,
                        void* q8, void* k8, void* vt8, int dtype, void* stream);
int gd_attn_fwd_fp8(const void* q8, const void* k8, const void* vt8, const float* amax_q, const float* amax_k, const float* amax_v,
                    int BH, int heads, int N, int M, int D, float scale, void* out, float* lse, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * R8  losses.
 * ---------------------------------------------------------------------------------------------- */

/*
 * removal_loss_geodiff forward (U/attention_processors.py:248-268):
 *   corr[h,r,j] = sum_m Pe[h,r,m] * Pb[h,j,m];  (p_in,j_in) = max_j corr*m_inp[j];  (p_wo,j_wo) = max_j corr*m_wo[j]
 * Pe [H,R,Mpad], Pb [H,N,Mpad] 16-bit (gd_attn_probs); m_inp, m_wo [N] f32.
 * best [H,R,2] u64 scratch: (value bits << 32 | ~j) for the inpaint / wo-edit mask (first index wins ties); unpacked by
 * gd_removal_loss_reduce / gd_edit_losses_fwd's tail.  clear != 0: the call zeroes it first; clear == 0: the caller did
 * (gd_attn_probs's zero_ptr, which precedes this launch in a hooked layer).  n_valid_dev (DEVICE int32[1] or NULL): 128-row tiles of
 * Pe at or beyond n_valid are padding and are skipped (their `best` stays 0).
 * variant: 0 = the launcher's choice; tests / benchmarks force a kernel with 1 (the general vector-pipe kernel), 22 or 24 (the MFMA
 * kernel on 2 x 2 / 2 x 4 tiles; falls back to the choice where the shape does not fit).  All variants give identical bits.
 */
int gd_removal_corr_max(const void* Pe, const void* Pb, const float* m_inp, const float* m_wo, const int32_t* n_valid_dev,
                        int H, int R, int N, int Mpad, unsigned long long* best, int clear, int variant, int dtype, void* stream);

/* Unpacks best -> p_in,p_wo [H,R] f32, j_in,j_wo [H,R] i32, writes wgt[h,r] = exp(-dist(rows[r], j_wo)) and
 * loss_acc[0] += sum_{h,r} wgt * (-log(p_wo+1e-4) + log(p_in+1e-4))      (U/attention_processors.py:262-268).
 * dist = CoordinateDistances (U/generic_torch.py:126-140) evaluated analytically on the S x S grid.
 * n_valid_dev (DEVICE int32[1], may be NULL = R): only rows[0 .. n_valid) are inpaint rows, the rest of the list is padding (any
 * valid row index) that gets weight 0 — so that R, and with it every launch dimension of the layer, can be rounded up to a
 * value that repeats from edit to edit (hipGraph reuse). */
int gd_removal_loss_reduce(const unsigned long long* best, const int32_t* rows, const int32_t* n_valid_dev, int H, int R, int S,
                           float* p_in, int32_t* j_in, float* p_wo, int32_t* j_wo, float* wgt,
                           float* loss_acc, void* stream);

/*
 * Backward of the removal loss through replace_att rows into q (and k for cross):
 *   dA[h,r,m] = coef * wgt[h,r] * ( -Pb[h,j_wo,m] * m_wo[j_wo]/(p_wo+1e-4) + Pb[h,j_in,m] * m_inp[j_in]/(p_in+1e-4) )
 *   dS = A o (dA - rowsum(A o dA));  dq[h,rows[r]] += scale * dS K;  dk_f32[h] += scale * dS^T q   (dk_f32 may be NULL)
 * coef is multiplied by the optional DEVICE scalars gscale[0] and gscale2[0] (upstream gradient of the loss and the loss weight: no host
 * sync to read them).  n_valid (DEVICE int32[1] or NULL): slots [n_valid, R) of the row list are padding and are skipped.
 * workspace: gd_removal_bwd_workspace_bytes(H, R, M, Mpad, D, dk_f32 != NULL) bytes — row dots [H*R], per-key-chunk dq partials
 * [msplit, H, R, D] folded in a fixed order (no f32 atomics, bit-reproducible), and dS when dk_f32 != NULL.
 * dq_f32 [H,N,D] f32 accumulated (caller zeroes) and / or dq16_inout, a 16-bit [H,N,D] gradient the contribution is added to in place
 * (element = T(float(element) + contribution)): the COMPLETE backward (row dots, products, fold).
 * dq_f32 == NULL and dq16_inout == NULL: the dS K products only — the row dots were computed by gd_edit_losses_bwd (its `rm` argument)
 * into rm->workspace, and gd_edit_dq_fold folds the partials together with the attention backward's (ONE rounding of dq instead of two).
 */
typedef struct gd_removal_bwd {
    const void* Pe; const void* Pb; const void* q; const void* k; const int32_t* rows;
    const float* p_in; const int32_t* j_in; const float* p_wo; const int32_t* j_wo; const float* wgt;
    const float* m_inp; const float* m_wo; const float* gscale; const float* gscale2;
    const int32_t* n_valid;
    float* dk_f32; float* workspace;
    float coef, scale;
    int32_t H, R, N, M, Mpad, D;
    int32_t variant;   /* 0 = the launcher's choice; 1 forces the general kernel where the MFMA one would run (tests / benchmarks) */
} gd_removal_bwd_t;
size_t gd_removal_bwd_workspace_bytes(int H, int R, int M, int Mpad, int D, int need_dk);
int gd_removal_bwd(const gd_removal_bwd_t* rm, float* dq_f32, void* dq16_inout, int dtype, void* stream);

/*
 * The mask-only half of interpolate_from_mask (U/attention_sharing.py:81-83,103), once per edit and resolution:
 * for every pixel of the S x S grid the 4 foreground pixels (fg > 0.5) with the largest 1/(dist*256 + 1e5*bg + 1e-4),
 * ties resolved (value desc, index asc).  nn_idx [N,4] i32, nn_w [N,4] f32 (those inverse distances),
 * w_dist [N] f32 = exp(-(1/max_i nn_w)/5).
 */
int gd_nn_table(const float* fg, int S, int32_t* nn_idx, float* nn_w, float* w_dist, void* stream);

/*
 * interpolate_from_mask + overwrite + 5x5 gaussian (U/attention_sharing.py:67-105,
 * U/attention_processors.py:291-293, U/generic_torch.py:145-154): the amodal-loss target.
 *   eo [H,N,D] 16-bit; nn_idx [N,4] i32, nn_w [N,4] f32 (precomputed from the mask per edit);
 *   fg [N] f32 (mask_edit > 0.5 rows keep eo); tmp, target [H,N,D] f32.
 */
int gd_amodal_target(const void* eo, const int32_t* nn_idx, const float* nn_w, const float* fg,
                     int H, int S, int D, float* tmp, float* target, int dtype, void* stream);

/*
 * The four feature losses in one pass (U/attention_processors.py:231-246,283-305; U/loss.py:29-41).
 * sums[0] += sum |eo-ro| m_wo      sums[1] += sum |eo-ro| m_edit      sums[2] += sum |tgt-ro| w_am m_amodal
 * sums[3] += sum |ro[y+1]-ro[y]|   sums[4] += sum |ro[x+1]-ro[x]|      (tgt/w_am/m_amodal may be NULL)
 * Bit-reproducible: per-workgroup partials go through `workspace` (gd_edit_losses_fwd_workspace_bytes) and are folded in a
 * fixed order — no floating-point atomics.
 *
 * wv == NULL: the reductions only — out12[0..4] += the five sums (the caller zeroes them; two launches: partials, fold); best / ticket /
 * the running sums unused.
 * wv != NULL: ONE launch for everything between the attention outputs and the backward of a hooked layer — the workgroup that finishes
 * last (an arrival ticket, agent-scope release / acquire) unpacks `best` like gd_removal_loss_reduce (best == NULL: no removal term,
 * rm = 0), folds the partials in the same order and does gd_loss_assemble's arithmetic:
 *   t = sums * inv5; terms = [t0, t1, rm * inv_rm, t3 + t4, use_amodal ? t2 : t1 * 0]; loss = sum terms_i * wv_i;
 *   coefs = wv[{0,1,4,3,3}] * inv5_bwd (what gd_edit_losses_bwd reads); rm_coef = wv[2] * inv_rm
 *   out12 = terms[0:5], loss[5], coefs[6:11], rm_coef[11]; every operand is device f32 (inv5 / wv / inv5_bwd: 5 entries; inv_rm: 1)
 * and the controller's running sums (each may be NULL): log_acc[k] += terms[k] for the four logged terms (sim, movement, removal,
 * smoothness: generic.py:34-39) and loss_out[0] = (loss_in ? loss_in[0] : 0) + loss (U/attention_processors.py:494,604
 * `self.loss = self.loss + loss`) — plain f32 adds in layer order, as the 0-d torch adds were.
 * ticket: ONE int32, zero before the launch, left zero by it.
 */
typedef struct gd_edit_losses {
    const void* eo; const void* ro; const float* tgt; const float* m_wo; const float* m_edit; const float* w_am; const float* m_amodal;
    const unsigned long long* best; const int32_t* rows; const int32_t* n_valid;
    float* p_in; int32_t* j_in; float* p_wo; int32_t* j_wo; float* wgt;
    const float* inv5; const float* inv_rm; const float* wv; const float* inv5_bwd;
    float* out12; float* workspace; int32_t* ticket;
    float* log_acc; const float* loss_in; float* loss_out;
    int32_t H, S, D, R, use_amodal;
} gd_edit_losses_t;
size_t gd_edit_losses_fwd_workspace_bytes(int H, int S, int D);
int gd_edit_losses_fwd(const gd_edit_losses_t* a, int dtype, void* stream);
/* The stand-alone stage of that tail (callers that ran the reductions with wv == NULL and gd_removal_loss_reduce themselves):
 * sums / rm -> out12 as above. */
int gd_loss_assemble(const float* sums, const float* rm, const float* inv5, const float* inv_rm, const float* wv, const float* inv5_bwd,
                     int use_amodal, float* out12, void* stream);

/*
 * d(loss)/d(ro) for the weighted sum of those losses plus the blend path:
 *   g = c[0]*(-sgn(eo-ro)) m_wo + c[1]*(-sgn(eo-ro)) m_edit + c[2]*(-sgn(tgt-ro)) w_am m_amodal
 *     + c[3]*d|D_h| + c[4]*d|D_w| + gout * (blend ? (1-m_edit) : 1)
 * coef_dev: c[5] f32 in DEVICE memory (loss weight / denominator; on the device so that a captured hipGraph of the
 * optimisation pass follows the adaptive weight schedule without re-capture), all multiplied by the optional DEVICE
 * scalar gscale_dev[0] (upstream gradient of the loss); gout [H,N,D] 16-bit (may be NULL);
 * dro [H,N,D] 16-bit.  `blend`: bit 0 = the blend factor above; bit 1 = gout is the token-major row [N, H*D] of the layer's
 * output gradient (batch_to_head_dim's autograd, U/attention_processors.py:213, read in place instead of a permuted copy).
 * rm (may be NULL): the same launch also computes the row dots sum_m A dA of the removal loss's backward into rm->workspace
 * (independent work, one launch less; see gd_removal_bwd).
 */
int gd_edit_losses_bwd(const void* eo, const void* ro, const float* tgt, const float* m_wo, const float* m_edit,
                       const float* w_am, const float* m_amodal, const void* gout, const float* coef_dev, const float* gscale_dev,
                       int blend, int H, int S, int D, void* dro, const gd_removal_bwd_t* rm, int dtype, void* stream);

/* The edit row's output assembly in one pass (U/attention_processors.py:424-428,502-508,544-549,617-622):
 *   e[h,n]   = (act != NULL && pos[n] >= 0) ? act[h, pos[n]] : base[h,n]          (the full edit_out: rows outside the soft edit mask
 *                                                                                    are the reference rows' outputs, gd_attn_seg_t.q_rows)
 *   eo_out   = e                                     (may be NULL)
 *   out      = e*m + ro*(1-m), op by op in the tensor dtype as torch evaluates it on 16-bit tensors   (may be NULL; then ro / m may be NULL)
 * base, ro, eo_out, out [H,N,D]; act [H,R,D]; pos [N] i32; m [N] f32; D % 8 == 0.  act == NULL: the plain blend of :504,619;
 * out == NULL: the plain row merge. */
int gd_blend_merge(const void* base, const void* act, const int32_t* pos, const void* ro, const float* m, int H, int N, int R, int D,
                   void* eo_out, void* out, int dtype, void* stream);

/* The last launch of a hooked layer's backward: dq16[h,n,:] = T( sum_c dq_part[c][h,n,:]  (+ sum_c removal partials of row n, if n is
 * a live inpaint row) ) — ONE rounding (a complete gd_attn_bwd followed by a complete gd_removal_bwd rounds twice).
 * dq_part / kchunks: what gd_attn_bwd left (kchunks_out / dq_part_out); chunk_stride: floats from one run's partials to the next (0 = BH*N*D;
 * larger when BH heads are a slice of a wider backward — one edit's heads of a batch: dq_part and dq16 then point at that slice).  inp_pos [N] i32: slot of row n in the inpaint-row list or -1;
 * rm_workspace: the removal backward's workspace (its partials [msplit, H, R, D] f32 start H*R floats in), NULL = none.
 * dq_part == NULL (kchunks == 1: the dq kernel wrote dq16 directly): only the live inpaint rows are touched,
 * dq16 = T(float(dq16) + removal contribution). */
int gd_edit_dq_fold(const float* dq_part, int kchunks, int64_t chunk_stride, int BH, int N, int D, const float* rm_workspace, int M, int R,
                    const int32_t* inp_pos, const float* wgt, void* dq16, int dtype, void* stream);

/* The layer's layout boundary in ONE launch each way (U/attention_processors.py:118-120,201-203 head_to_batch_dim of q / k / v and
 * :124,213 batch_to_head_dim of the output, and their autograd): the projections hand over token-major [B, rows, heads*D] tensors, the
 * optimisation pass's kernels work on head-major [B*heads, rows, D] ones.
 *   gd_heads_split : n <= GD_HEADS_SPLIT_MAX tensors at once, dst[i][(b*heads + h), r, :] = src[i][b, r, h*D : (h+1)*D]   (rows[i] rows each;
 *                    16-bit).  ABI 6: six tensors (was three) — the edit row's q / k / v and the reference row's q / k / v handed in from the
 *                    previous step's CFG pass (the optimisation pass on the edit row alone) in ONE launch.
 *   gd_heads_merge : out[b, r, h*D:(h+1)*D] = src[b][h, r, :] for every batch row b < B (<= 16: the two roles of up to 8 edits) whose
 *                    source is non-NULL, zeros for a NULL source (the rows that receive no gradient); src_f32 != 0: the sources are f32
 *                    and are rounded once (the key gradient); blend_b[b] != NULL: row b is  src[b]*m[b] + blend_b[b]*(1-m[b])  op by op
 *                    in the tensor dtype, exactly gd_blend_merge's blend (U/attention_processors.py:502-508,617-622) — the blend and
 *                    the layout change in one pass (one mask per row: every edit of a batch blends with its own). */
#define GD_HEADS_SPLIT_MAX 6
typedef struct gd_heads_split {
    const void* src[GD_HEADS_SPLIT_MAX]; void* dst[GD_HEADS_SPLIT_MAX]; int32_t rows[GD_HEADS_SPLIT_MAX];
    int32_t n, B, heads, D;
} gd_heads_split_t;
int gd_heads_split(const gd_heads_split_t* a, int dtype, void* stream);
#define GD_HEADS_MERGE_MAX_ROWS 16
typedef struct gd_heads_merge {
    const void* src[GD_HEADS_MERGE_MAX_ROWS]; const void* blend_b[GD_HEADS_MERGE_MAX_ROWS]; const float* m[GD_HEADS_MERGE_MAX_ROWS]; void* out;
    int32_t src_f32, B, rows, heads, D;
} gd_heads_merge_t;
int gd_heads_merge(const gd_heads_merge_t* a, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * R10-R12  scheduler / latent arithmetic (f32 or 16-bit latents, n elements).
 * ---------------------------------------------------------------------------------------------- */

/* eps = eps_u + g*(eps_c-eps_u) (U/diffusion.py:45-46) then the DDIM closed form (U/inversion.py:47-65):
 *   x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);  out = sqrt(a_to) x0 + sqrt(1-a_to) eps.
 * a_t/a_to are the alphas_cumprod at the source / destination timestep (denoise: t -> t-20; invert: t-20 -> t).
 * eps_c == NULL: eps = eps_u (no CFG). */
int gd_ddim_step(const void* x, const void* eps_u, const void* eps_c, float guidance, float a_t, float a_to,
                 void* out, int64_t n, int dtype, void* stream);

/* The same step for a v-prediction UNet (SD2.1-768, BASELINE configs[3]; the reference has no such path — README.md:61 lists it as
 * to do — so the arithmetic is the published DDIM step of diffusers' DDIMScheduler with prediction_type="v_prediction"):
 *   v = v_u + g*(v_c-v_u);  x0 = sqrt(a_t) x - sqrt(1-a_t) v;  eps = sqrt(a_t) v + sqrt(1-a_t) x;  out = sqrt(a_to) x0 + sqrt(1-a_to) eps. */
int gd_ddim_step_v(const void* x, const void* v_u, const void* v_c, float guidance, float a_t, float a_to,
                   void* out, int64_t n, int dtype, void* stream);

/* U/optimization.py:228-231: x1 <- x1 - step*(1+m)*nan_to_num(g)  (the two chained updates), m [hw] f32
 * broadcast over C channels; x, g, out f32 [C*hw]. */
int gd_masked_latent_update(const float* x, const float* g, const float* m, float step, int C, int hw,
                            float* out, void* stream);

/* U/editor.py:219,316: sumsq[0] += sum x^2 (f32, caller zeroes);  gd_scale: out = x * s[0]/s[1] style rescale is
 * done by the host from two sumsq results: out = x * sqrt(num[0]+1e-12)/sqrt(den[0]+1e-12). */
int gd_sumsq(const float* x, int64_t n, float* sumsq, void* stream);
int gd_norm_rescale(const float* x, const float* num_sumsq, const float* den_sumsq, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * UNet plumbing (not rows of the hot path): element-wise / norm fusions for the no-grad passes of the SD-shaped UNet, which is
 * launch-latency-bound at batch 1-3.  16-bit, channels-last / token-major rows of C channels, C % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */

/* GroupNorm (+ SiLU) of x (+ add_bc): x, y [B, HW, C] (NHWC memory), add_bc [B, C] with row stride add_ld elements (0 = C), or NULL
 * (norm of x + add_bc[b,c]: the ResNet block's time-embedding add), gamma/beta [C];
 * scratch: gd_group_norm_nhwc_scratch_floats(B, HW, G) f32 (no need to clear). */
int64_t gd_group_norm_nhwc_scratch_floats(int B, int HW, int G);
/* Small maps (HW * C/G * 2 B <= 40 KB per (batch entry, group): the 8^2 / 16^2 levels and the narrow 32^2 norms) take ONE launch.
 * single_launch: -1 = that rule (default), 0 = the two-launch form everywhere (benchmarks / tests compare both), 1 = as -1. */
int gd_group_norm_nhwc(const void* x, const void* add_bc, int add_ld, const void* gamma, const void* beta, int B, int HW, int C, int G, float eps,
                       int silu, int single_launch, float* scratch, void* y, int dtype, void* stream);

/* dx of the above for frozen gamma / beta (the optimisation pass differentiates w.r.t. activations only): x, add_bc, gamma, beta as in
 * the forward, dy and dx [B, HW, C]; fwd_scratch = the forward call's scratch (its slab moments give mean / rstd); scratch: same size. */
int gd_group_norm_nhwc_bwd(const void* x, const void* add_bc, int add_ld, const void* gamma, const void* beta, const void* dy,
                           int B, int HW, int C, int G, float eps, int silu, int single_launch, const float* fwd_scratch, float* scratch,
                           void* dx, int dtype, void* stream);

/* y = x + bias[c] (+ res): convolution epilogue; x, res, y [rows, C]; bias [C]; res may be NULL. */
int gd_bias_residual(const void* x, const void* bias, const void* res, int64_t rows, int C, void* y, int dtype, void* stream);

/* y [rows, C] = x[:, :C] * gelu(x[:, C:]) with x [rows, 2C] (GEGLU, exact erf GELU). */
int gd_geglu(const void* x, int64_t rows, int C, void* y, int dtype, void* stream);
/* dx [rows, 2C] of the above for dy [rows, C]: dx[:, :C] = dy * gelu(gate), dx[:, C:] = (dy * h) * gelu'(gate) (16-bit rounding where
 * autograd's unfused chain rounds). */
int gd_geglu_bwd(const void* x, const void* dy, int64_t rows, int C, void* dx, int dtype, void* stream);

/* Gradient of the above w.r.t. s for frozen gamma / beta: ds = dLayerNorm(gy | s) + gs (gs [rows, C] or NULL: the gradient that reaches
 * the residual stream s directly); mean / rstd are recomputed from s.  a and b of the forward both receive ds. */
int gd_layer_norm_bwd(const void* s, const void* gamma, const void* gy, const void* gs, int64_t rows, int C, float eps, void* ds,
                      int dtype, void* stream);
/* s = a + b (16-bit, written to sum_out unless NULL; b may be NULL: s = a);  y = LayerNorm(s) * gamma + beta over the C channels of
 * each row; C <= 2048. */
int gd_add_layer_norm(const void* a, const void* b, const void* gamma, const void* beta, int64_t rows, int C, float eps,
                      void* sum_out, void* y, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * UNet harness: the 3x3 convolutions of the UNet the reference drives (diffusers' UNet2DConditionModel -> torch.nn.functional.conv2d;
 * GeoDiffuser/utils/diffusion.py:87-143 builds it, U/editor.py:248,305 call it) as an implicit GEMM on the matrix cores.
 * in [n, Hi, Wi, C] and out [n, Ho, Wo, K] are NHWC (torch channels_last), w [K, 3, 3, C] (a channels_last conv weight), bias [K] or
 * NULL, residual [n, Ho, Wo, K] or NULL (added in f32 before the one rounding: the ResnetBlock's `x + h`), all 16-bit of `dtype`; padding 1; stride 1 or 2 (Ho = (Hi-1)/stride + 1); upsample = 1: the convolution runs over the
 * nearest-neighbour 2x upsampling of `in` (Ho = 2 Hi) without materialising it (diffusers Upsample2D).  C % 64 == 0, K % 8 == 0
 * (GD_EUNSUPPORTED otherwise: the caller keeps its library convolution for conv_in / conv_out).  Accumulation in f32 over
 * (ky, kx, c) in that order, one rounding at the end; launches that cannot fill the chip split the reduction and fold the f32
 * partials in a fixed order (deterministic) through `workspace` (gd_conv3x3_workspace_bytes; 0 = no split).
 * cfg (NULL = the launcher's heuristics): per-call tuning, replacing the process-wide hooks of ABI <= 4.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gd_conv3x3_cfg {
    int32_t pi, ki, ksplit;   /* force the tile shape (64 PI pixels x 64 KI channels; PI, KI in {1, 2}) and the reduction split (1..64);
                                 pi <= 0: heuristic */
    int32_t dma;              /* -1 / 1: stage the operand tiles with direct-to-LDS loads (buffer_load ... lds) where three LDS stages fit
                                 (default), 0: registers + ds_write (same results) */
} gd_conv3x3_cfg_t;
#define GD_CONV3X3_CFG_DEFAULT {0, 0, 0, -1}
size_t gd_conv3x3_workspace_bytes(int n, int Ho, int Wo, int C, int K, const gd_conv3x3_cfg_t* cfg);
int gd_conv3x3(const void* in, const void* w, const void* bias, const void* residual, void* out, int n, int Hi, int Wi, int C, int K, int stride,
               int upsample, const gd_conv3x3_cfg_t* cfg, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N2  post-process: masked per-channel histogram matching (GeoDiffuser/utils/image_processing.py:24-77).
 * src, tmpl [npix, C] uint8 (interleaved channels, C <= 4); m_src, m_tmpl [npix] uint8 (non-zero = the reference's
 * `mask > 0.5`); counts [2,C,256] u32 scratch (cleared by the call; on return counts[0] = source histogram inside m_src,
 * counts[1] = template inside m_tmpl); lut [C,256] f64 = np.interp(src_quantiles, tmpl_quantiles, 0..255);
 * out [npix, C] f64 = lut[c][src].  Counts are exact; lut/out are bit-identical to numpy's binary64 result.
 * ---------------------------------------------------------------------------------------------- */
int gd_hist_match(const uint8_t* src, const uint8_t* tmpl, const uint8_t* m_src, const uint8_t* m_tmpl, int npix, int C,
                  uint32_t* counts, double* lut, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N4  bilinear forward splatting, summation form (GeoDiffuser/utils/softsplat.py:284-510; dead on the reference's live path).
 * in [N,C,H,W], flow [N,2,H,W] (x, y displacement in pixels), out / outgrad / ingrad [N,C,H,W], flowgrad [N,2,H,W]; all f32 NCHW.
 * out is cleared by the call; source pixels with a non-finite target contribute nothing (gradients 0); ingrad or flowgrad may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int gd_softsplat_fwd(const float* in, const float* flow, int N, int C, int H, int W, float* out, void* stream);
int gd_softsplat_bwd(const float* in, const float* flow, const float* outgrad, int N, int C, int H, int W, float* ingrad,
                     float* flowgrad, void* stream);

/* R10 step glue (ABI 6) — many row copies in ONE launch: for every entry e < n, dst[e][0 .. bytes[e]) = src[e][row * bytes[e] ..): the
 * reference row of optimisation step `row` out of the per-layer tensors one batched UNet pass left for ALL optimisation steps of an edit
 * (editor.REF_AHEAD: ~128 tensors, 140 MB per row) into the persistent one-row tensors the captured passes read by address.
 * `entries` lives in DEVICE memory (n x gd_copy_rows_t); every bytes[e] is a multiple of 16, pointers 16-byte aligned. */
typedef struct gd_copy_rows { const void* src; void* dst; int64_t bytes; } gd_copy_rows_t;
int gd_copy_rows(const gd_copy_rows_t* entries, int n, int row, int64_t max_bytes, void* stream);

/* Host helper: *id = 1 + the runtime's id of the capture sequence `stream` is recording, 0 when it is not capturing (callers that hand
 * out pre-zeroed scratch must not share a chunk between two hipGraph captures). */
int gd_stream_capture_id(void* stream, unsigned long long* id);

#ifdef __cplusplus
}
#endif
#endif /* GEODIFF_HIP_H */
