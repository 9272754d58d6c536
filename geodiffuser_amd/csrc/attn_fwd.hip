// R5/R6/R7 — flash-style attention forward on the matrix cores (gfx950 MFMA), D = 64.
//
// Replaces compute_attention (torch.baddbmm + F.softmax, GeoDiffuser/utils/attention_sharing.py:30-47)
// + torch.bmm(attn, v) (GeoDiffuser/utils/attention_processors.py:428,433,549,557,644,647).  The reference
// materialises sim (fp16) and attn (fp32) [f,N,N] in HBM — 168 + 335 MB per 64^2 layer per map; here the
// map never leaves registers.  Up to 4 independent (q,k,v,out) segments run in one launch (vanilla rows,
// edit_out with the warped queries, replace_out) so that small resolutions still fill 256 CUs.
//
// MFMA-bound: algorithmic FLOPs = 4 * BH * N * M * D per launch (QK^T and PV).
#include <stdlib.h>
#include "attn_common.hpp"

// largest partial row sum (16 probabilities of one lane) tolerated before the reference maximum is raised: every probability
// then stays <= 2^14, inside fp16 range (bf16 / f32 have far more)
#define P_SUM_LIMIT 16384.0f

// (this file is built with -fno-slp-vectorize, see build.py: packed f32 VALU beside MFMAs is an anti-lever.  An inline-asm v_add_f32
// is NOT an alternative: the compiler's hazard recogniser does not look inside asm, and a VALU read of a v_exp_f32 result needs a wait
// state — an asm add read stale registers.)

// one 64-key tile, processed as two 32-key half steps (S^T = K Q^T, online softmax, O^T += V^T P^T).  Half steps keep
// only 16 score + 8 probability registers live, which fits 4 waves per SIMD (<= 128 VGPRs): on this kernel latency
// hiding by occupancy is worth more than the few extra max/rescale checks.
// softmax of one 32-key half step: probabilities of s_acc against the row's reference value -> two 16-bit B fragments, l_run updated
template <typename T, int NO>
__device__ __forceinline__ void softmax_half(const f32x16& s_acc, f32x16 (&o)[NO], float& m_run, float& l_run, float c,
                                             typename elem_traits<T>::vec8& pf0, typename elem_traits<T>::vec8& pf1) {
    using TR = elem_traits<T>;
    // Probabilities against the CURRENT reference maximum, no per-step row maximum: p = exp2(c*s - c*m_run).  A row maximum is
    // only needed to keep p inside the 16-bit range, so it is recomputed (slow path, wave-uniform, executed at most once per
    // half step) only when a lane's partial row sum leaves [0, 2^14] — which also covers the first tile (m_run = -inf gives
    // p = inf) and NaN.  Everything downstream (l, O, lse = m_run*scale + ln l, the split-KV merge) is exact for ANY reference
    // value; it need not be the true maximum.  This removes a 10-deep dependent max chain, an LDS round trip and a branch
    // from every half step (+9-12 % on the 64^2 launches).
    float mc = m_run * c;
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[i], c, -mc));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[i + 1], c, -mc));
        pf0[i] = TR::from_f32(p0); pf0[i + 1] = TR::from_f32(p1);
        ps0 += p0; ps1 += p1;
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[8 + i], c, -mc));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[8 + i + 1], c, -mc));
        pf1[i] = TR::from_f32(p0); pf1[i + 1] = TR::from_f32(p1);
        ps0 += p0; ps1 += p1;
    }
    float ps = ps0 + ps1;
    if (__builtin_amdgcn_ballot_w64(!(ps <= P_SUM_LIMIT)) != 0) {
        float mx = s_acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s_acc[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);       // 0 on the first tile
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < NO; ++j) o[j][i] *= alpha;
        mc = m_run * c;
        ps = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[i], c, -mc));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[8 + i], c, -mc));
            pf0[i] = TR::from_f32(p0); pf1[i] = TR::from_f32(p1);
            ps += p0 + p1;
        }
    }
    l_run += ps;
}

// one 64-key tile as two 32-key half steps (S^T = K Q^T, online softmax, O^T += V^T P^T), software-pipelined inside the tile: both
// score GEMMs are issued up front and the first half's P.V before the second half's softmax, so that within ONE wave the matrix
// pipe has independent work (QK^T of half 1, PV of half 0) while the vector pipe runs the exponentials of the other half.
// NCH = head dim / 64: the K / V tiles are NCH consecutive 64-column images, qf / o carry NCH x 4 fragments / NCH x 2 accumulators.
template <typename T, bool MASKED, int NCH>
__device__ __forceinline__ void fwd_tile(const char* lk, const char* lv, const FragOffs& fo, const typename elem_traits<T>::vec8 (&qf)[4 * NCH],
                                         f32x16 (&o)[2 * NCH], float& m_run, float& l_run, float c, int kv0, int M, int h) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    f32x16 s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int s = 0; s < 4; ++s) s0 = TR::mfma32(rd_row<T>(lk + ch * ATT_TILE_BYTES, fo, 0, s), qf[4 * ch + s], s0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int s = 0; s < 4; ++s) s1 = TR::mfma32(rd_row<T>(lk + ch * ATT_TILE_BYTES, fo, 1, s), qf[4 * ch + s], s1);
    if (MASKED) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (kv0 + acc_key(i, h) >= M) s0[i] = -INFINITY;
            if (kv0 + 32 + acc_key(i, h) >= M) s1[i] = -INFINITY;
        }
    }
    V8 pa0, pa1, pb0, pb1;
    softmax_half<T, 2 * NCH>(s0, o, m_run, l_run, c, pa0, pa1);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
            o[2 * ch + dblk] = TR::mfma32(rd_tr<T>(lv + ch * ATT_TILE_BYTES, fo, dblk, 0), pa0, o[2 * ch + dblk]);
            o[2 * ch + dblk] = TR::mfma32(rd_tr<T>(lv + ch * ATT_TILE_BYTES, fo, dblk, 1), pa1, o[2 * ch + dblk]);
        }
    softmax_half<T, 2 * NCH>(s1, o, m_run, l_run, c, pb0, pb1);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
            o[2 * ch + dblk] = TR::mfma32(rd_tr<T>(lv + ch * ATT_TILE_BYTES, fo, dblk, 2), pb0, o[2 * ch + dblk]);
            o[2 * ch + dblk] = TR::mfma32(rd_tr<T>(lv + ch * ATT_TILE_BYTES, fo, dblk, 3), pb1, o[2 * ch + dblk]);
        }
}

// NCH = head dim / 64 (1: the SD2.1 / SDXL head; 2, 3: the zero-padded 80- and 160-wide SD1.x heads, one workgroup per CU)
template <typename T, int NCH>
__global__ void __launch_bounds__(256, NCH == 1 ? 2 : 1)
k_attn_fwd(const FwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int D = ATT_D * NCH;
    __shared__ __attribute__((aligned(16))) char lds[2][2][NCH * ATT_TILE_BYTES];   // [buf][K|V][64-column chunk]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wg0 = xcd_remap(blockIdx.x, a.nwg);
    const int sp = a.nsplit > 1 ? wg0 % a.nsplit : 0;
    const int wg = a.nsplit > 1 ? wg0 / a.nsplit : wg0;
    const int gbh = wg / a.tiles, tile = wg - gbh * a.tiles;
    int sidx = 0;
#pragma unroll
    for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
        if (i < a.nseg - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
    const int bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
    const gd_attn_seg_t sg = a.seg[sidx];
    const int N = a.N, M = a.M;
    // row stride / base offsets: head-major [bh, N, D] or token-major [B, N, heads*D]
    const int rs = sg.heads > 0 ? sg.heads * D : D;
    size_t qoff, koff;
    if (sg.heads > 0) {
        const int b = bh / sg.heads, hh = bh - b * sg.heads;
        qoff = (size_t)b * N * rs + (size_t)hh * D;
        koff = (size_t)b * M * rs + (size_t)hh * D;
    } else {
        qoff = (size_t)bh * N * D;
        koff = (size_t)bh * M * D;
    }
    const T* __restrict__ qp = (const T*)sg.q + qoff;
    const T* __restrict__ kp = (const T*)sg.k + koff;
    const T* __restrict__ vp = (const T*)sg.v + koff;

    // this lane's query (B operand column); lanes l and l^32 share the query, split d / keys
    const int qrow = tile * ATT_BM + wave * 32 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    V8 qf[4 * NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        V8 f4[4];
        load_q_frags<T>(sg, qp + ch * ATT_D, rs, qld, h, f4);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[4 * ch + s] = f4[s];
    }
    const FragOffs fo = make_frag_offs(lane);

    f32x16 o[2 * NCH];
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[j][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int T_all = (M + ATT_BN - 1) / ATT_BN;
    const int t0 = sp * a.tps;                          // this split's key tiles [t0, T_tiles)
    const int T_tiles = (t0 + a.tps) < T_all ? (t0 + a.tps) : T_all;
    const int T_full = (M / ATT_BN) < T_tiles ? (M / ATT_BN) : T_tiles;     // tiles without a key tail
    u32x4 kr[NCH][2], vr[NCH][2];
    // Two key tiles in all (the 77-key text context of every cross-attention layer: ~2,000 launches per edit, each a chain of
    // load -> tile -> load -> tile): both tiles are fetched in ONE round trip into the two LDS buffers and the loop below runs without
    // loads (12.7 -> ~11 us per launch in the edit's trace)
    const bool two_tiles = NCH == 1 && (T_tiles - t0) == 2;
    if (two_tiles) {
        u32x4 kr2[2], vr2[2];
        tile_load<T>(kp, t0 * ATT_BN, M, tid, kr[0], rs);
        tile_load<T>(vp, t0 * ATT_BN, M, tid, vr[0], rs);
        tile_load<T>(kp, (t0 + 1) * ATT_BN, M, tid, kr2, rs);
        tile_load<T>(vp, (t0 + 1) * ATT_BN, M, tid, vr2, rs);
        tile_store(lds[0][0], tid, kr[0]); tile_store(lds[0][1], tid, vr[0]);
        tile_store(lds[1][0], tid, kr2); tile_store(lds[1][1], tid, vr2);
    } else {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            tile_load<T>(kp + ch * ATT_D, t0 * ATT_BN, M, tid, kr[ch], rs);
            tile_load<T>(vp + ch * ATT_D, t0 * ATT_BN, M, tid, vr[ch], rs);
            tile_store(lds[0][0] + ch * ATT_TILE_BYTES, tid, kr[ch]);
            tile_store(lds[0][1] + ch * ATT_TILE_BYTES, tid, vr[ch]);
        }
    }
    __syncthreads();
    // this thread's chunk of the NEXT tile (rows tid/8 and tid/8 + 32 of tile t0 + 1); advanced by one tile per iteration
    const T* kq = kp + (size_t)((t0 + 1) * ATT_BN + (tid >> 3)) * rs + (tid & 7) * 8;
    const T* vq = vp + (size_t)((t0 + 1) * ATT_BN + (tid >> 3)) * rs + (tid & 7) * 8;

#pragma unroll 1
    for (int t = t0; t < T_full; ++t) {
        const int cur = (t - t0) & 1;
        const bool more = (t + 1) < T_tiles && !two_tiles;
        if (two_tiles) {
        } else if (t + 1 < T_full) {               // next tile is full: no clamping
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                kr[ch][0] = *(const u32x4*)(kq + ch * ATT_D); kr[ch][1] = *(const u32x4*)(kq + ch * ATT_D + (size_t)32 * rs);
                vr[ch][0] = *(const u32x4*)(vq + ch * ATT_D); vr[ch][1] = *(const u32x4*)(vq + ch * ATT_D + (size_t)32 * rs);
            }
            kq += (size_t)ATT_BN * rs; vq += (size_t)ATT_BN * rs;
        } else if (more) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                tile_load<T>(kp + ch * ATT_D, (t + 1) * ATT_BN, M, tid, kr[ch], rs);
                tile_load<T>(vp + ch * ATT_D, (t + 1) * ATT_BN, M, tid, vr[ch], rs);
            }
        }
        fwd_tile<T, false, NCH>(lds[cur][0], lds[cur][1], fo, qf, o, m_run, l_run, a.c, t * ATT_BN, M, h);
        if (more) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                tile_store(lds[cur ^ 1][0] + ch * ATT_TILE_BYTES, tid, kr[ch]);
                tile_store(lds[cur ^ 1][1] + ch * ATT_TILE_BYTES, tid, vr[ch]);
            }
        }
        __syncthreads();
    }
    if (T_full < T_tiles)        // key tail (M % 64 != 0): one masked tile, always the last tile of the last split
        fwd_tile<T, true, NCH>(lds[(T_full - t0) & 1][0], lds[(T_full - t0) & 1][1], fo, qf, o, m_run, l_run, a.c, T_full * ATT_BN, M, h);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (NCH == 1 && a.nsplit > 1) {                      // partial result of this split, merged by k_attn_combine
        if (qrow < N) {
            const size_t r = ((size_t)sp * a.tot_bh + gbh) * N + qrow;
            float* wo = a.ws_o + r * ATT_D;
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = o[dblk][4 * g + j];
                    *(f32x4*)(wo + dblk * 32 + 8 * g + 4 * h) = w;
                }
            if (h == 0) { a.ws_ml[r * 2] = m_run; a.ws_ml[r * 2 + 1] = l_tot; }
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (qrow < N) {
        T* __restrict__ op = (T*)sg.out + qoff + (size_t)qrow * rs;
#pragma unroll
        for (int dblk = 0; dblk < 2 * NCH; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(o[dblk][4 * g + j] * inv);
                *(typename TR::vec4*)(op + dblk * 32 + 8 * g + 4 * h) = w;
            }
        if (sg.lse && h == 0) sg.lse[(size_t)bh * N + qrow] = m_run * a.scale + __logf(l_tot);
    }
}

// ---- short-key launches with the blend inside: gd_attn_fwd_pair ------------------------------------------------------------------
// Round 4.  In a no-grad pass every cross-attention layer (77 text keys) and the 8^2 self-attention layer computed the edit rows twice
// — with the warped reference queries (edit_out) and with the edit latent's own queries (replace_out), two segments of one launch — and
// a second launch blended them: out = edit_out*m + replace_out*(1-m) (U/attention_processors.py:502-508,617-622; for the remover past
// its blend window the identity attention and replace_out under m_inp, :831-834): 1,200 blend launches of ~5 us per edit.  With at
// most two key tiles there is no loop to speak of, so the two sides of a 64-query tile share a workgroup: waves 0-1 run side A, waves
// 2-3 side B (each wave exactly the chain of loads and MFMAs it ran as a wave of the two-segment launch, both sides' K / V tiles staged
// by all four waves), side B's waves hand their rounded 16-bit rows over through LDS, side A's waves blend op by op like k_blend and
// store.  Bit-identical to the two launches, one launch less.  (First version: one 128-query workgroup running side A then side B —
// 20 instead of 25 heads of workgroups, but the two dependent chains in a row made the launch 2 us slower than attention + blend.)
struct PairArgs {
    gd_attn_seg_t seg[GD_ATTN_MAX_SEGS];      // seg[nseg - npair + p] is side A of pair p; its `out` receives the blend
    gd_attn_seg_t b[GD_ATTN_MAX_PAIRS];       // side B of pair p (same bh / heads as side A; out unused)
    const float* m[GD_ATTN_MAX_PAIRS];        // [N] blend mask of pair p
    int bh_end[GD_ATTN_MAX_SEGS];
    int nseg, npair, N, M, tiles, tiles_p, nwg_plain, nwg_pair, nwg;
    float c;
};

// a16 / b16: the two attention outputs as the two-segment launch stored them; every product and the sum rounded to the tensor dtype
// like k_blend
template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec4 pair_blend4(typename elem_traits<T>::vec4 a16, typename elem_traits<T>::vec4 b16, float mraw) {
#pragma clang fp contract(off)
    const float mm = (float)(T)mraw;
    const float om = (float)(T)(1.0f - mm);
    typename elem_traits<T>::vec4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t1 = (float)(T)((float)a16[j] * mm);
        const float t2 = (float)(T)((float)b16[j] * om);
        w[j] = (T)(t1 + t2);
    }
    return w;
}

#define PAIR_XROW 136                          // bytes per exchanged row (128 + 8: the 32 rows of a wave fall into distinct banks)
template <typename T>
__global__ void __launch_bounds__(256, 2)
k_attn_fwd_pair(const PairArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    using V4 = typename TR::vec4;
    constexpr int D = ATT_D;
    __shared__ __attribute__((aligned(16))) char lds[2][2][2][ATT_TILE_BYTES];   // [side][key tile][K|V]
    __shared__ __attribute__((aligned(16))) char xch[64 * PAIR_XROW];           // side B's rounded rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const bool paired = wg >= a.nwg_plain;                                    // workgroup-uniform
    int sidx, bh, tile, pair = 0;
    if (!paired) {
        const int gbh = wg / a.tiles;
        tile = wg - gbh * a.tiles;
        sidx = 0;
#pragma unroll
        for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
            if (i < a.nseg - a.npair - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
        bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
    } else {
        int w2 = wg - a.nwg_plain;
        pair = w2 / a.nwg_pair;                                               // every pair has the same head count
        w2 -= pair * a.nwg_pair;
        bh = w2 / a.tiles_p;
        tile = w2 - bh * a.tiles_p;
        sidx = a.nseg - a.npair + pair;
    }
    const int side = paired ? (wave >> 1) : 0;                                // wave-uniform
    const gd_attn_seg_t sa = a.seg[sidx];
    const gd_attn_seg_t sg = (paired && side) ? a.b[pair] : sa;               // the segment THIS wave attends with
    const int N = a.N, M = a.M;
    const int rs = sa.heads > 0 ? sa.heads * D : D;
    size_t qoff, koff;
    if (sa.heads > 0) {
        const int b = bh / sa.heads, hh = bh - b * sa.heads;
        qoff = (size_t)b * N * rs + (size_t)hh * D;
        koff = (size_t)b * M * rs + (size_t)hh * D;
    } else {
        qoff = (size_t)bh * N * D;
        koff = (size_t)bh * M * D;
    }
    const int qrow = paired ? tile * 64 + (wave & 1) * 32 + (lane & 31) : tile * ATT_BM + wave * 32 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    const FragOffs fo = make_frag_offs(lane);
    const int T_all = (M + ATT_BN - 1) / ATT_BN;                              // 1 or 2 (launcher)
    const int T_full = M / ATT_BN;
    const bool same_k = !paired || a.b[pair].k == sa.k, same_v = !paired || a.b[pair].v == sa.v;

    // every wave's own queries (side A: the gather-composite prologue when the segment carries warp tables) and, by all four waves, the
    // K / V tiles of side A and whichever of side B's differ: one round trip, as in the two-segment launch
    V8 qf[4];
    {
        u32x4 kr[2][2], vr[2][2], kbr[2][2], vbr[2][2];
        const T* __restrict__ kp = (const T*)sa.k + koff;
        const T* __restrict__ vp = (const T*)sa.v + koff;
        const T* __restrict__ kbp = (const T*)a.b[pair].k + koff;
        const T* __restrict__ vbp = (const T*)a.b[pair].v + koff;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < T_all) {
                tile_load<T>(kp, t * ATT_BN, M, tid, kr[t], rs);
                tile_load<T>(vp, t * ATT_BN, M, tid, vr[t], rs);
                if (!same_k) tile_load<T>(kbp, t * ATT_BN, M, tid, kbr[t], rs);
                if (!same_v) tile_load<T>(vbp, t * ATT_BN, M, tid, vbr[t], rs);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < T_all) {
                tile_store(lds[0][t][0], tid, kr[t]); tile_store(lds[0][t][1], tid, vr[t]);
                if (!same_k) tile_store(lds[1][t][0], tid, kbr[t]);
                if (!same_v) tile_store(lds[1][t][1], tid, vbr[t]);
            }
    }
    load_q_frags<T>(sg, (const T*)sg.q + qoff, rs, qld, h, qf);
    const float mraw = (paired && !side) ? a.m[pair][qld] : 0.f;
    __syncthreads();

    const int sk = (side && !same_k) ? 1 : 0, sv = (side && !same_v) ? 1 : 0;
    f32x16 o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[j][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int t = 0; t < T_full; ++t) fwd_tile<T, false, 1>(lds[sk][t][0], lds[sv][t][1], fo, qf, o, m_run, l_run, a.c, t * ATT_BN, M, h);
    if (T_full < T_all) fwd_tile<T, true, 1>(lds[sk][T_full][0], lds[sv][T_full][1], fo, qf, o, m_run, l_run, a.c, T_full * ATT_BN, M, h);
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    // the rounded outputs, exactly as the stand-alone kernel stores them: an f32 product, then ONE conversion.  (Where the 16-bit value
    // feeds 16-bit arithmetic hipcc would select v_fma_mixlo_f16 for fp16 — product and conversion with a single rounding, one ulp away
    // from the stored value in ~5e-5 of the elements; the empty asm keeps the product an f32 register value.)
    V4 w16[2][4];
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pr = o[dblk][4 * g + j] * inv;
                asm volatile("" : "+v"(pr));
                w16[dblk][g][j] = TR::from_f32(pr);
            }
    T* __restrict__ op = (T*)sa.out + qoff + (size_t)qrow * rs;
    if (!paired) {
        if (qrow < N) {
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(V4*)(op + dblk * 32 + 8 * g + 4 * h) = w16[dblk][g];
        }
        return;
    }
    char* xr = xch + ((wave & 1) * 32 + (lane & 31)) * PAIR_XROW;              // the row both sides' lanes of this query share
    if (side) {
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) *(V4*)(xr + (dblk * 32 + 8 * g + 4 * h) * (int)sizeof(T)) = w16[dblk][g];
    }
    __syncthreads();
    if (side || qrow >= N) return;
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const V4 wb = *(const V4*)(xr + (dblk * 32 + 8 * g + 4 * h) * (int)sizeof(T));
            *(V4*)(op + dblk * 32 + 8 * g + 4 * h) = pair_blend4<T>(w16[dblk][g], wb, mraw);
        }
}

extern "C" int gd_attn_fwd_pair(const gd_attn_seg_t* segs, int nseg, const gd_attn_seg_t* side_b, const float* const* blend_m, int npair, int N,
                                int M, int D, float scale, int dtype, void* stream) {
    GD_REQUIRE(segs && side_b && blend_m && nseg >= 1 && nseg <= GD_ATTN_MAX_SEGS && npair >= 1 && npair <= GD_ATTN_MAX_PAIRS && npair <= nseg, GD_EINVAL,
               "gd_attn_fwd_pair: null pointer, nseg=%d (1..%d) or npair=%d (1..%d, <= nseg)", nseg, GD_ATTN_MAX_SEGS, npair, GD_ATTN_MAX_PAIRS);
    GD_REQUIRE(D == ATT_D && M > 0 && M <= 2 * ATT_BN && N > 0, GD_EUNSUPPORTED, "gd_attn_fwd_pair: head dim 64 and at most %d keys (D=%d, M=%d)", 2 * ATT_BN, D, M);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_fwd_pair: dtype must be f16/bf16");
    PairArgs a;
    memset(&a, 0, sizeof(a));
    int tot = 0;
    for (int i = 0; i < nseg; ++i) {
        GD_REQUIRE(segs[i].q && segs[i].k && segs[i].v && segs[i].out && segs[i].bh > 0, GD_EINVAL, "gd_attn_fwd_pair: segment %d has a null pointer or bh<=0", i);
        GD_REQUIRE(!segs[i].lse && !segs[i].q_rows, GD_EUNSUPPORTED, "gd_attn_fwd_pair: no LSE output and no query row list on this path");
        GD_REQUIRE((segs[i].q_scaled != 0) == (segs[0].q_scaled != 0), GD_EINVAL, "gd_attn_fwd_pair: segments disagree on q_scaled");
        a.seg[i] = segs[i];
        tot += segs[i].bh;
        a.bh_end[i] = tot;
    }
    const gd_attn_seg_t& sa0 = segs[nseg - npair];
    for (int p = 0; p < npair; ++p) {
        const gd_attn_seg_t& sa = segs[nseg - npair + p];
        const gd_attn_seg_t& sb = side_b[p];
        GD_REQUIRE(sa.bh == sa0.bh && sa.heads == sa0.heads, GD_EINVAL, "gd_attn_fwd_pair: the pairs of a launch must share one head count and layout");
        GD_REQUIRE(sb.q && sb.k && sb.v && sb.bh == sa.bh && sb.heads == sa.heads && !sb.lse && !sb.q_rows && (sb.q_scaled != 0) == (sa.q_scaled != 0) && blend_m[p],
                   GD_EINVAL, "gd_attn_fwd_pair: pair %d: side B must match side A's head count, layout and query scaling; the mask must not be NULL", p);
        a.b[p] = sb;
        a.m[p] = blend_m[p];
    }
    a.nseg = nseg; a.npair = npair; a.N = N; a.M = M;
    a.tiles = (N + ATT_BM - 1) / ATT_BM;
    a.tiles_p = (N + 63) / 64;                                                // a pair's workgroups hold 64 queries (two waves per side)
    a.nwg_plain = a.tiles * (tot - npair * sa0.bh);
    a.nwg_pair = a.tiles_p * sa0.bh;
    a.nwg = a.nwg_plain + npair * a.nwg_pair;
    a.c = segs[0].q_scaled ? 1.0f : scale * 1.4426950408889634f;
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) k_attn_fwd_pair<f16_t><<<a.nwg, 256, 0, st>>>(a);
    else k_attn_fwd_pair<bf16_t><<<a.nwg, 256, 0, st>>>(a);
    GD_CHECK_LAUNCH("gd_attn_fwd_pair");
    return GD_OK;
}

// merge the split-KV partials: one thread per (row, 4 channels)
template <typename T>
__global__ void k_attn_combine(const FwdArgs a) {
    using TR = elem_traits<T>;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.tot_bh * a.N * (ATT_D / 4);
    if (gid >= total) return;
    const int dq = (int)(gid % (ATT_D / 4));
    const long long rr = gid / (ATT_D / 4);
    const int row = (int)(rr % a.N), gbh = (int)(rr / a.N);
    float mref = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) mref = fmaxf(mref, a.ws_ml[(((size_t)s * a.tot_bh + gbh) * a.N + row) * 2]);
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.nsplit; ++s) {
        const size_t r = ((size_t)s * a.tot_bh + gbh) * a.N + row;
        const float w = __builtin_amdgcn_exp2f((a.ws_ml[r * 2] - mref) * a.c);
        L = __builtin_fmaf(w, a.ws_ml[r * 2 + 1], L);
        const f32x4 o = *(const f32x4*)(a.ws_o + r * ATT_D + dq * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(w, o[j], acc[j]);
    }
    int sidx = 0;
#pragma unroll
    for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
        if (i < a.nseg - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
    const int bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
    const gd_attn_seg_t sg = a.seg[sidx];
    size_t off;
    if (sg.heads > 0) {
        const int b = bh / sg.heads, hh = bh - b * sg.heads;
        off = ((size_t)b * a.N + row) * sg.heads * ATT_D + (size_t)hh * ATT_D;
    } else {
        off = ((size_t)bh * a.N + row) * ATT_D;
    }
    const float inv = 1.0f / L;
    typename TR::vec4 w4;
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = TR::from_f32(acc[j] * inv);
    *(typename TR::vec4*)((T*)sg.out + off + dq * 4) = w4;
    if (sg.lse && dq == 0) sg.lse[(size_t)bh * a.N + row] = mref * a.scale + __logf(L);
}

static void mp_config(long long blocks, int N, int M, int* qb, int* ks, int q_prescaled = 1, int cfg_qb = -1, int cfg_ks = 0);

// Split-KV plan: how many key splits make a launch of tot_bh heads fill the chip (1 = none), and the workspace they need.
extern "C" int gd_attn_fwd_plan(int tot_bh, int N, int M, size_t* workspace_bytes) {
    if (workspace_bytes) *workspace_bytes = 0;
    if (tot_bh <= 0 || N <= 0 || M <= 0) return 1;
    {
        int qb = 0, ks = 0;
        mp_config((long long)((N + 31) / 32) * tot_bh, N, M, &qb, &ks);
        if (qb > 0) return 1;          // the pipelined kernel splits keys inside the workgroup (no workspace)
    }
    const int tiles = (N + ATT_BM - 1) / ATT_BM, t_all = (M + ATT_BN - 1) / ATT_BN;
    const long long nwg = (long long)tiles * tot_bh;
    // measured on MI355X (tools/bench_splitkv.py): 5 heads at 64^2 68 -> 47 us with 4 splits, 10 heads 86 -> 73 us with 2, nothing from
    // 15 heads (480 workgroups) up; 32^2 launches lose more to the merge kernel than they gain
    int ns = (int)(768 / nwg);                        // 3 resident workgroups per CU (140 VGPRs)
    if (ns > 4) ns = 4;
    if (ns > t_all / 16) ns = t_all / 16;             // at least 16 key tiles (1024 keys) per split
    if (ns < 2) return 1;
    if (workspace_bytes) *workspace_bytes = (size_t)ns * tot_bh * N * (ATT_D + 2) * sizeof(float);
    return ns;
}

// Which pipelined configuration serves a launch of tot_bh heads: QB query blocks (32 rows each) x KS key ranges per workgroup; qb = 0:
// use k_attn_fwd (key tails, short key lists).  cfg_qb / cfg_ks: the caller's gd_attn_cfg_t (qb < 0: the heuristics below, 0: never).
static void mp_config(long long blocks, int N, int M, int* qb, int* ks, int q_prescaled, int cfg_qb, int cfg_ks) {
    *qb = 0; *ks = 0;
    const int T = M / ATT_BN;
    if (M % ATT_BN != 0 || cfg_qb == 0) return;
    if (cfg_qb > 0) {
        if (cfg_qb == 8) { if (T % 4 == 0) { *qb = 8; *ks = 1; } return; }
        if (T % (2 * cfg_ks) == 0) { *qb = cfg_qb; *ks = cfg_ks; }
        return;
    }
    if (T % 2 != 0) return;
    // wave-sized query blocks against the chip's 1024 SIMDs x 2 resident waves.  Measured on MI355X (tools/bench_mp.py, bf16, 64^2):
    // 5 heads (640 blocks) 43 -> 34 us with two key ranges per workgroup, 10 heads (1280 blocks) 59 us unsplit vs 69 split, 32^2 with
    // 30 heads (960 blocks) 14.5 -> 13.3 us split; from 1280 blocks up the unsplit 128-query workgroup wins.
    // From 160 units of 256 queries up (10 heads at 64^2) the 64-query-per-wave kernel wins (k_attn_fwd_w64; fp16 runs its rescue
    // variant): tools/bench_sk.py, 64^2: 15 heads 66.5 -> 59.1 us, 20 heads 102.3 -> 86.2 us (even split),
    // 32 heads 135.3 -> 122.0 us; 96^2: 20 heads 427 -> 377 us; 32^2 x 10 heads (40 units) 12.1 -> 17.5 us: stays below.
    // Both variants: pre-scaled queries (no-grad passes; reference = first tile's maximum, retry with exact maxima on overflow) and exact
    // scale (optimisation pass; in-loop rescue like k_attn_fwd_mp's, so a dominant probability is exactly 1.0 after a rescue — the
    // property the optimisation pass's parity tolerances were calibrated on).
    if (T % 4 == 0 && blocks >= 1280 && M >= 1024) { *qb = 8; *ks = 1; return; }
    // 64-128 units against 48+ key tiles (the 5-head inversion launch at 64^2: 80 units): the same kernel with every unit cut into
    // 2-4 runs of key tiles, one per workgroup (needs the even split's workspace: without it the launcher falls back to 4 x 2 below):
    // 30.3 us against 34.5 us for two key ranges inside 160 workgroups (tools/bench_sk.py)
    if (T % 4 == 0 && T >= 48 && blocks >= 512 && blocks <= 1024) { *qb = 8; *ks = 1; return; }
    if (blocks < 1280 && T % 4 == 0) { *qb = 4; *ks = 2; return; }
    *qb = 4; *ks = 1;
}

extern "C" size_t gd_attn_fwd_workspace_bytes(int tot_bh, int N, int M) { return gd_attn_sk_workspace_bytes(tot_bh, N, M); }

// The one forward entry point (ABI 5): split-KV with cfg->nsplit > 1, otherwise the even split where a workspace is given, otherwise plain.
// No process-wide state: everything that selects a kernel variant arrives in `cfg`.
extern "C" int gd_attn_fwd(const gd_attn_seg_t* segs, int nseg, int N, int M, int D, float scale, const gd_attn_cfg_t* cfg_in, void* workspace,
                           size_t workspace_bytes, int dtype, void* stream) {
    gd_attn_cfg_t cfg = GD_ATTN_CFG_DEFAULT;
    if (cfg_in) cfg = *cfg_in;
    if (cfg.qb > 0) {
        const int c = cfg.qb * 10 + cfg.ks;
        GD_REQUIRE(c == 41 || c == 22 || c == 42 || c == 24 || c == 81, GD_EINVAL, "gd_attn_fwd: cfg: no kernel for QB=%d KS=%d", cfg.qb, cfg.ks);
    }
#if defined(GD_MP_DBG) && GD_MP_DBG
    const int max_handoff = 2;       // timing builds only: 2 = no merge (WRONG outputs)
#else
    const int max_handoff = 1;
#endif
    GD_REQUIRE(cfg.even_split >= -1 && cfg.even_split <= 2 && cfg.handoff >= 0 && cfg.handoff <= max_handoff, GD_EINVAL,
               "gd_attn_fwd: cfg: even_split=%d (-1..2), handoff=%d (0..%d)", cfg.even_split, cfg.handoff, max_handoff);
    int nsplit = cfg.nsplit > 1 ? cfg.nsplit : 1;
    GD_REQUIRE(segs && nseg >= 1 && nseg <= GD_ATTN_MAX_SEGS, GD_EINVAL, "gd_attn_fwd: nseg=%d (1..%d)", nseg, GD_ATTN_MAX_SEGS);
    GD_REQUIRE(D == 64 || D == 128 || D == 192, GD_EUNSUPPORTED,
               "gd_attn_fwd: head dim %d unsupported (64, 128, 192; zero-pad 40 / 80 / 160 and pass the true scale)", D);
    GD_REQUIRE(D == ATT_D || nsplit == 1, GD_EUNSUPPORTED, "gd_attn_fwd: split-KV: head dim %d unsupported (only 64)", D);
    GD_REQUIRE(N > 0 && M > 0, GD_EINVAL, "gd_attn_fwd: bad sizes N=%d M=%d", N, M);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_fwd: dtype must be f16/bf16");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.cseg = -1;
    int tot = 0, ncs = 0;
    for (int i = 0; i < nseg; ++i) {
        GD_REQUIRE(segs[i].q && segs[i].k && segs[i].v && segs[i].out && segs[i].bh > 0, GD_EINVAL,
                   "gd_attn_fwd: segment %d has a null pointer or bh<=0", i);
        GD_REQUIRE((segs[i].q_scaled != 0) == (segs[0].q_scaled != 0), GD_EINVAL, "gd_attn_fwd: segments disagree on q_scaled");
        a.seg[i] = segs[i];
        tot += segs[i].bh;
        a.bh_end[i] = tot;
        if (segs[i].q_rows) {
            GD_REQUIRE(ncs < GD_ATTN_MAX_ROWLIST_SEGS, GD_EUNSUPPORTED, "gd_attn_fwd: at most %d segments of a launch may carry a query row list",
                       GD_ATTN_MAX_ROWLIST_SEGS);
            GD_REQUIRE(segs[i].q_rows_n && segs[i].q_rows_len > 0, GD_EINVAL, "gd_attn_fwd: segment %d: q_rows without q_rows_n / q_rows_len", i);
            if (a.cseg < 0) a.cseg = i;
            ++ncs;
        }
    }
    // without split-KV a workspace is the even split's (arrival counters, zero before the first launch, + part slots)
    bool sk_ws_ok = false;
    if (nsplit == 1) {
        if (workspace && cfg.even_split != 0) {
            const size_t need = gd_attn_sk_workspace_bytes(tot, N, M);
            GD_REQUIRE(workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0, GD_EINVAL,
                       "gd_attn_fwd: workspace %zu B < %zu B (gd_attn_fwd_workspace_bytes) or not 256-byte aligned", workspace_bytes, need);
            sk_ws_ok = true;
        }
        workspace = sk_ws_ok ? workspace : nullptr;
    }
    a.nseg = nseg; a.N = N; a.M = M;
    a.tiles = (N + ATT_BM - 1) / ATT_BM;
    if (sk_ws_ok) {
        a.sk_ws = (f32x4*)workspace;
        a.sk_cnt = (int*)((char*)workspace + GD_SK_SLOT_BYTES);
        a.sk_mode = cfg.handoff;
        a.sk_force = cfg.even_split == 2;
    }
    a.scale = scale;
    a.c = scale * 1.4426950408889634f;
    a.q_prescaled = segs[0].q_scaled != 0;
    a.lsum = segs[0].q_scaled == 2;
    if (a.q_prescaled) {             // scores arrive as exponents of 2: nothing left to multiply; lse = m ln 2 + ln l
        a.c = 1.0f;
        a.scale = 0.6931471805599453f;
    }
    const int t_all = (M + ATT_BN - 1) / ATT_BN;
    GD_REQUIRE(nsplit >= 1 && nsplit <= 8 && nsplit <= t_all, GD_EINVAL, "gd_attn_fwd: nsplit=%d (1..min(8, key tiles=%d))", nsplit, t_all);
    a.nsplit = nsplit; a.tot_bh = tot;
    a.tps = (t_all + nsplit - 1) / nsplit;
    GD_REQUIRE((long long)a.tps * (nsplit - 1) < t_all, GD_EINVAL, "gd_attn_fwd: nsplit=%d leaves an empty split for %d key tiles", nsplit, t_all);
    if (nsplit > 1) {
        const size_t need = (size_t)nsplit * tot * N * (ATT_D + 2) * sizeof(float);
        GD_REQUIRE(workspace && workspace_bytes >= need, GD_EINVAL, "gd_attn_fwd: split-KV workspace %zu B < %zu B (gd_attn_fwd_plan)", workspace_bytes, need);
        a.ws_o = (float*)workspace;
        a.ws_ml = a.ws_o + (size_t)nsplit * tot * N * ATT_D;
    }
    hipStream_t st = as_stream(stream);
    if (nsplit == 1 && D == ATT_D) {
        int qb = 0, ks = 0;
        // 32-query blocks of the launch (a row-list segment counts its list, not N)
        long long blocks = (long long)((N + 31) / 32) * tot;
        for (int i = 0; i < nseg; ++i)
            if (segs[i].q_rows) blocks -= (long long)((N + 31) / 32 - (segs[i].q_rows_len + 31) / 32) * segs[i].bh;
        mp_config(blocks, N, M, &qb, &ks, a.q_prescaled, cfg.qb, cfg.ks);
        if (qb > 0) return gd_attn_fwd_mp_launch(a, qb, ks, dtype, st);       // software-pipelined kernels (attn_fwd_mp.hip)
    }
    GD_REQUIRE(a.cseg < 0, GD_EUNSUPPORTED, "gd_attn_fwd: a query row list needs head dim 64 and full key tiles (M=%d, D=%d)", M, D);
    a.nwg = a.tiles * tot * nsplit;
    if (D == 128) {
        if (dtype == GD_F16) k_attn_fwd<f16_t, 2><<<a.nwg, 256, 0, st>>>(a);
        else k_attn_fwd<bf16_t, 2><<<a.nwg, 256, 0, st>>>(a);
    } else if (D == 192) {
        if (dtype == GD_F16) k_attn_fwd<f16_t, 3><<<a.nwg, 256, 0, st>>>(a);
        else k_attn_fwd<bf16_t, 3><<<a.nwg, 256, 0, st>>>(a);
    } else if (dtype == GD_F16) k_attn_fwd<f16_t, 1><<<a.nwg, 256, 0, st>>>(a);
    else k_attn_fwd<bf16_t, 1><<<a.nwg, 256, 0, st>>>(a);
    if (nsplit > 1) {
        const long long total = (long long)tot * N * (ATT_D / 4);
        const int blocks = (int)((total + 255) / 256);
        if (dtype == GD_F16) k_attn_combine<f16_t><<<blocks, 256, 0, st>>>(a);
        else k_attn_combine<bf16_t><<<blocks, 256, 0, st>>>(a);
    }
    GD_CHECK_LAUNCH("gd_attn_fwd");
    return GD_OK;
}

// ---------------------------------------------------------------------------------------------------
// gd_attn_probs: P[bh, r, m] = exp(scale * q[rows[r]] . k[m] - lse[rows[r]])   (16-bit, row stride Mpad)
// Same swapped QK^T tile; the 32 x 64 P tile of each wave is transposed through LDS so that every store
// is a full 128-B row segment.
// ---------------------------------------------------------------------------------------------------
struct ProbsArgs {
    const void* q; const void* k; const float* lse; const int32_t* rows; void* P;
    const int32_t* n_valid;       // device scalar: rows [n_valid, R) are padding of the row list and are not computed (NULL: all R)
    int N, R, M, Mpad, tiles, nwg;
    int kchunks, tpc;             // the key tiles are cut into kchunks ranges of tpc tiles, one workgroup each (P tiles are independent)
    float c;      // scale * log2(e)
    float l2e;
};

template <typename T, int NCH>
__device__ __forceinline__ void probs_body(const ProbsArgs& a, const int bid, char (&ldsk)[2][NCH * ATT_TILE_BYTES], T (&stage)[4][32][ATT_BN + 8]) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int D = ATT_D * NCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wg0 = xcd_remap(bid, a.nwg);
    const int kc = wg0 % a.kchunks, wg = wg0 / a.kchunks;
    const int bh = wg / a.tiles, tile = wg - bh * a.tiles;
    const int N = a.N, M = a.M, R = a.R;
    if (a.n_valid && tile * ATT_BM >= a.n_valid[0]) return;       // a tile of padding slots only
    const T* __restrict__ qp = (const T*)a.q + (size_t)bh * N * D;
    const T* __restrict__ kp = (const T*)a.k + (size_t)bh * M * D;
    T* __restrict__ Pp = (T*)a.P + (size_t)bh * R * a.Mpad;

    const int r_out = tile * ATT_BM + wave * 32 + (lane & 31);
    const int r_c = r_out < R ? r_out : R - 1;
    const int qrow = a.rows ? a.rows[r_c] : r_c;
    V8 qf[4 * NCH];
#pragma unroll
    for (int s = 0; s < 4 * NCH; ++s) qf[s] = *(const V8*)(qp + (size_t)qrow * D + 16 * s + 8 * h);
    const float lse2 = a.lse[(size_t)bh * N + qrow] * a.l2e;

    const int T_all = (a.Mpad + ATT_BN - 1) / ATT_BN;
    const int t_lo = kc * a.tpc;
    const int T_tiles = (t_lo + a.tpc) < T_all ? (t_lo + a.tpc) : T_all;
    if (t_lo >= T_tiles) return;
    u32x4 kr[NCH][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        tile_load<T>(kp + ch * ATT_D, t_lo * ATT_BN, M, tid, kr[ch], D);
        tile_store(ldsk[0] + ch * ATT_TILE_BYTES, tid, kr[ch]);
    }
    __syncthreads();
    for (int t = t_lo; t < T_tiles; ++t) {
        const int cur = (t - t_lo) & 1;
        const bool more = (t + 1) < T_tiles;
        if (more) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) tile_load<T>(kp + ch * ATT_D, (t + 1) * ATT_BN, M, tid, kr[ch], D);
        }
        const int kv0 = t * ATT_BN;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 s_acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) s_acc[i] = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    s_acc = TR::mfma32(read_row_frag<T>(ldsk[cur] + ch * ATT_TILE_BYTES, blk, s, lane), qf[4 * ch + s], s_acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = kv0 + blk * 32 + 8 * g + 4 * h + j;
                    const float p = key < M ? __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[4 * g + j], a.c, -lse2)) : 0.f;
                    w[j] = TR::from_f32(p);
                }
                *(typename TR::vec4*)(&stage[wave][lane & 31][blk * 32 + 8 * g + 4 * h]) = w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // each wave writes its own 32 rows x 64 keys: 8 lanes x 16 B per row, 8 rows per pass
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rr = pass * 8 + (lane >> 3), cc = (lane & 7) * 8;
            const int rg = tile * ATT_BM + wave * 32 + rr;
            if (rg < R && kv0 + cc < a.Mpad)
                *(u32x4*)(Pp + (size_t)rg * a.Mpad + kv0 + cc) = *(const u32x4*)(&stage[wave][rr][cc]);
        }
        if (more) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) tile_store(ldsk[cur ^ 1] + ch * ATT_TILE_BYTES, tid, kr[ch]);
        }
        __syncthreads();
    }
}

// one or two problems in one grid (a's workgroups first; b.nwg == 0: none) + an optional clear of zero_n4 16-byte words (the first workgroups' threads)
template <typename T, int NCH>
__global__ void __launch_bounds__(256, 2)
k_attn_probs2(const ProbsArgs a, const ProbsArgs b, u32x4* __restrict__ zero_ptr, int zero_n4) {
    __shared__ __attribute__((aligned(16))) char ldsk[2][NCH * ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) T stage[4][32][ATT_BN + 8];
    if (zero_ptr) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < zero_n4; i += (int)gridDim.x * 256) zero_ptr[i] = z;
    }
    if ((int)blockIdx.x < a.nwg) probs_body<T, NCH>(a, (int)blockIdx.x, ldsk, stage);
    else probs_body<T, NCH>(b, (int)blockIdx.x - a.nwg, ldsk, stage);
}

static ProbsArgs probs_args(const void* q, const void* k, const float* lse, const int32_t* rows, const int32_t* n_valid_dev, int BH, int N, int R,
                            int M, int Mpad, float scale, void* P) {
    ProbsArgs a;
    a.q = q; a.k = k; a.lse = lse; a.rows = rows; a.P = P; a.n_valid = n_valid_dev;
    a.N = N; a.R = R; a.M = M; a.Mpad = Mpad;
    a.tiles = (R + ATT_BM - 1) / ATT_BM;
    // Fill the chip: 5 heads x 32 row tiles are 160 workgroups of one wave per SIMD, each alternating MFMA -> LDS transpose -> global store
    // (75 us for the 168 MB base map of a 64^2 layer, 2.2 TB/s).  The tiles of a row block are independent, so the key range is cut until
    // there are >= 1024 workgroups (at least 8 key tiles = 512 keys each).
    {
        const int t_all = (Mpad + ATT_BN - 1) / ATT_BN;
        int kcn = 1;
        while ((long long)a.tiles * BH * kcn < 1024 && t_all / (kcn * 2) >= 8) kcn *= 2;
        a.kchunks = kcn;
        a.tpc = (t_all + kcn - 1) / kcn;
    }
    a.nwg = a.tiles * BH * a.kchunks;
    a.c = scale * 1.4426950408889634f;
    a.l2e = 1.4426950408889634f;
    return a;
}

extern "C" int gd_attn_probs(const gd_probs_t* pa, const gd_probs_t* pb, int D, float scale, void* zero_ptr, size_t zero_bytes, int dtype,
                             void* stream) {
    GD_REQUIRE(pa, GD_EINVAL, "gd_attn_probs: null pointer");
    GD_REQUIRE(D == 64 || D == 128 || D == 192, GD_EUNSUPPORTED, "gd_attn_probs: head dim %d unsupported (64, 128, 192)", D);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_probs: dtype must be f16/bf16");
    GD_REQUIRE(!zero_ptr || (zero_bytes % 16 == 0 && ((uintptr_t)zero_ptr & 15) == 0 && zero_bytes < ((size_t)1 << 31)), GD_EINVAL,
               "gd_attn_probs: the clear must be 16-byte aligned and a multiple of 16 bytes");
    const gd_probs_t* ps[2] = {pa, pb};
    ProbsArgs args[2];
    memset(args, 0, sizeof(args));
    for (int i = 0; i < (pb ? 2 : 1); ++i) {
        const gd_probs_t* p = ps[i];
        GD_REQUIRE(p->q && p->k && p->lse && p->P, GD_EINVAL, "gd_attn_probs: problem %d: null pointer", i);
        GD_REQUIRE(p->BH > 0 && p->N > 0 && p->R > 0 && p->M > 0 && p->Mpad >= p->M && (p->Mpad & 7) == 0, GD_EINVAL,
                   "gd_attn_probs: problem %d: bad sizes (Mpad must be a multiple of 8 and >= M)", i);
        args[i] = probs_args(p->q, p->k, p->lse, p->rows, p->n_valid, p->BH, p->N, p->R, p->M, p->Mpad, scale, p->P);
    }
    hipStream_t st = as_stream(stream);
    const int grid = args[0].nwg + args[1].nwg;
    const int nch = D / ATT_D;
    u32x4* zp = (u32x4*)zero_ptr;
    const int zn = zero_ptr ? (int)(zero_bytes / 16) : 0;
#define GD_PROBS2(NCH)                                                                            \
    if (dtype == GD_F16) k_attn_probs2<f16_t, NCH><<<grid, 256, 0, st>>>(args[0], args[1], zp, zn); \
    else k_attn_probs2<bf16_t, NCH><<<grid, 256, 0, st>>>(args[0], args[1], zp, zn)
    if (nch == 1) { GD_PROBS2(1); } else if (nch == 2) { GD_PROBS2(2); } else { GD_PROBS2(3); }
#undef GD_PROBS2
    GD_CHECK_LAUNCH("gd_attn_probs");
    return GD_OK;
}
