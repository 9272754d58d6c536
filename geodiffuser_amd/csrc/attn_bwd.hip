// Attention backward for the edit path (gfx950 MFMA), D = 64 NCH (NCH = 1: SD2.1; 2, 3: zero-padded 80 / 160-wide SD1.x heads).
//
// Reference: torch.autograd through compute_attention + torch.bmm (GeoDiffuser/utils/attention_sharing.py:30-47,
// GeoDiffuser/utils/attention_processors.py:432-433,555-557) as driven by torch.autograd.grad in
// GeoDiffuser/utils/optimization.py:182.  On this path only q_edit (always) and k_edit (cross-attention)
// carry gradient; k_base / v_base are detached (U/attention_sharing.py:242, U/attention_processors.py:433,555-557).
//
//   k_attn_bwd_dq : same swapped geometry as the forward.  S^T = K Q^T and dP^T = V dO^T have the query on the
//                   lane, so P, dS = P o (dP - delta) stay lane-local and dS (16-bit) is directly the B operand
//                   of dQ^T += K^T dS^T (K^T by hardware-transposed LDS read).  No atomics, no [N,N] traffic.
//   k_attn_bwd_dk : cross-attention only (M <= 128 keys): key on the MFMA lane, query tiles streamed through LDS, per-chunk
//                   partial dK summed by a second kernel (see below).
// MFMA-bound (dq): algorithmic FLOPs = 6 * BH * N * M * D.
#include <stdlib.h>
#include "attn_common.hpp"

struct BwdArgs {
    const void* q; const void* k; const void* v; const void* o; const float* lse; const void* dout;
    void* dq; float* dk; float* dk_part;
    int N, M, tiles, nwg;
    float c, scale, l2e;
    // dq with few workgroups: the key tiles are cut into kchunks ranges of tpc tiles (one workgroup each); the f32 partials
    // dq_part [kchunks, BH, N, D] are summed in chunk order by k_attn_bwd_dq_fold (no atomics)
    int kchunks, tpc;
    float* dq_part;
};

template <typename T, int NCH>
__global__ void __launch_bounds__(256, NCH == 1 ? 2 : 1)
k_attn_bwd_dq(const BwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int D = ATT_D * NCH;
    __shared__ __attribute__((aligned(16))) char lds[2][2][NCH * ATT_TILE_BYTES];   // [buf][K|V][64-column chunk]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wg0 = xcd_remap(blockIdx.x, a.nwg);
    const int kc = wg0 % a.kchunks, wg = wg0 / a.kchunks;
    const int bh = wg / a.tiles, tile = wg - bh * a.tiles;
    const int N = a.N, M = a.M;
    const T* __restrict__ qp = (const T*)a.q + (size_t)bh * N * D;
    const T* __restrict__ kp = (const T*)a.k + (size_t)bh * M * D;
    const T* __restrict__ vp = (const T*)a.v + (size_t)bh * M * D;
    const T* __restrict__ op = (const T*)a.o + (size_t)bh * N * D;
    const T* __restrict__ gp = (const T*)a.dout + (size_t)bh * N * D;

    const int qrow = tile * ATT_BM + wave * 32 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    V8 qf[4 * NCH], gf[4 * NCH];
    float dpart = 0.f;
#pragma unroll
    for (int s = 0; s < 4 * NCH; ++s) {
        qf[s] = *(const V8*)(qp + (size_t)qld * D + 16 * s + 8 * h);
        gf[s] = *(const V8*)(gp + (size_t)qld * D + 16 * s + 8 * h);
        const V8 of = *(const V8*)(op + (size_t)qld * D + 16 * s + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart = __builtin_fmaf(TR::to_f32(gf[s][j]), TR::to_f32(of[j]), dpart);
    }
    const float delta = dpart + __shfl_xor(dpart, 32, 64);         // rowsum(dO o O)
    const float lse2 = a.lse[(size_t)bh * N + qld] * a.l2e;

    f32x16 dq[2 * NCH];
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[j][i] = 0.f;

    const int T_all = (M + ATT_BN - 1) / ATT_BN;
    const int t_lo = kc * a.tpc;
    const int T_tiles = (t_lo + a.tpc) < T_all ? (t_lo + a.tpc) : T_all;        // launcher: every chunk holds at least one tile
    u32x4 kr[NCH][2], vr[NCH][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        tile_load<T>(kp + ch * ATT_D, t_lo * ATT_BN, M, tid, kr[ch], D);
        tile_load<T>(vp + ch * ATT_D, t_lo * ATT_BN, M, tid, vr[ch], D);
        tile_store(lds[0][0] + ch * ATT_TILE_BYTES, tid, kr[ch]);
        tile_store(lds[0][1] + ch * ATT_TILE_BYTES, tid, vr[ch]);
    }
    __syncthreads();

    for (int t = t_lo; t < T_tiles; ++t) {
        const int cur = (t - t_lo) & 1;
        const bool more = (t + 1) < T_tiles;
        if (more) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                tile_load<T>(kp + ch * ATT_D, (t + 1) * ATT_BN, M, tid, kr[ch], D);
                tile_load<T>(vp + ch * ATT_D, (t + 1) * ATT_BN, M, tid, vr[ch], D);
            }
        }
        const char* lk = lds[cur][0];
        const char* lv = lds[cur][1];
        const int kv0 = t * ATT_BN;
        V8 dsf[4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 s_acc, p_acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s_acc[i] = 0.f; p_acc[i] = 0.f; }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) s_acc = TR::mfma32(read_row_frag<T>(lk + ch * ATT_TILE_BYTES, blk, s, lane), qf[4 * ch + s], s_acc);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) p_acc = TR::mfma32(read_row_frag<T>(lv + ch * ATT_TILE_BYTES, blk, s, lane), gf[4 * ch + s], p_acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[i], a.c, -lse2));
                if (kv0 + blk * 32 + acc_key(i, h) >= M) p = 0.f;
                s_acc[i] = p * (p_acc[i] - delta);
            }
            dsf[2 * blk] = acc_to_frag<T>(s_acc, 0);
            dsf[2 * blk + 1] = acc_to_frag<T>(s_acc, 1);
        }
        // dQ^T += K^T dS^T
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    dq[2 * ch + dblk] = TR::mfma32(read_tr_frag<T>(lk + ch * ATT_TILE_BYTES, dblk, ks, lane), dsf[ks], dq[2 * ch + dblk]);

        if (more) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                tile_store(lds[cur ^ 1][0] + ch * ATT_TILE_BYTES, tid, kr[ch]);
                tile_store(lds[cur ^ 1][1] + ch * ATT_TILE_BYTES, tid, vr[ch]);
            }
        }
        __syncthreads();
    }
    if (a.kchunks > 1) {                 // f32 partial of this key range
        if (qrow < N) {
            const int n_bh = a.nwg / (a.kchunks * a.tiles);
            float* __restrict__ pp = a.dq_part + (((size_t)kc * n_bh + bh) * N + qrow) * D;
#pragma unroll
            for (int dblk = 0; dblk < 2 * NCH; ++dblk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = dq[dblk][4 * g + j] * a.scale;
                    *(f32x4*)(pp + dblk * 32 + 8 * g + 4 * h) = w;
                }
        }
        return;
    }
    if (qrow < N) {
        T* __restrict__ dp = (T*)a.dq + ((size_t)bh * N + qrow) * D;
#pragma unroll
        for (int dblk = 0; dblk < 2 * NCH; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(dq[dblk][4 * g + j] * a.scale);
                *(typename TR::vec4*)(dp + dblk * 32 + 8 * g + 4 * h) = w;
            }
    }
}


// ---- k_attn_bwd_dq2: the same arithmetic on direct-to-LDS staging --------------------------------------------------------------
// Round 4 (what k_corr_max2 / the 64-query forward learnt): K and V tiles arrive by buffer_load ... lds into THREE stages (16 KB each:
// one tile pair), tile t + 2 issued right behind the barrier that retires tile t - 1, counted s_waitcnt vmcnt + raw s_barrier (a
// __syncthreads() fence would drain the loads in flight), no staging registers and no ds_write; two workgroups of four 32-query waves per
// CU (two waves per SIMD: one wave's exponentials under the other's MFMAs).  The key tiles are cut into `kchunks` runs chosen so that the
// launch is ONE round of <= 512 resident workgroups (5 heads at 64^2: 160 query tiles x 3 runs of 21-22 key tiles; k_attn_bwd_dq's
// power-of-two split made 640 workgroups = 1.25 rounds).  Full key tiles only (M % 64 == 0), head dim 64.
// Same MFMA operand order per tile as k_attn_bwd_dq; the partials of a row are summed in run order by the fold: results differ from the
// unsplit kernel only by the f32 grouping of the key range.
#define DQ2_PPW 4                                  // direct-to-LDS pieces per wave and tile pair: 2 of K, 2 of V
// PRE: c == 1 exactly (the optimisation pass's queries carry scale * log2 e, attention_processors._project_qkv): the score accumulator
// starts at -lse * log2 e, so p = exp2(acc) with no vector instruction in front.  Both variants start the dP accumulator at -delta
// (dS = p * acc): two of the five vector instructions per probability move into the MFMAs' C operand, the loop becomes MFMA-bound
// (32 probabilities per wave and tile: 14.5 + 8 issue cycles each against 24 MFMAs x 32).
template <typename T, bool PRE>
__global__ void __launch_bounds__(256, 2)
k_attn_bwd_dq2(const BwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) char st0[2 * ATT_TILE_BYTES];      // [K | V]
    __shared__ __attribute__((aligned(16))) char st1[2 * ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) char st2[2 * ATT_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg0 = xcd_remap(blockIdx.x, a.nwg);
    const int kc = wg0 % a.kchunks, wg = wg0 / a.kchunks;
    const int bh = wg / a.tiles, tile = wg - bh * a.tiles;
    const int N = a.N, M = a.M;
    const T* __restrict__ qp = (const T*)a.q + (size_t)bh * N * ATT_D;
    const T* __restrict__ op = (const T*)a.o + (size_t)bh * N * ATT_D;
    const T* __restrict__ gp = (const T*)a.dout + (size_t)bh * N * ATT_D;
    const int T_all = M / ATT_BN;
    const int t_lo = kc * a.tpc;
    const int t_hi = (t_lo + a.tpc) < T_all ? (t_lo + a.tpc) : T_all;
    const int Tg = t_hi - t_lo;                                             // launcher: every run holds at least one tile
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.k + ((size_t)bh * M + (size_t)t_lo * ATT_BN) * ATT_D), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.v + ((size_t)bh * M + (size_t)t_lo * ATT_BN) * ATT_D), 0, 0x7FFFFFFF, 0x00020000);
    uint32_t voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wave + j) * 8 + (lane >> 3);
        const int x = (row >> 1) & 7, g = (x & 2) | ((x & 1) << 2) | ((x >> 2) & 1);
        voff[j] = (uint32_t)((row * ATT_D + ((lane & 7) ^ g) * 8) * (int)sizeof(T));
    }
#define DQ2_ISSUE(ST, TI)                                                                        \
    {                                                                                            \
        const int so_ = (TI) * ATT_TILE_BYTES;                                                   \
        dma_asm(rsK, (ST) + wave * 2048, voff[0], so_);                                          \
        dma_asm(rsK, (ST) + wave * 2048 + 1024, voff[1], so_);                                   \
        dma_asm(rsV, (ST) + ATT_TILE_BYTES + wave * 2048, voff[0], so_);                         \
        dma_asm(rsV, (ST) + ATT_TILE_BYTES + wave * 2048 + 1024, voff[1], so_);                  \
    }
    DQ2_ISSUE(st0, 0)
    DQ2_ISSUE(st1, Tg > 1 ? 1 : 0)

    const int qrow = tile * ATT_BM + wave * 32 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    V8 qf[4], gf[4];
    float dpart = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *(const V8*)(qp + (size_t)qld * ATT_D + 16 * s + 8 * h);
        gf[s] = *(const V8*)(gp + (size_t)qld * ATT_D + 16 * s + 8 * h);
        const V8 of = *(const V8*)(op + (size_t)qld * ATT_D + 16 * s + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart = __builtin_fmaf(TR::to_f32(gf[s][j]), TR::to_f32(of[j]), dpart);
    }
    const float delta = dpart + __shfl_xor(dpart, 32, 64);         // rowsum(dO o O)
    const float lse2 = a.lse[(size_t)bh * N + qld] * a.l2e;
    // The compiler counts only ITS loads: a wait it emits for q / dO / O / lse inside the loop (`vmcnt(0)`: "my last load") would drain
    // the asm-issued pieces of later tiles on every trip.  Retire them here, once: an empty asm that reads every loaded register.
    float delta_ = delta, lse2_ = lse2;
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(gf[0]), "+v"(gf[1]), "+v"(gf[2]), "+v"(gf[3]), "+v"(delta_), "+v"(lse2_));
    f32x16 c_s, c_p;                     // the chains' C operands: -lse2 (PRE; else 0) and -delta in every register
#pragma unroll
    for (int i = 0; i < 16; ++i) { c_s[i] = PRE ? -lse2_ : 0.f; c_p[i] = -delta_; }
    const FragOffs fo = make_frag_offs(lane);
    f32x16 dq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[j][i] = 0.f;

    // One tile.  The fragment reads of a group are issued one group AHEAD of its MFMAs and the order is pinned (sched_barrier): left to
    // itself the scheduler sinks every ds_read to just in front of the MFMA that consumes it (`s_waitcnt lgkmcnt(1)` before each of the
    // 24 MFMAs of a tile: the LDS latency 24 times per tile; PMC of that build: MFMA busy 30 %, waves parked 39 % of their cycles).
    //   read K / V rows of keys 0..31 | read rows of keys 32..63, MFMAs of keys 0..31 | read K^T (k-steps 0, 1), MFMAs of keys 32..63 beside
    //   the exponentials of keys 0..31 | read K^T (k-steps 2, 3), exponentials of keys 32..63 beside dQ (k-steps 0, 1) | dQ (k-steps 2, 3)
#define DQ2_SOFTMAX(SA, PA, D0, D1)                                                                          \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                     \
            const float p = __builtin_amdgcn_exp2f(PRE ? SA[i] : __builtin_fmaf(SA[i], a.c, -lse2_));        \
            SA[i] = p * PA[i];                                                                               \
        }                                                                                                    \
        D0 = acc_to_frag<T>(SA, 0);                                                                          \
        D1 = acc_to_frag<T>(SA, 1);                                                                          \
    }
#define DQ2_STEP(CUR, NXT, TT)                                                                               \
    if ((TT) < Tg) {                                                                                         \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DQ2_PPW) : "memory");                            \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        {                                                                                                    \
            const int tn_ = (TT) + 2 < Tg ? (TT) + 2 : Tg - 1;      /* past the end: a harmless repeat into a dead stage */ \
            DQ2_ISSUE(NXT, tn_)                                                                              \
        }                                                                                                    \
        const char* lk = (CUR);                                                                              \
        const char* lv = (CUR) + ATT_TILE_BYTES;                                                             \
        V8 ka[4], va[4], kb[4], vb[4], kt0[4], kt1[4], dsf[4];                                               \
        f32x16 s0 = c_s, p0 = c_p, s1 = c_s, p1 = c_p;                                                       \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) { ka[s] = rd_row<T>(lk, fo, 0, s); va[s] = rd_row<T>(lv, fo, 0, s); } \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) { kb[s] = rd_row<T>(lk, fo, 1, s); vb[s] = rd_row<T>(lv, fo, 1, s); } \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) { s0 = TR::mfma32(ka[s], qf[s], s0); p0 = TR::mfma32(va[s], gf[s], p0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        kt0[0] = rd_tr<T>(lk, fo, 0, 0); kt0[1] = rd_tr<T>(lk, fo, 0, 1); kt1[0] = rd_tr<T>(lk, fo, 1, 0); kt1[1] = rd_tr<T>(lk, fo, 1, 1); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) { s1 = TR::mfma32(kb[s], qf[s], s1); p1 = TR::mfma32(vb[s], gf[s], p1); } \
        DQ2_SOFTMAX(s0, p0, dsf[0], dsf[1])                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        kt0[2] = rd_tr<T>(lk, fo, 0, 2); kt0[3] = rd_tr<T>(lk, fo, 0, 3); kt1[2] = rd_tr<T>(lk, fo, 1, 2); kt1[3] = rd_tr<T>(lk, fo, 1, 3); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        dq[0] = TR::mfma32(kt0[0], dsf[0], dq[0]); dq[1] = TR::mfma32(kt1[0], dsf[0], dq[1]);                \
        dq[0] = TR::mfma32(kt0[1], dsf[1], dq[0]); dq[1] = TR::mfma32(kt1[1], dsf[1], dq[1]);                \
        DQ2_SOFTMAX(s1, p1, dsf[2], dsf[3])                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        dq[0] = TR::mfma32(kt0[2], dsf[2], dq[0]); dq[1] = TR::mfma32(kt1[2], dsf[2], dq[1]);                \
        dq[0] = TR::mfma32(kt0[3], dsf[3], dq[0]); dq[1] = TR::mfma32(kt1[3], dsf[3], dq[1]);                \
    }
#pragma unroll 1
    for (int t = 0; t < Tg; t += 3) {
        DQ2_STEP(st0, st2, t)
        DQ2_STEP(st1, st0, t + 1)
        DQ2_STEP(st2, st1, t + 2)
    }
#undef DQ2_STEP
#undef DQ2_SOFTMAX
#undef DQ2_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the repeats of the last tile pair: nothing may land after the wave ends
    if (qrow >= N) return;
    if (a.kchunks > 1) {                 // f32 partial of this key run
        const int n_bh = a.nwg / (a.kchunks * a.tiles);
        float* __restrict__ pp = a.dq_part + (((size_t)kc * n_bh + bh) * N + qrow) * ATT_D;
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = dq[dblk][4 * g + j] * a.scale;
                *(f32x4*)(pp + dblk * 32 + 8 * g + 4 * h) = w;
            }
        return;
    }
    T* __restrict__ dp = (T*)a.dq + ((size_t)bh * N + qrow) * ATT_D;
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typename TR::vec4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(dq[dblk][4 * g + j] * a.scale);
            *(typename TR::vec4*)(dp + dblk * 32 + 8 * g + 4 * h) = w;
        }
}

// dq2's run count: the launch should be ONE round of at most 512 resident workgroups (two per CU), every run at least 8 key tiles
static int dq2_kchunks(int BH, int N, int M) {
    const long long wgs = (long long)((N + ATT_BM - 1) / ATT_BM) * BH;
    const int t_all = M / ATT_BN;
    int kc = (int)(512 / (wgs > 0 ? wgs : 1));
    if (kc > t_all / 8) kc = t_all / 8;
    if (kc > 8) kc = 8;
    return kc < 1 ? 1 : kc;
}
// k_attn_bwd_dq2 where it applies (D = 64, full key tiles, >= 4 tiles); k_attn_bwd_dq (register staging) serves every other shape
static bool dq2_applies(int M, int D, int variant) { return variant != 1 && D == ATT_D && M % ATT_BN == 0 && M >= 4 * ATT_BN; }

// dq[i] = 16-bit( sum_c dq_part[c][i] ), c ascending (4 elements per thread)
template <typename T>
__global__ void k_attn_bwd_dq_fold(const float* __restrict__ part, int kchunks, long long n4, T* __restrict__ dq) {
    using TR = elem_traits<T>;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = *(const f32x4*)(part + i * 4);
    for (int c = 1; c < kchunks; ++c) {
        const f32x4 p = *(const f32x4*)(part + ((long long)c * n4 + i) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += p[j];
    }
    typename TR::vec4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(acc[j]);
    *(typename TR::vec4*)(dq + i * 4) = w;
}

// How many key ranges the dq kernel is cut into: launches under two workgroups per CU (5 heads x 32 query tiles at 64^2 = 160
// workgroups of one wave per SIMD: 96 us, 13 % MFMA busy) are split until there are >= 512, keeping >= 8 key tiles per range.
static int dq_kchunks(int BH, int N, int M) {
    const long long wgs = (long long)((N + ATT_BM - 1) / ATT_BM) * BH;
    const int t_all = (M + ATT_BN - 1) / ATT_BN;
    int kc = 1;
    while (wgs * kc < 512 && t_all / (kc * 2) >= 8 && kc < 8) kc *= 2;
    return kc;
}

// ---- dK for cross-attention (M <= 128 keys) on the matrix cores ------------------------------------------
// The forward's geometry with the roles swapped: the KEY sits on the MFMA lane (4 waves x 32 keys = every key of a 77-key text
// context in one workgroup, K / V fragments in registers for the whole kernel) and 64-query tiles of Q and dO stream through LDS.
//   S[q, key]  = Q_tile K^T,   dP[q, key] = dO_tile V^T           (A = row fragments of the LDS tile, B = K / V registers)
//   dS = P o (dP - delta[q]),  P = exp2(c S - lse[q] log2 e)      (per-ROW constants: broadcast reads from a small LDS table)
//   dK^T[d, key] += Q_tile^T dS                                   (A = hardware-transposed read of the SAME Q tile, B = dS as it
//                                                                  leaves the accumulators — the forward's P.V trick)
// One workgroup per (head, DK_QCHUNK queries); the per-chunk partials are summed by k_attn_bwd_dk_reduce (no atomics).
#define DK_QCHUNK 128
template <typename T, int NCH>
__global__ void __launch_bounds__(256, NCH == 1 ? 2 : 1)
k_attn_bwd_dk(const BwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int D = ATT_D * NCH;
    __shared__ __attribute__((aligned(16))) char lds[2][NCH * ATT_TILE_BYTES];      // [Q | dO] tile images
    __shared__ float s_lse2[ATT_BN], s_delta[ATT_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int bh = blockIdx.y, chunk = blockIdx.x;
    const int N = a.N, M = a.M;
    const T* __restrict__ qp = (const T*)a.q + (size_t)bh * N * D;
    const T* __restrict__ kp = (const T*)a.k + (size_t)bh * M * D;
    const T* __restrict__ vp = (const T*)a.v + (size_t)bh * M * D;
    const T* __restrict__ op = (const T*)a.o + (size_t)bh * N * D;
    const T* __restrict__ gp = (const T*)a.dout + (size_t)bh * N * D;

    const int key = wave * 32 + (lane & 31);
    const int kld = key < M ? key : M - 1;
    V8 kf[4 * NCH], vf[4 * NCH];
#pragma unroll
    for (int s = 0; s < 4 * NCH; ++s) {
        kf[s] = *(const V8*)(kp + (size_t)kld * D + 16 * s + 8 * h);
        vf[s] = *(const V8*)(vp + (size_t)kld * D + 16 * s + 8 * h);
    }
    f32x16 dk[2 * NCH];
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) dk[j][i] = 0.f;

    const int q_begin = chunk * DK_QCHUNK;
    const int q_end = (q_begin + DK_QCHUNK) < N ? (q_begin + DK_QCHUNK) : N;
    for (int q0 = q_begin; q0 < q_end; q0 += ATT_BN) {
        u32x4 qr[NCH][2], gr[NCH][2], orr[NCH][2];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            tile_load<T>(qp + ch * ATT_D, q0, N, tid, qr[ch], D);
            tile_load<T>(gp + ch * ATT_D, q0, N, tid, gr[ch], D);
            tile_load<T>(op + ch * ATT_D, q0, N, tid, orr[ch], D);
        }
        __syncthreads();                                  // the previous tile's reads are done
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            tile_store(lds[0] + ch * ATT_TILE_BYTES, tid, qr[ch]);
            tile_store(lds[1] + ch * ATT_TILE_BYTES, tid, gr[ch]);
        }
        // delta[row] = sum_d dO o O: this thread holds chunk (tid & 7) of rows (tid >> 3) and (tid >> 3) + 32
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float d = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const V8 g8 = __builtin_bit_cast(V8, gr[ch][i]), o8 = __builtin_bit_cast(V8, orr[ch][i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) d = __builtin_fmaf(TR::to_f32(g8[j]), TR::to_f32(o8[j]), d);
            }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if ((tid & 7) == 0) {
                const int row = (tid >> 3) + 32 * i, qi = q0 + row;
                s_delta[row] = d;
                s_lse2[row] = qi < N ? a.lse[(size_t)bh * N + qi] * a.l2e : INFINITY;     // exp2(-inf) = 0 for padding queries
            }
        }
        __syncthreads();
        V8 dsf[4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 s_acc, p_acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s_acc[i] = 0.f; p_acc[i] = 0.f; }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) s_acc = TR::mfma32(read_row_frag<T>(lds[0] + ch * ATT_TILE_BYTES, blk, s, lane), kf[4 * ch + s], s_acc);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) p_acc = TR::mfma32(read_row_frag<T>(lds[1] + ch * ATT_TILE_BYTES, blk, s, lane), vf[4 * ch + s], p_acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = blk * 32 + acc_key(i, h);                 // query row of accumulator register i
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[i], a.c, -s_lse2[row]));
                if (key >= M) p = 0.f;
                s_acc[i] = p * (p_acc[i] - s_delta[row]);
            }
            dsf[2 * blk] = acc_to_frag<T>(s_acc, 0);
            dsf[2 * blk + 1] = acc_to_frag<T>(s_acc, 1);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    dk[2 * ch + dblk] = TR::mfma32(read_tr_frag<T>(lds[0] + ch * ATT_TILE_BYTES, dblk, ks, lane), dsf[ks], dk[2 * ch + dblk]);
    }
    if (key < M) {            // partial dK of this query chunk (plain stores; summed by k_attn_bwd_dk_reduce)
        float* dst = a.dk_part + (((size_t)bh * gridDim.x + chunk) * M + key) * D;
#pragma unroll
        for (int dblk = 0; dblk < 2 * NCH; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = dk[dblk][4 * g + j] * a.scale;
                *(f32x4*)(dst + dblk * 32 + 8 * g + 4 * h) = w;
            }
    }
}

__global__ void k_attn_bwd_dk_reduce(const float* __restrict__ part, int nchunks, int MD, float* __restrict__ dk, int store) {
    const int bh = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= MD) return;
    const float* p = part + (size_t)bh * nchunks * MD + o;
    float acc = 0.f;
    for (int c = 0; c < nchunks; ++c) acc += p[(size_t)c * MD];
    if (store) dk[(size_t)bh * MD + o] = 0.0f + acc;     // what the accumulating form leaves in a zeroed buffer, without the caller's fill launch
    else dk[(size_t)bh * MD + o] += acc;
}

// ---- dK and dV for any key count (the full backward of vanilla attention) -------------------------------------------------
// k_attn_bwd_dk's geometry with a key-block grid dimension and a second accumulator pair:
//   dV^T[d, key] += dO_tile^T P        (A = hardware-transposed read of the dO tile, B = P as it leaves the accumulators)
//   dK^T[d, key] += Q_tile^T dS
// One workgroup per (query chunk, block of 128 keys, head); a chunk is `qchunk` queries — ALL queries when the key count is large
// (self-attention: no partial sums at all), 128 for short key lists (cross-attention: more workgroups, per-chunk partials).  The
// partials [BH, chunks, M, 64] f32 x 2 are summed in chunk order by k_attn_bwd_dk_reduce (no atomics).  Used by the autograd of the
// vanilla attention op (null-text optimisation differentiates the UNet w.r.t. its text context: U/inversion.py:213-259); the edit
// path itself never needs dV or self-attention dK (those tensors are detached in the reference, U/attention_sharing.py:242).
struct DkvArgs {
    const void* q; const void* k; const void* v; const void* o; const float* lse; const void* dout;
    float* dk_part; float* dv_part;
    int N, M, qchunk, nchunks;
    float c, scale, l2e;
};

template <typename T, int NCH>
__global__ void __launch_bounds__(256, 1)
k_attn_bwd_dkv(const DkvArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int D = ATT_D * NCH;
    __shared__ __attribute__((aligned(16))) char lds[2][NCH * ATT_TILE_BYTES];      // [Q | dO] tile images
    __shared__ float s_lse2[ATT_BN], s_delta[ATT_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int chunk = blockIdx.x, kblk = blockIdx.y, bh = blockIdx.z;
    const int N = a.N, M = a.M;
    const T* __restrict__ qp = (const T*)a.q + (size_t)bh * N * D;
    const T* __restrict__ kp = (const T*)a.k + (size_t)bh * M * D;
    const T* __restrict__ vp = (const T*)a.v + (size_t)bh * M * D;
    const T* __restrict__ op = (const T*)a.o + (size_t)bh * N * D;
    const T* __restrict__ gp = (const T*)a.dout + (size_t)bh * N * D;

    const int key = kblk * 128 + wave * 32 + (lane & 31);
    const int kld = key < M ? key : M - 1;
    V8 kf[4 * NCH], vf[4 * NCH];
#pragma unroll
    for (int s = 0; s < 4 * NCH; ++s) {
        kf[s] = *(const V8*)(kp + (size_t)kld * D + 16 * s + 8 * h);
        vf[s] = *(const V8*)(vp + (size_t)kld * D + 16 * s + 8 * h);
    }
    f32x16 dk[2 * NCH], dv[2 * NCH];
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[j][i] = 0.f; dv[j][i] = 0.f; }

    const int q_begin = chunk * a.qchunk;
    const int q_end = (q_begin + a.qchunk) < N ? (q_begin + a.qchunk) : N;
    for (int q0 = q_begin; q0 < q_end; q0 += ATT_BN) {
        u32x4 qr[NCH][2], gr[NCH][2], orr[NCH][2];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            tile_load<T>(qp + ch * ATT_D, q0, N, tid, qr[ch], D);
            tile_load<T>(gp + ch * ATT_D, q0, N, tid, gr[ch], D);
            tile_load<T>(op + ch * ATT_D, q0, N, tid, orr[ch], D);
        }
        __syncthreads();                                  // the previous tile's reads are done
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            tile_store(lds[0] + ch * ATT_TILE_BYTES, tid, qr[ch]);
            tile_store(lds[1] + ch * ATT_TILE_BYTES, tid, gr[ch]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // delta[row] = sum_d dO o O (see k_attn_bwd_dk)
            float d = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const V8 g8 = __builtin_bit_cast(V8, gr[ch][i]), o8 = __builtin_bit_cast(V8, orr[ch][i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) d = __builtin_fmaf(TR::to_f32(g8[j]), TR::to_f32(o8[j]), d);
            }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if ((tid & 7) == 0) {
                const int row = (tid >> 3) + 32 * i, qi = q0 + row;
                s_delta[row] = d;
                s_lse2[row] = qi < N ? a.lse[(size_t)bh * N + qi] * a.l2e : INFINITY;     // exp2(-inf) = 0 for padding queries
            }
        }
        __syncthreads();
        V8 dsf[4], pf[4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 s_acc, p_acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s_acc[i] = 0.f; p_acc[i] = 0.f; }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) s_acc = TR::mfma32(read_row_frag<T>(lds[0] + ch * ATT_TILE_BYTES, blk, s, lane), kf[4 * ch + s], s_acc);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) p_acc = TR::mfma32(read_row_frag<T>(lds[1] + ch * ATT_TILE_BYTES, blk, s, lane), vf[4 * ch + s], p_acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = blk * 32 + acc_key(i, h);                 // query row of accumulator register i
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[i], a.c, -s_lse2[row]));
                if (key >= M) p = 0.f;
                s_acc[i] = p;
                p_acc[i] = p * (p_acc[i] - s_delta[row]);
            }
            pf[2 * blk] = acc_to_frag<T>(s_acc, 0);
            pf[2 * blk + 1] = acc_to_frag<T>(s_acc, 1);
            dsf[2 * blk] = acc_to_frag<T>(p_acc, 0);
            dsf[2 * blk + 1] = acc_to_frag<T>(p_acc, 1);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    dk[2 * ch + dblk] = TR::mfma32(read_tr_frag<T>(lds[0] + ch * ATT_TILE_BYTES, dblk, ks, lane), dsf[ks], dk[2 * ch + dblk]);
                    dv[2 * ch + dblk] = TR::mfma32(read_tr_frag<T>(lds[1] + ch * ATT_TILE_BYTES, dblk, ks, lane), pf[ks], dv[2 * ch + dblk]);
                }
    }
    if (key < M) {            // partials of this query chunk (plain stores; summed by k_attn_bwd_dk_reduce)
        const size_t off = (((size_t)bh * a.nchunks + chunk) * M + key) * D;
#pragma unroll
        for (int dblk = 0; dblk < 2 * NCH; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { wk[j] = dk[dblk][4 * g + j] * a.scale; wv[j] = dv[dblk][4 * g + j]; }
                *(f32x4*)(a.dk_part + off + dblk * 32 + 8 * g + 4 * h) = wk;
                *(f32x4*)(a.dv_part + off + dblk * 32 + 8 * g + 4 * h) = wv;
            }
    }
}

// launch K<f16_t / bf16_t, D / 64>
#define GD_LAUNCH_NCH(K, GRID, ARGS)                                                                   \
    do {                                                                                               \
        const int nch_ = D / ATT_D;                                                                    \
        if (dtype == GD_F16) {                                                                         \
            if (nch_ == 1) K<f16_t, 1><<<GRID, 256, 0, st>>>(ARGS);                                    \
            else if (nch_ == 2) K<f16_t, 2><<<GRID, 256, 0, st>>>(ARGS);                               \
            else K<f16_t, 3><<<GRID, 256, 0, st>>>(ARGS);                                              \
        } else {                                                                                       \
            if (nch_ == 1) K<bf16_t, 1><<<GRID, 256, 0, st>>>(ARGS);                                   \
            else if (nch_ == 2) K<bf16_t, 2><<<GRID, 256, 0, st>>>(ARGS);                              \
            else K<bf16_t, 3><<<GRID, 256, 0, st>>>(ARGS);                                             \
        }                                                                                              \
    } while (0)
#define GD_HEAD_DIM_OK(D) ((D) == 64 || (D) == 128 || (D) == 192)

static int dkv_qchunk(int N, int M) { return M >= 1024 ? N : DK_QCHUNK; }

extern "C" size_t gd_attn_bwd_dkv_workspace_bytes(int BH, int N, int M, int D) {
    const int qc = dkv_qchunk(N, M);
    const size_t chunks = (size_t)(N + qc - 1) / qc;
    return 2 * (size_t)BH * chunks * M * D * sizeof(float);
}

extern "C" int gd_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* out, const float* lse, const void* dout,
                               int BH, int N, int M, int D, float scale, float* dk_f32, float* dv_f32,
                               void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    GD_REQUIRE(q && k && v && out && lse && dout && dk_f32 && dv_f32, GD_EINVAL, "gd_attn_bwd_dkv: null pointer");
    GD_REQUIRE(GD_HEAD_DIM_OK(D), GD_EUNSUPPORTED, "gd_attn_bwd_dkv: head dim %d unsupported (64, 128, 192)", D);
    GD_REQUIRE(BH > 0 && N > 0 && M > 0, GD_EINVAL, "gd_attn_bwd_dkv: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_bwd_dkv: dtype must be f16/bf16");
    GD_REQUIRE(workspace && workspace_bytes >= gd_attn_bwd_dkv_workspace_bytes(BH, N, M, D), GD_EWORKSPACE,
               "gd_attn_bwd_dkv: needs a workspace of gd_attn_bwd_dkv_workspace_bytes() bytes");
    DkvArgs a;
    a.q = q; a.k = k; a.v = v; a.o = out; a.lse = lse; a.dout = dout;
    a.N = N; a.M = M;
    a.qchunk = dkv_qchunk(N, M);
    a.nchunks = (N + a.qchunk - 1) / a.qchunk;
    a.dk_part = (float*)workspace;
    a.dv_part = a.dk_part + (size_t)BH * a.nchunks * M * D;
    a.scale = scale;
    a.c = scale * 1.4426950408889634f;
    a.l2e = 1.4426950408889634f;
    hipStream_t st = as_stream(stream);
    dim3 grid(a.nchunks, (M + 127) / 128, BH);
    GD_LAUNCH_NCH(k_attn_bwd_dkv, grid, a);
    dim3 rgrid((M * D + 255) / 256, BH);
    k_attn_bwd_dk_reduce<<<rgrid, 256, 0, st>>>(a.dk_part, a.nchunks, M * D, dk_f32, 0);
    k_attn_bwd_dk_reduce<<<rgrid, 256, 0, st>>>(a.dv_part, a.nchunks, M * D, dv_f32, 0);
    GD_CHECK_LAUNCH("gd_attn_bwd_dkv");
    return GD_OK;
}

// workspace: [dK partials (need_dk)] [dq partials (when the dq kernel is split over the keys)]
static size_t bwd_dk_ws_bytes(int BH, int N, int M, int D, int need_dk) {
    if (!need_dk) return 0;
    const size_t chunks = (size_t)(N + DK_QCHUNK - 1) / DK_QCHUNK;
    return (size_t)BH * chunks * M * D * sizeof(float);
}

extern "C" size_t gd_attn_bwd_workspace_bytes(int BH, int N, int M, int D, int need_dk, int variant) {
    const int kc = dq2_applies(M, D, variant) ? dq2_kchunks(BH, N, M) : dq_kchunks(BH, N, M);
    return bwd_dk_ws_bytes(BH, N, M, D, need_dk) + (kc > 1 ? (size_t)kc * BH * N * D * sizeof(float) : 0);
}

static int attn_bwd_launch(const void* q, const void* k, const void* v, const void* out, const float* lse,
                           const void* dout, int BH, int N, int M, int D, float scale,
                           void* dq, float* dk_f32, void* workspace, size_t workspace_bytes, int dtype, void* stream, bool fold,
                           int* kchunks_out, float** dq_part_out, int variant) {
    GD_REQUIRE(q && k && v && out && lse && dout && dq, GD_EINVAL, "gd_attn_bwd: null pointer");
    GD_REQUIRE(GD_HEAD_DIM_OK(D), GD_EUNSUPPORTED, "gd_attn_bwd: head dim %d unsupported (64, 128, 192)", D);
    GD_REQUIRE(BH > 0 && N > 0 && M > 0, GD_EINVAL, "gd_attn_bwd: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_bwd: dtype must be f16/bf16");
    GD_REQUIRE(!dk_f32 || M <= 128, GD_EUNSUPPORTED, "gd_attn_bwd: dK is only implemented for M <= 128 keys (cross-attention); M=%d", M);
    BwdArgs a;
    const size_t need_ws = gd_attn_bwd_workspace_bytes(BH, N, M, D, dk_f32 != nullptr, variant);
    GD_REQUIRE(need_ws == 0 || (workspace && workspace_bytes >= need_ws), GD_EWORKSPACE,
               "gd_attn_bwd: needs a workspace of gd_attn_bwd_workspace_bytes() = %zu bytes", need_ws);
    a.q = q; a.k = k; a.v = v; a.o = out; a.lse = lse; a.dout = dout; a.dq = dq; a.dk = dk_f32; a.dk_part = (float*)workspace;
    a.N = N; a.M = M;
    a.tiles = (N + ATT_BM - 1) / ATT_BM;
    const bool dq2 = dq2_applies(M, D, variant);
    a.kchunks = dq2 ? dq2_kchunks(BH, N, M) : dq_kchunks(BH, N, M);
    a.tpc = ((M + ATT_BN - 1) / ATT_BN + a.kchunks - 1) / a.kchunks;
    while (a.kchunks > 1 && (long long)a.tpc * (a.kchunks - 1) >= (M + ATT_BN - 1) / ATT_BN) --a.kchunks;      // no empty run
    a.dq_part = (float*)((char*)workspace + bwd_dk_ws_bytes(BH, N, M, D, dk_f32 != nullptr));
    a.nwg = a.tiles * BH * a.kchunks;
    a.scale = scale;
    a.c = scale * 1.4426950408889634f;
    a.l2e = 1.4426950408889634f;
    hipStream_t st = as_stream(stream);
    if (dq2) {
        const bool pre = a.c == 1.0f;
        if (dtype == GD_F16) { if (pre) k_attn_bwd_dq2<f16_t, true><<<a.nwg, 256, 0, st>>>(a); else k_attn_bwd_dq2<f16_t, false><<<a.nwg, 256, 0, st>>>(a); }
        else { if (pre) k_attn_bwd_dq2<bf16_t, true><<<a.nwg, 256, 0, st>>>(a); else k_attn_bwd_dq2<bf16_t, false><<<a.nwg, 256, 0, st>>>(a); }
    } else {
        GD_LAUNCH_NCH(k_attn_bwd_dq, a.nwg, a);
    }
    if (kchunks_out) *kchunks_out = a.kchunks;
    if (dq_part_out) *dq_part_out = a.dq_part;
    if (a.kchunks > 1 && fold) {
        const long long n4 = (long long)BH * N * D / 4;
        const int fb = (int)((n4 + 255) / 256);
        if (dtype == GD_F16) k_attn_bwd_dq_fold<f16_t><<<fb, 256, 0, st>>>(a.dq_part, a.kchunks, n4, (f16_t*)dq);
        else k_attn_bwd_dq_fold<bf16_t><<<fb, 256, 0, st>>>(a.dq_part, a.kchunks, n4, (bf16_t*)dq);
    }
    if (dk_f32) {
        dim3 grid((N + DK_QCHUNK - 1) / DK_QCHUNK, BH);
        GD_LAUNCH_NCH(k_attn_bwd_dk, grid, a);
        dim3 rgrid((M * D + 255) / 256, BH);
        k_attn_bwd_dk_reduce<<<rgrid, 256, 0, st>>>((const float*)workspace, (int)grid.x, M * D, dk_f32, fold ? 0 : 1);
    }
    GD_CHECK_LAUNCH("gd_attn_bwd");
    return GD_OK;
}

// kchunks_out == NULL: complete (dq folded here, dk accumulated); != NULL: the dq partials of a split key range are left to gd_edit_dq_fold
extern "C" int gd_attn_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                           const void* dout, int BH, int N, int M, int D, float scale,
                           void* dq, float* dk_f32, void* workspace, size_t workspace_bytes, int* kchunks_out, float** dq_part_out,
                           int variant, int dtype, void* stream) {
    GD_REQUIRE((kchunks_out == nullptr) == (dq_part_out == nullptr), GD_EINVAL, "gd_attn_bwd: kchunks_out and dq_part_out go together");
    GD_REQUIRE(variant == 0 || variant == 1, GD_EINVAL, "gd_attn_bwd: variant %d (0, 1)", variant);
    return attn_bwd_launch(q, k, v, out, lse, dout, BH, N, M, D, scale, dq, dk_f32, workspace, workspace_bytes, dtype, stream,
                           kchunks_out == nullptr, kchunks_out, dq_part_out, variant);
}
