// R5/R6/R7 — software-pipelined flash-attention forward for full key tiles (gfx950 MFMA, D = 64): the kernel behind gd_attn_fwd on
// every self-attention launch of the UNet (M = 4096 / 1024 / 256).  Same contract as k_attn_fwd (attn_fwd.hip): replaces
// compute_attention + torch.bmm (GeoDiffuser/utils/attention_sharing.py:30-47, GeoDiffuser/utils/attention_processors.py:428,433,549,
// 557,644,647); up to 4 (q,k,v,out) segments per launch; token-major or head-major rows; optional fused query warp
// (U/attention_processors.py:424-428,544-549) in the prologue.
//
// Why this shape.  At head dim 64 a 64-key tile costs a wave 16 MFMAs (512 matrix-pipe cycles) but, in the textbook form, 32 x (fma, exp,
// add) + 16 packs of vector work: ~700 issue cycles (measured: tools/ub/ub_gap.hip, 49.5 cycles per MFMA + 2 probabilities on one
// wave, 46 with two waves per SIMD; MFMA alone 32.5).  The kernel is bound by vector ISSUE, so
//   (1) the scale is folded into the queries once per workgroup: Q' = 16-bit(c Q), c = scale log2(e), and the running reference mu is
//       the INITIAL ACCUMULATOR of the score MFMAs (S' = K Q'^T - mu), so a probability is ONE instruction, p = exp2(S'): no per-score
//       fma.  Cost: one extra 16-bit rounding of the queries (relative 2^-12 fp16 / 2^-9 bf16 per element, i.e. the class of the
//       storage rounding q and k already carry; the reference's own GPU path rounds every SCORE to fp16, U/attention_sharing.py:40);
//   (2) one wave keeps both pipes busy by modulo scheduling across tiles — iteration t (scores of tile t in registers), 16 "gaps" of
//       1 MFMA + 2 probabilities + 1 LDS fragment read, pinned with sched_barrier:
//         gaps  1- 4   O += V(t-1)[keys 32..63] P(t-1)      | exp2 of S'(t)[keys  0..15]
//         gaps  5- 8   S'(t+1)[keys 0..31]  = K(t+1) Q'^T   | exp2 of S'(t)[keys 16..31]      then the 16-bit range check
//         gaps  9-12   O += V(t)[keys 0..31] P(t)           | exp2 of S'(t)[keys 32..47]
//         gaps 13-16   S'(t+1)[keys 32..63] = K(t+1) Q'^T   | exp2 of S'(t)[keys 48..63]      range check
//       every MFMA operand is at least four gaps old, every exp2 input one iteration old;
//   (3) no per-step row maximum: the reference mu is raised (cold, wave-uniform "rescue" path) only when a half step's probability
//       sum leaves [0, 2^14], i.e. before probabilities could leave the 16-bit range; l, O, lse and the key-range merge are exact for
//       any reference;
//   (4) key tiles are staged two at a time where LDS allows (NT = 2: one barrier / load batch / store batch per 128 keys), global
//       loads are buffer loads with a wave-uniform descriptor + scalar tile offset (no per-thread address arithmetic in the loop).
// Under-filled launches (the single-row inversion pass: 5 heads x 4096 queries = 640 wave-sized query blocks for 1024 SIMDs) split the
// KEYS inside the workgroup: KS groups of QB waves, each with its own K/V ring over 1/KS of the keys, merged through LDS at the end
// — no f32 partials in HBM, no second kernel (the r01 split-KV launch moved 5.8x the algorithmic bytes).
//
// MFMA-bound target: algorithmic FLOPs = 4 * BH * N * M * D per launch.  Registers ~200 VGPRs: two waves per SIMD.
#include <stdlib.h>
#include "attn_common.hpp"

#define P_SUM_LIMIT 16384.0f
#define GD_SB() __builtin_amdgcn_sched_barrier(0)

// two probabilities -> one packed 16-bit pair, pinned where it is written: without the empty asm the compiler sinks the conversions of a
// half step below its range check (their results are dead on the rescue path) and they stop overlapping the MFMAs
template <typename T>
__device__ __forceinline__ uint32_t pack2(float p0, float p1) {
    typedef __attribute__((ext_vector_type(2))) decltype(elem_traits<T>::from_f32(0.f)) V2;
    V2 t;
    t[0] = elem_traits<T>::from_f32(p0);
    t[1] = elem_traits<T>::from_f32(p1);
    uint32_t w = __builtin_bit_cast(uint32_t, t);
    asm volatile("" : "+v"(w));
    return w;
}
template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec8 as_frag(const u32x4& w) {
    return __builtin_bit_cast(typename elem_traits<T>::vec8, w);
}

// two probabilities of a half step, partial row sums, packed pair.  PRE (queries pre-scaled, reference in the accumulator): p = exp2(S');
// otherwise p = exp2(c S - mu) with the raw scores in the accumulator (one more vector instruction per probability)
#define GD_EG2(SRC, i, DST, j)                                                                           \
    {                                                                                                    \
        const float p0_ = __builtin_amdgcn_exp2f(PRE ? SRC[i] : __builtin_fmaf(SRC[i], c, -m2));         \
        const float p1_ = __builtin_amdgcn_exp2f(PRE ? SRC[(i) + 1] : __builtin_fmaf(SRC[(i) + 1], c, -m2)); \
        DST[(j) >> 1] = pack2<T>(p0_, p1_);                                                              \
        ps0 += p0_;                                                                                      \
        ps1 += p1_;                                                                                      \
    }

// in-place vector op on all 16 registers of an accumulator tile.  Plain C++ (`x -= d`) would give the updated tile new virtual
// registers on the cold path, and the register allocator then copies whole tiles on the FAST path to rejoin the two versions.
#define GD_TILE_OP(OP, TILE, SCALAR)                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                            \
        float t_ = TILE[i_];                                                       \
        asm volatile(OP " %0, %0, %1" : "+v"(t_) : "v"(SCALAR));                   \
        TILE[i_] = t_;                                                             \
    }

// Cold path of the reference-value softmax: a half step H (16 scores per lane) produced probabilities whose sum left [0, 2^14].
// Raise the reference mu (m2, log2 domain) by the half's maximum exponent mx (> 0), rescale O and l by 2^-mx, recompute the half's
// probabilities.  PRE: the scores in registers carry the bias -mu, so every score tile that is live or about to be written (the
// current tile's X0/X1, the next tile's Y0/Y1) and the bias tile negmu itself are re-biased in place.  Wave-uniform.  The asm blocks
// carry their own wait states: the hazard recogniser does not look inside asm (an MFMA result needs up to 18 wait states before a
// VALU read, a transcendental result one).
template <typename T, bool PRE>
__device__ __forceinline__ void softmax_rescue(f32x16& H, f32x16& S1, f32x16& S2, f32x16& S3, f32x16& negmu, f32x16 (&o)[2],
                                               float& m2, float& l_run, const float c, u32x4& pf0, u32x4& pf1, float& ps) {
    float mx = H[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, H[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (!PRE) mx = __builtin_fmaf(mx, c, -m2);             // largest exponent of the half against the current reference
    mx = fmaxf(mx, 0.f);                                   // the reference only ever rises
    float alpha = __builtin_amdgcn_exp2f(-mx);
    l_run *= alpha;
    m2 += mx;
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(alpha), "+v"(mx));      // in-flight MFMA results + the v_exp_f32 above have landed
    GD_TILE_OP("v_mul_f32", o[0], alpha)
    GD_TILE_OP("v_mul_f32", o[1], alpha)
    if (PRE) {
        GD_TILE_OP("v_sub_f32", H, mx)
        GD_TILE_OP("v_sub_f32", S1, mx)
        GD_TILE_OP("v_sub_f32", S2, mx)
        GD_TILE_OP("v_sub_f32", S3, mx)
        GD_TILE_OP("v_sub_f32", negmu, mx)
    }
    asm volatile("s_nop 7" ::: "memory");                              // VALU writes above -> MFMA reads below
    ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(PRE ? H[i] : __builtin_fmaf(H[i], c, -m2));
        const float p1 = __builtin_amdgcn_exp2f(PRE ? H[i + 1] : __builtin_fmaf(H[i + 1], c, -m2));
        const float p2 = __builtin_amdgcn_exp2f(PRE ? H[8 + i] : __builtin_fmaf(H[8 + i], c, -m2));
        const float p3 = __builtin_amdgcn_exp2f(PRE ? H[8 + i + 1] : __builtin_fmaf(H[8 + i + 1], c, -m2));
        pf0[i >> 1] = pack2<T>(p0, p1);
        pf1[i >> 1] = pack2<T>(p2, p3);
        ps += (p0 + p1) + (p2 + p3);
    }
}

// One pipelined iteration: consumes the biased scores X0/X1 of tile t and the K tile t+1 (lk) / V tile t (lv) in LDS, produces the
// scores Y0/Y1 of tile t+1 (unless LAST), finishes tile t-1's P.V from the carried fragments (pb0/pb1 probabilities, vc V fragments)
// and leaves tile t's second half in them.
template <typename T, bool LAST, bool PRE>
__device__ __forceinline__ void mp_iter(const char* lk, const char* lv, const FragOffs& fo, const typename elem_traits<T>::vec8 (&qf)[4],
                                        f32x16& X0, f32x16& X1, f32x16& Y0, f32x16& Y1, f32x16& negmu, f32x16 (&o)[2], float& m2,
                                        float& l_run, const float c, u32x4& pb0, u32x4& pb1, typename elem_traits<T>::vec8 (&vc)[4]) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    u32x4 pa0, pa1;
    V8 kf[4], va[4];
    float ps0 = 0.f, ps1 = 0.f;
    // ---- first half of tile t: keys 0..31 (X0) ----
    if (!LAST) kf[0] = rd_row<T>(lk, fo, 0, 0);
    o[0] = TR::mfma32(vc[0], as_frag<T>(pb0), o[0]); GD_EG2(X0, 0, pa0, 0); GD_SB();
    if (!LAST) kf[1] = rd_row<T>(lk, fo, 0, 1);
    o[1] = TR::mfma32(vc[1], as_frag<T>(pb0), o[1]); GD_EG2(X0, 2, pa0, 2); GD_SB();
    if (!LAST) kf[2] = rd_row<T>(lk, fo, 0, 2);
    o[0] = TR::mfma32(vc[2], as_frag<T>(pb1), o[0]); GD_EG2(X0, 4, pa0, 4); GD_SB();
    if (!LAST) kf[3] = rd_row<T>(lk, fo, 0, 3);
    o[1] = TR::mfma32(vc[3], as_frag<T>(pb1), o[1]); GD_EG2(X0, 6, pa0, 6); GD_SB();
    va[0] = rd_tr<T>(lv, fo, 0, 0);
    if (!LAST) Y0 = TR::mfma32(kf[0], qf[0], negmu);
    GD_EG2(X0, 8, pa1, 0); GD_SB();
    va[1] = rd_tr<T>(lv, fo, 1, 0);
    if (!LAST) Y0 = TR::mfma32(kf[1], qf[1], Y0);
    GD_EG2(X0, 10, pa1, 2); GD_SB();
    va[2] = rd_tr<T>(lv, fo, 0, 1);
    if (!LAST) Y0 = TR::mfma32(kf[2], qf[2], Y0);
    GD_EG2(X0, 12, pa1, 4); GD_SB();
    va[3] = rd_tr<T>(lv, fo, 1, 1);
    if (!LAST) Y0 = TR::mfma32(kf[3], qf[3], Y0);
    GD_EG2(X0, 14, pa1, 6); GD_SB();
    {
        float ps = ps0 + ps1;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ps <= P_SUM_LIMIT)) != 0, 0))
            softmax_rescue<T, PRE>(X0, X1, Y0, Y1, negmu, o, m2, l_run, c, pa0, pa1, ps);
        l_run += ps;
    }
    GD_SB();
    // ---- second half: keys 32..63 (X1) ----
    ps0 = 0.f; ps1 = 0.f;
    if (!LAST) kf[0] = rd_row<T>(lk, fo, 1, 0);
    o[0] = TR::mfma32(va[0], as_frag<T>(pa0), o[0]); GD_EG2(X1, 0, pb0, 0); GD_SB();
    if (!LAST) kf[1] = rd_row<T>(lk, fo, 1, 1);
    o[1] = TR::mfma32(va[1], as_frag<T>(pa0), o[1]); GD_EG2(X1, 2, pb0, 2); GD_SB();
    if (!LAST) kf[2] = rd_row<T>(lk, fo, 1, 2);
    o[0] = TR::mfma32(va[2], as_frag<T>(pa1), o[0]); GD_EG2(X1, 4, pb0, 4); GD_SB();
    if (!LAST) kf[3] = rd_row<T>(lk, fo, 1, 3);
    o[1] = TR::mfma32(va[3], as_frag<T>(pa1), o[1]); GD_EG2(X1, 6, pb0, 6); GD_SB();
    vc[0] = rd_tr<T>(lv, fo, 0, 2);
    if (!LAST) Y1 = TR::mfma32(kf[0], qf[0], negmu);
    GD_EG2(X1, 8, pb1, 0); GD_SB();
    vc[1] = rd_tr<T>(lv, fo, 1, 2);
    if (!LAST) Y1 = TR::mfma32(kf[1], qf[1], Y1);
    GD_EG2(X1, 10, pb1, 2); GD_SB();
    vc[2] = rd_tr<T>(lv, fo, 0, 3);
    if (!LAST) Y1 = TR::mfma32(kf[2], qf[2], Y1);
    GD_EG2(X1, 12, pb1, 4); GD_SB();
    vc[3] = rd_tr<T>(lv, fo, 1, 3);
    if (!LAST) Y1 = TR::mfma32(kf[3], qf[3], Y1);
    GD_EG2(X1, 14, pb1, 6); GD_SB();
    {
        float ps = ps0 + ps1;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ps <= P_SUM_LIMIT)) != 0, 0))
            softmax_rescue<T, PRE>(X1, X0, Y0, Y1, negmu, o, m2, l_run, c, pb0, pb1, ps);
        l_run += ps;
    }
}

// NT = key tiles staged per barrier (1 or 2).  LDS per key range: NT = 1: K ring [2] + V ring [2] = 32 KB; NT = 2: two sets of
// {K(2p+1), K(2p+2), V(2p), V(2p+1)} = 64 KB.  Key tiles per range: even (NT = 1) / a multiple of 4 (NT = 2).
// PRE: probabilities as ONE instruction (queries carry c, the reference is the score MFMAs' initial accumulator).  The queries either
// arrive pre-scaled (FwdArgs::q_prescaled: the projection GEMM applied c in its fp32 epilogue, one rounding) or are scaled here (one
// EXTRA 16-bit rounding of c q: score error 2^-9 |score| in bf16 — only on request, GD_ATTN_PRESCALE=1).  !PRE: exact scores, p =
// exp2(c S - mu).
template <typename T, int QB, int KS, int NT, bool PRE>
__global__ void __launch_bounds__(QB * KS * 64, 2)
k_attn_fwd_mp(const FwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int GT = QB * 64;                 // threads of one key-range group
    constexpr int CPT = 512 / GT;               // 16-byte chunks of a 64 x 64 tile per thread
    constexpr int BMQ = QB * 32;                // query rows per workgroup
    __shared__ __attribute__((aligned(16))) char lds[KS][NT * 4][ATT_TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / QB, qb = wave - ks * QB, gtid = tid - ks * GT;
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const int gbh = wg / a.tiles, tile = wg - gbh * a.tiles;
    int sidx = 0, bh;
    if (a.n_order > 0) {                                   // interleaved head order (see gd_attn_fwd_mp_launch)
        const int code = a.order[gbh];
        sidx = code >> 12;
        bh = code & 4095;
    } else {
#pragma unroll
        for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
            if (i < a.nseg - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
        bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
    }
    const gd_attn_seg_t sg = a.seg[sidx];
    const int N = a.N, M = a.M;
    // row stride / base offsets: head-major [bh, N, 64] or token-major [B, N, heads*64]
    const int rs = sg.heads > 0 ? sg.heads * ATT_D : ATT_D;
    size_t qoff, koff;
    if (sg.heads > 0) {
        const int b = bh / sg.heads, hh = bh - b * sg.heads;
        qoff = (size_t)b * N * rs + (size_t)hh * ATT_D;
        koff = (size_t)b * M * rs + (size_t)hh * ATT_D;
    } else {
        qoff = (size_t)bh * N * ATT_D;
        koff = (size_t)bh * M * ATT_D;
    }
    const T* __restrict__ qp = (const T*)sg.q + qoff;
    const int Tg = (M / ATT_BN) / KS;                                       // key tiles of this key range
    // wave-uniform buffer descriptors (scalar registers) over this key range's K / V rows: buffer_load takes the per-thread part as a
    // 32-bit offset register and the tile step as a scalar offset, so the loop carries no per-thread address arithmetic
    const __amdgpu_buffer_rsrc_t kb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)sg.k + koff + (size_t)ks * Tg * ATT_BN * rs), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t vb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)sg.v + koff + (size_t)ks * Tg * ATT_BN * rs), 0, 0x7FFFFFFF, 0x00020000);
    const int tB = ATT_BN * rs * (int)sizeof(T);                            // bytes from one key tile to the next
    uint32_t voff[CPT];
    int loff[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        voff[j] = (uint32_t)(((unsigned)((gtid + j * GT) >> 3) * (unsigned)rs + (unsigned)(gtid & 7) * 8u) * (unsigned)sizeof(T));
        loff[j] = img_off((gtid + j * GT) >> 3, gtid & 7);
    }
#define GD_GLD(BASE, TILE, R) \
    _Pragma("unroll") for (int j = 0; j < CPT; ++j) R[j] = __builtin_amdgcn_raw_buffer_load_b128(BASE, voff[j], (TILE) * tB, 0)
#define GD_LST(DST, R) _Pragma("unroll") for (int j = 0; j < CPT; ++j) *(u32x4*)((DST) + loff[j]) = R[j]
    // LDS tiles.  NT = 1: K ring lK0/lK1, V ring lV0/lV1.  NT = 2: set A = tiles 0-3 {Ka, Kb, Va, Vb}, set B = tiles 4-7.
    char* const L0 = lds[ks][0]; char* const L1 = lds[ks][1]; char* const L2 = lds[ks][2]; char* const L3 = lds[ks][3];
    char* const L4 = lds[ks][NT == 2 ? 4 : 0]; char* const L5 = lds[ks][NT == 2 ? 5 : 1];
    char* const L6 = lds[ks][NT == 2 ? 6 : 2]; char* const L7 = lds[ks][NT == 2 ? 7 : 3];

    u32x4 kr0[CPT], kr1[CPT], vr0[CPT], vr1[CPT];
    char* k0_at;                                                            // where K(0) sits for the prologue's score GEMM
    if (NT == 2) {
        GD_GLD(kb, 0, kr0); GD_GLD(kb, 1, kr1); GD_GLD(vb, 0, vr0); GD_GLD(vb, 1, vr1);
        GD_LST(L4, kr0); GD_LST(L0, kr1); GD_LST(L2, vr0); GD_LST(L3, vr1);         // K(0) borrows set B's Ka until the first pair ends
        GD_GLD(kb, 2, kr0);
        GD_LST(L1, kr0);
        k0_at = L4;
    } else {                                                                // lK0 = L0, lK1 = L1, lV0 = L2, lV1 = L3
        GD_GLD(kb, 0, kr0); GD_GLD(kb, 1, kr1); GD_GLD(vb, 0, vr0);
        GD_LST(L0, kr0); GD_LST(L1, kr1); GD_LST(L2, vr0);
        k0_at = L0;
    }

    const int qrow = tile * BMQ + qb * 32 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    V8 qf[4];
    load_q_frags<T>(sg, qp, rs, qld, h, qf);
    if (PRE && !a.q_prescaled) {         // Q' = 16-bit(c Q): the softmax scale and log2(e) folded into the queries here
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = TR::from_f32(TR::to_f32(qf[s][j]) * a.c);
    }
    const float c = (PRE || a.q_prescaled) ? 1.0f : a.c;      // what is left to apply to the accumulated scores
    const FragOffs fo = make_frag_offs(lane);
    __syncthreads();

    f32x16 o[2], X0, X1, Y0, Y1, negmu;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; X0[i] = 0.f; X1[i] = 0.f; Y0[i] = 0.f; Y1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) X0 = TR::mfma32(rd_row<T>(k0_at, fo, 0, s), qf[s], X0);
#pragma unroll
    for (int s = 0; s < 4; ++s) X1 = TR::mfma32(rd_row<T>(k0_at, fo, 1, s), qf[s], X1);
    // the reference starts at the exact row maximum of tile 0 (log2 domain), so the loop never sees an infinite reference
    float m2 = X0[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m2 = fmaxf(m2, X0[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m2 = fmaxf(m2, X1[i]);
    m2 = fmaxf(m2, __shfl_xor(m2, 32, 64)) * c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (PRE) { X0[i] -= m2; X1[i] -= m2; }
        negmu[i] = PRE ? -m2 : 0.f;
    }
    float l_run = 0.f;
    u32x4 pb0 = {0u, 0u, 0u, 0u}, pb1 = {0u, 0u, 0u, 0u};
    V8 vc[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) vc[j][i] = TR::from_f32(0.f);
    __syncthreads();                  // every wave has read K(0) before the loop's first stores reuse its tile

#define GD_ITER(LAST_, LK, LV, SA0, SA1, SB0, SB1) \
    mp_iter<T, LAST_, PRE>(LK, LV, fo, qf, SA0, SA1, SB0, SB1, negmu, o, m2, l_run, c, pb0, pb1, vc)
    if (NT == 2) {
        // pairs of tiles; pair p reads {Ka = K(2p+1), Kb = K(2p+2), Va = V(2p), Vb = V(2p+1)} from one set while the next pair's
        // tiles arrive in registers and are stored to the other set before the pair's single barrier.  The last two pairs are
        // peeled: no control flow inside the loop besides the cold rescue branches.
        int t = 0;                    // first tile of the pair on set A
#pragma unroll 1
        for (; t + 4 < Tg; t += 4) {
            GD_GLD(kb, t + 3, kr0); GD_GLD(kb, t + 4, kr1); GD_GLD(vb, t + 2, vr0); GD_GLD(vb, t + 3, vr1);
            GD_ITER(false, L0, L2, X0, X1, Y0, Y1);
            GD_ITER(false, L1, L3, Y0, Y1, X0, X1);
            GD_LST(L4, kr0); GD_LST(L5, kr1); GD_LST(L6, vr0); GD_LST(L7, vr1);
            __syncthreads();
            GD_GLD(kb, t + 5, kr0); GD_GLD(kb, t + 6, kr1); GD_GLD(vb, t + 4, vr0); GD_GLD(vb, t + 5, vr1);
            GD_ITER(false, L4, L6, X0, X1, Y0, Y1);
            GD_ITER(false, L5, L7, Y0, Y1, X0, X1);
            GD_LST(L0, kr0); GD_LST(L1, kr1); GD_LST(L2, vr0); GD_LST(L3, vr1);
            __syncthreads();
        }
        GD_GLD(kb, t + 3, kr0); GD_GLD(vb, t + 2, vr0); GD_GLD(vb, t + 3, vr1);       // the final pair has no K(Tg)
        GD_ITER(false, L0, L2, X0, X1, Y0, Y1);
        GD_ITER(false, L1, L3, Y0, Y1, X0, X1);
        GD_LST(L4, kr0); GD_LST(L6, vr0); GD_LST(L7, vr1);
        __syncthreads();
        GD_ITER(false, L4, L6, X0, X1, Y0, Y1);
        GD_ITER(true, L5, L7, Y0, Y1, X0, X1);
    } else {
        // one tile per barrier: iteration t reads K(t+1) from lK[(t+1)&1] and V(t) from lV[t&1]; K(t+2) / V(t+1) arrive in registers
        // and are stored before the barrier.  lK0 = L0, lK1 = L1, lV0 = L2, lV1 = L3.  Last pair peeled.
        int t = 0;
#pragma unroll 1
        for (; t + 2 < Tg; t += 2) {
            GD_GLD(kb, t + 2, kr0); GD_GLD(vb, t + 1, vr0);
            GD_ITER(false, L1, L2, X0, X1, Y0, Y1);
            GD_LST(L0, kr0); GD_LST(L3, vr0);
            __syncthreads();
            GD_GLD(kb, t + 3, kr0); GD_GLD(vb, t + 2, vr0);
            GD_ITER(false, L0, L3, Y0, Y1, X0, X1);
            GD_LST(L1, kr0); GD_LST(L2, vr0);
            __syncthreads();
        }
        GD_GLD(vb, t + 1, vr0);
        GD_ITER(false, L1, L2, X0, X1, Y0, Y1);
        GD_LST(L3, vr0);
        __syncthreads();
        GD_ITER(true, L0, L3, Y0, Y1, X0, X1);
    }
#undef GD_ITER
#undef GD_GLD
#undef GD_LST
    // second half of the last tile
    o[0] = TR::mfma32(vc[0], as_frag<T>(pb0), o[0]);
    o[1] = TR::mfma32(vc[1], as_frag<T>(pb0), o[1]);
    o[0] = TR::mfma32(vc[2], as_frag<T>(pb1), o[0]);
    o[1] = TR::mfma32(vc[3], as_frag<T>(pb1), o[1]);

    if (KS > 1) {
        // merge the KS key ranges of each query block through LDS (exact for any per-range reference values)
        float* const ws = (float*)&lds[0][0][0];                        // (KS-1)*QB slots of [34][64] f32, aliasing the tile rings
        __syncthreads();
        if (ks > 0) {
            float* sl = ws + (size_t)((ks - 1) * QB + qb) * 34 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sl[i * 64] = o[0][i]; sl[(16 + i) * 64] = o[1][i]; }
            sl[32 * 64] = m2; sl[33 * 64] = l_run;
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll 1
        for (int k2 = 1; k2 < KS; ++k2) {
            const float* sl = ws + (size_t)((k2 - 1) * QB + qb) * 34 * 64 + lane;
            const float mk = sl[32 * 64], lk2 = sl[33 * 64];
            const float mn = fmaxf(m2, mk);
            const float a0 = __builtin_amdgcn_exp2f(m2 - mn), a1 = __builtin_amdgcn_exp2f(mk - mn);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o[0][i] = __builtin_fmaf(o[0][i], a0, sl[i * 64] * a1);
                o[1][i] = __builtin_fmaf(o[1][i], a0, sl[(16 + i) * 64] * a1);
            }
            l_run = __builtin_fmaf(l_run, a0, lk2 * a1);
            m2 = mn;
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < N) {
        T* __restrict__ op = (T*)sg.out + qoff + (size_t)qrow * rs;
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(o[dblk][4 * g + j] * inv);
                *(typename TR::vec4*)(op + dblk * 32 + 8 * g + 4 * h) = w;
            }
        // natural-log sum-exp of the scaled scores: m2 is in the log2 domain
        if (sg.lse && h == 0) sg.lse[(size_t)bh * N + qrow] = m2 * 0.6931471805599453f + __logf(l_tot);
    }
}

int gd_attn_fwd_mp_launch(FwdArgs a, int qb, int ks, int dtype, hipStream_t st) {
    a.tiles = (a.N + 32 * qb - 1) / (32 * qb);
    int tot = 0;
    for (int i = 0; i < a.nseg; ++i) tot = a.bh_end[i];
    a.nwg = a.tiles * tot;
    // Head order of the grid.  The 1-D grid is cut into 8 contiguous chunks, one per XCD (xcd_remap), and an XCD runs its chunk in
    // ascending order.  Segment after segment, the heads whose workgroups build warped queries in their prologue (~8 us of dependent
    // gathers) all land on two XCDs, which then finish last (measured: +17 us on a 106 us launch).  Interleaving the segments by
    // relative position spreads them over the XCDs, puts a warped head before the plain heads of its neighbourhood, and places the
    // edit_out / replace_out heads that share k_base / v_base on the same XCD's L2.
    a.n_order = 0;
    if (a.nseg > 1 && tot <= GD_ATTN_MAX_ORDER) {
        float key[GD_ATTN_MAX_ORDER];
        int n = 0, start = 0;
        for (int sgi = 0; sgi < a.nseg; ++sgi) {
            const int cnt = a.bh_end[sgi] - start;
            for (int i = 0; i < cnt; ++i) {
                key[n] = ((float)i + (a.seg[sgi].warp_idx ? 0.25f : 0.5f)) / (float)cnt + 1e-4f * (float)sgi;
                a.order[n++] = (unsigned short)((sgi << 12) | i);
            }
            start = a.bh_end[sgi];
        }
        for (int i = 1; i < n; ++i) {                                        // insertion sort by key (n <= 160)
            const float kx = key[i];
            const unsigned short ox = a.order[i];
            int j = i - 1;
            for (; j >= 0 && key[j] > kx; --j) { key[j + 1] = key[j]; a.order[j + 1] = a.order[j]; }
            key[j + 1] = kx; a.order[j + 1] = ox;
        }
        a.n_order = n;
    }
    const int Tg = (a.M / ATT_BN) / ks;
    GD_REQUIRE(a.M % ATT_BN == 0 && (a.M / ATT_BN) % ks == 0 && Tg >= 2 && Tg % 2 == 0, GD_EINVAL,
               "gd_attn_fwd: the pipelined kernel needs an even number of full key tiles per key range (M=%d, KS=%d)", a.M, ks);
    const int nt = (ks <= 2 && Tg % 4 == 0) ? 2 : 1;                      // two tiles per barrier where LDS (64 KB per key range) allows
    static int env_pre = -1;
    if (env_pre < 0) { const char* e = getenv("GD_ATTN_PRESCALE"); env_pre = e ? atoi(e) : 0; }
    const int pre = (a.q_prescaled || env_pre) ? 1 : 0;
#define GD_MP_LAUNCH(QB_, KS_, NT_, PRE_)                                                                                \
    {                                                                                                                    \
        if (dtype == GD_F16) k_attn_fwd_mp<f16_t, QB_, KS_, NT_, PRE_><<<a.nwg, QB_ * KS_ * 64, 0, st>>>(a);              \
        else k_attn_fwd_mp<bf16_t, QB_, KS_, NT_, PRE_><<<a.nwg, QB_ * KS_ * 64, 0, st>>>(a);                            \
    }
#define GD_MP_CASE(QB_, KS_, NT_)                                                                                        \
    case (QB_ * 10 + KS_) * 10 + NT_:                                                                                    \
        if (pre) GD_MP_LAUNCH(QB_, KS_, NT_, true) else GD_MP_LAUNCH(QB_, KS_, NT_, false)                               \
        break;
    switch ((qb * 10 + ks) * 10 + nt) {
        GD_MP_CASE(4, 1, 1) GD_MP_CASE(4, 1, 2)
        GD_MP_CASE(2, 2, 1) GD_MP_CASE(2, 2, 2)
        GD_MP_CASE(4, 2, 1) GD_MP_CASE(4, 2, 2)
        GD_MP_CASE(2, 4, 1)
        default: GD_REQUIRE(false, GD_EINVAL, "gd_attn_fwd: no pipelined kernel for QB=%d KS=%d", qb, ks);
    }
#undef GD_MP_LAUNCH
#undef GD_MP_CASE
    GD_CHECK_LAUNCH("gd_attn_fwd");
    return GD_OK;
}
