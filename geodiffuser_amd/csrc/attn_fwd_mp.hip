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

// Timing-only builds (tools/build_dbg.sh; results are WRONG, never shipped): bit 0 no K/V loads or LDS stores in the loop, bit 1 no
// barriers in the loop, bit 2 no range checks, bit 3 fragment reads replaced by register moves
#ifndef GD_MP_DBG
#define GD_MP_DBG 0
#endif
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
    if (!LAST) kf[0] = ((GD_MP_DBG & 8) ? qf[0] : rd_row<T>(lk, fo, 0, 0));
    o[0] = TR::mfma32(vc[0], as_frag<T>(pb0), o[0]); GD_EG2(X0, 0, pa0, 0); GD_SB();
    if (!LAST) kf[1] = ((GD_MP_DBG & 8) ? qf[1] : rd_row<T>(lk, fo, 0, 1));
    o[1] = TR::mfma32(vc[1], as_frag<T>(pb0), o[1]); GD_EG2(X0, 2, pa0, 2); GD_SB();
    if (!LAST) kf[2] = ((GD_MP_DBG & 8) ? qf[2] : rd_row<T>(lk, fo, 0, 2));
    o[0] = TR::mfma32(vc[2], as_frag<T>(pb1), o[0]); GD_EG2(X0, 4, pa0, 4); GD_SB();
    if (!LAST) kf[3] = ((GD_MP_DBG & 8) ? qf[3] : rd_row<T>(lk, fo, 0, 3));
    o[1] = TR::mfma32(vc[3], as_frag<T>(pb1), o[1]); GD_EG2(X0, 6, pa0, 6); GD_SB();
    va[0] = ((GD_MP_DBG & 8) ? qf[0] : rd_tr<T>(lv, fo, 0, 0));
    if (!LAST) Y0 = TR::mfma32(kf[0], qf[0], negmu);
    GD_EG2(X0, 8, pa1, 0); GD_SB();
    va[1] = ((GD_MP_DBG & 8) ? qf[0] : rd_tr<T>(lv, fo, 1, 0));
    if (!LAST) Y0 = TR::mfma32(kf[1], qf[1], Y0);
    GD_EG2(X0, 10, pa1, 2); GD_SB();
    va[2] = ((GD_MP_DBG & 8) ? qf[1] : rd_tr<T>(lv, fo, 0, 1));
    if (!LAST) Y0 = TR::mfma32(kf[2], qf[2], Y0);
    GD_EG2(X0, 12, pa1, 4); GD_SB();
    va[3] = ((GD_MP_DBG & 8) ? qf[1] : rd_tr<T>(lv, fo, 1, 1));
    if (!LAST) Y0 = TR::mfma32(kf[3], qf[3], Y0);
    GD_EG2(X0, 14, pa1, 6); GD_SB();
    {
        float ps = ps0 + ps1;
        if (!(GD_MP_DBG & 4) && __builtin_expect(__builtin_amdgcn_ballot_w64(!(ps <= P_SUM_LIMIT)) != 0, 0))
            softmax_rescue<T, PRE>(X0, X1, Y0, Y1, negmu, o, m2, l_run, c, pa0, pa1, ps);
        l_run += ps;
    }
    GD_SB();
    // ---- second half: keys 32..63 (X1) ----
    ps0 = 0.f; ps1 = 0.f;
    if (!LAST) kf[0] = ((GD_MP_DBG & 8) ? qf[0] : rd_row<T>(lk, fo, 1, 0));
    o[0] = TR::mfma32(va[0], as_frag<T>(pa0), o[0]); GD_EG2(X1, 0, pb0, 0); GD_SB();
    if (!LAST) kf[1] = ((GD_MP_DBG & 8) ? qf[1] : rd_row<T>(lk, fo, 1, 1));
    o[1] = TR::mfma32(va[1], as_frag<T>(pa0), o[1]); GD_EG2(X1, 2, pb0, 2); GD_SB();
    if (!LAST) kf[2] = ((GD_MP_DBG & 8) ? qf[2] : rd_row<T>(lk, fo, 1, 2));
    o[0] = TR::mfma32(va[2], as_frag<T>(pa1), o[0]); GD_EG2(X1, 4, pb0, 4); GD_SB();
    if (!LAST) kf[3] = ((GD_MP_DBG & 8) ? qf[3] : rd_row<T>(lk, fo, 1, 3));
    o[1] = TR::mfma32(va[3], as_frag<T>(pa1), o[1]); GD_EG2(X1, 6, pb0, 6); GD_SB();
    vc[0] = ((GD_MP_DBG & 8) ? qf[2] : rd_tr<T>(lv, fo, 0, 2));
    if (!LAST) Y1 = TR::mfma32(kf[0], qf[0], negmu);
    GD_EG2(X1, 8, pb1, 0); GD_SB();
    vc[1] = ((GD_MP_DBG & 8) ? qf[2] : rd_tr<T>(lv, fo, 1, 2));
    if (!LAST) Y1 = TR::mfma32(kf[1], qf[1], Y1);
    GD_EG2(X1, 10, pb1, 2); GD_SB();
    vc[2] = ((GD_MP_DBG & 8) ? qf[3] : rd_tr<T>(lv, fo, 0, 3));
    if (!LAST) Y1 = TR::mfma32(kf[2], qf[2], Y1);
    GD_EG2(X1, 12, pb1, 4); GD_SB();
    vc[3] = ((GD_MP_DBG & 8) ? qf[3] : rd_tr<T>(lv, fo, 1, 3));
    if (!LAST) Y1 = TR::mfma32(kf[3], qf[3], Y1);
    GD_EG2(X1, 14, pb1, 6); GD_SB();
    {
        float ps = ps0 + ps1;
        if (!(GD_MP_DBG & 4) && __builtin_expect(__builtin_amdgcn_ballot_w64(!(ps <= P_SUM_LIMIT)) != 0, 0))
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
// SK: the key tiles of the whole launch are dealt out evenly (FwdArgs::sk_tpw per workgroup, see attn_common.hpp): a workgroup walks
// one or more SEGMENTS (a run of key tiles inside one unit); a unit that ends up in several workgroups is merged by the last to arrive.
template <typename T, int QB, int KS, int NT, bool PRE, bool SK>
__global__ void __launch_bounds__(QB * KS * 64, 2)
k_attn_fwd_mp(const FwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    static_assert(!SK || (KS == 1 && QB == 4 && NT == 2), "the even split runs on the 128-query workgroup");
    constexpr int GT = QB * 64;                 // threads of one key-range group
    constexpr int CPT = 512 / GT;               // 16-byte chunks of a 64 x 64 tile per thread
    constexpr int BMQ = QB * 32;                // query rows per workgroup
    __shared__ __attribute__((aligned(16))) char lds[KS][NT * 4][ATT_TILE_BYTES];
    __shared__ int sk_last;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / QB, qb = wave - ks * QB, gtid = tid - ks * GT;
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const int TU = a.M / ATT_BN;                                           // key tiles of a unit
    int lin = SK ? wg * a.sk_tpw : 0;                                      // SK: this workgroup's run of the launch's linear tile range
    const int lin_end = SK ? (lin + a.sk_tpw < a.sk_total ? lin + a.sk_tpw : a.sk_total) : 0;
    const int lin_first = lin;
#if GD_MP_DBG & 32
    int dbg_seg = 0;
#define MP_STAMP(E) { if (tid == 0 && a.sk_ws) ((unsigned long long*)((char*)a.sk_ws + GD_SK_SLOT_BYTES + (1u << 20)))[(wg * 4 + dbg_seg) * 8 + (E)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define MP_STAMP(E) {}
#endif
#pragma unroll 1
  do {
    MP_STAMP(0)
    int unit = wg, t_first = 0, Tg = TU / KS;                              // key tiles [t_first, t_first + Tg) of this segment
    if (SK) {
        unit = lin / TU;
        t_first = lin - unit * TU;
        Tg = TU - t_first < lin_end - lin ? TU - t_first : lin_end - lin;
    } else {
        t_first = ks * Tg;
    }
    int sidx = 0, bh, tile;
    const bool compact = a.cseg >= 0 && unit >= a.units_full;      // the segment with a query row list: units after everyone else's
    if (compact) {
        compact_unit(a, unit, sidx, bh, tile);
    } else {
        const int gbh = unit / a.tiles;
        tile = unit - gbh * a.tiles;
        if (a.n_order > 0) {                               // interleaved head order (see gd_attn_fwd_mp_launch)
            const int code = a.order[gbh];
            sidx = code >> 12;
            bh = code & 4095;
        } else {
#pragma unroll
            for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
                if (i < a.nseg - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
            bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
        }
    }
    const gd_attn_seg_t sg = a.seg[sidx];
    const int N = a.N, M = a.M;
    const int Nq = compact ? sg.q_rows_len : N;            // rows of this segment's query list / of its dense output
    // row stride / base offsets: head-major [bh, N, 64] or token-major [B, N, heads*64]
    const int rs = sg.heads > 0 ? sg.heads * ATT_D : ATT_D;
    size_t qoff, koff, ooff;
    if (sg.heads > 0) {
        const int b = bh / sg.heads, hh = bh - b * sg.heads;
        qoff = (size_t)b * N * rs + (size_t)hh * ATT_D;
        koff = (size_t)b * M * rs + (size_t)hh * ATT_D;
        ooff = (size_t)b * Nq * rs + (size_t)hh * ATT_D;
    } else {
        qoff = (size_t)bh * N * ATT_D;
        koff = (size_t)bh * M * ATT_D;
        ooff = (size_t)bh * Nq * ATT_D;
    }
    const T* __restrict__ qp = (const T*)sg.q + qoff;
    // wave-uniform buffer descriptors (scalar registers) over this key range's K / V rows: buffer_load takes the per-thread part as a
    // 32-bit offset register and the tile step as a scalar offset, so the loop carries no per-thread address arithmetic
    const __amdgpu_buffer_rsrc_t kb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)sg.k + koff + (size_t)t_first * ATT_BN * rs), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t vb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)sg.v + koff + (size_t)t_first * ATT_BN * rs), 0, 0x7FFFFFFF, 0x00020000);
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

    const int qrow = tile * BMQ + qb * 32 + (lane & 31);   // row of the output (= of the row list, for a compact segment)
    const int qok = qrow < (compact ? sg.q_rows_n[0] : N);
    const int qld = compact ? sg.q_rows[qrow < Nq ? qrow : Nq - 1] : (qrow < N ? qrow : N - 1);       // row of q it is computed from
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
    MP_STAMP(1)

#define GD_GLD_L(BASE, TILE, R) if (!(GD_MP_DBG & 1)) { GD_GLD(BASE, TILE, R); }
#define GD_LST_L(DST, R) if (!(GD_MP_DBG & 1)) { GD_LST(DST, R); }
#define GD_SYNC_L() if (!(GD_MP_DBG & 2)) __syncthreads()
#define GD_ITER(LAST_, LK, LV, SA0, SA1, SB0, SB1) \
    mp_iter<T, LAST_, PRE>(LK, LV, fo, qf, SA0, SA1, SB0, SB1, negmu, o, m2, l_run, c, pb0, pb1, vc)
    if (NT == 2) {
        // pairs of tiles; pair p reads {Ka = K(2p+1), Kb = K(2p+2), Va = V(2p), Vb = V(2p+1)} from one set while the next pair's
        // tiles arrive in registers and are stored to the other set before the pair's single barrier.  The last two pairs are
        // peeled: no control flow inside the loop besides the cold rescue branches.
        int t = 0;                    // first tile of the pair on set A
#pragma unroll 1
        for (; t + 4 < Tg; t += 4) {
            GD_GLD_L(kb, t + 3, kr0); GD_GLD_L(kb, t + 4, kr1); GD_GLD_L(vb, t + 2, vr0); GD_GLD_L(vb, t + 3, vr1);
            GD_ITER(false, L0, L2, X0, X1, Y0, Y1);
            GD_ITER(false, L1, L3, Y0, Y1, X0, X1);
            GD_LST_L(L4, kr0); GD_LST_L(L5, kr1); GD_LST_L(L6, vr0); GD_LST_L(L7, vr1);
            GD_SYNC_L();
            GD_GLD_L(kb, t + 5, kr0); GD_GLD_L(kb, t + 6, kr1); GD_GLD_L(vb, t + 4, vr0); GD_GLD_L(vb, t + 5, vr1);
            GD_ITER(false, L4, L6, X0, X1, Y0, Y1);
            GD_ITER(false, L5, L7, Y0, Y1, X0, X1);
            GD_LST_L(L0, kr0); GD_LST_L(L1, kr1); GD_LST_L(L2, vr0); GD_LST_L(L3, vr1);
            GD_SYNC_L();
        }
        GD_GLD_L(kb, t + 3, kr0); GD_GLD_L(vb, t + 2, vr0); GD_GLD_L(vb, t + 3, vr1);       // the final pair has no K(Tg)
        GD_ITER(false, L0, L2, X0, X1, Y0, Y1);
        GD_ITER(false, L1, L3, Y0, Y1, X0, X1);
        GD_LST_L(L4, kr0); GD_LST_L(L6, vr0); GD_LST_L(L7, vr1);
        GD_SYNC_L();
        GD_ITER(false, L4, L6, X0, X1, Y0, Y1);
        GD_ITER(true, L5, L7, Y0, Y1, X0, X1);
    } else {
        // one tile per barrier: iteration t reads K(t+1) from lK[(t+1)&1] and V(t) from lV[t&1]; K(t+2) / V(t+1) arrive in registers
        // and are stored before the barrier.  lK0 = L0, lK1 = L1, lV0 = L2, lV1 = L3.  Last pair peeled.
        int t = 0;
#pragma unroll 1
        for (; t + 2 < Tg; t += 2) {
            GD_GLD_L(kb, t + 2, kr0); GD_GLD_L(vb, t + 1, vr0);
            GD_ITER(false, L1, L2, X0, X1, Y0, Y1);
            GD_LST_L(L0, kr0); GD_LST_L(L3, vr0);
            GD_SYNC_L();
            GD_GLD_L(kb, t + 3, kr0); GD_GLD_L(vb, t + 2, vr0);
            GD_ITER(false, L0, L3, Y0, Y1, X0, X1);
            GD_LST_L(L1, kr0); GD_LST_L(L2, vr0);
            GD_SYNC_L();
        }
        GD_GLD_L(vb, t + 1, vr0);
        GD_ITER(false, L1, L2, X0, X1, Y0, Y1);
        GD_LST_L(L3, vr0);
        GD_SYNC_L();
        GD_ITER(true, L0, L3, Y0, Y1, X0, X1);
    }
#undef GD_ITER
#undef GD_GLD_L
#undef GD_LST_L
#undef GD_SYNC_L
#undef GD_GLD
#undef GD_LST
    MP_STAMP(2)
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

    MP_STAMP(3)
    bool write_out = true;
    if (SK && Tg != TU) {
        // This segment holds only part of its unit's keys.  Park the un-normalised (O, reference, row sum) of the 128 queries in this
        // workgroup's slot (write-through stores: the merging workgroup may sit on another XCD, whose L2 does not see this one's), then
        // take a ticket on the unit's counter; the workgroup that draws the LAST ticket folds all parts — its own included, read back,
        // in part order, so the result does not depend on who arrived last — and writes the unit's output.
        // Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): every storing wave drains its stores, workgroup barrier, one lane:
        // agent-scope release, drain, relaxed ticket; the last: agent-scope acquire, drain, workgroup barrier, then plain loads.
        const int w_first = (unit * TU) / a.sk_tpw, w_last = ((unit + 1) * TU - 1) / a.sk_tpw;
        // (s_nop after each asm store: a vector instruction must not overwrite the data registers of a > 8-byte store in the next two
        // issue slots; the compiler pads its own stores but does not look inside asm — seen as corrupted first two floats of a chunk)
        f32x4* const slot = a.sk_ws + (size_t)(2 * wg + (lin != lin_first ? 1 : 0)) * GD_SK_SLOT_F4 + (size_t)qb * 9 * 64 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 v = {o[j >> 2][4 * (j & 3)], o[j >> 2][4 * (j & 3) + 1], o[j >> 2][4 * (j & 3) + 2], o[j >> 2][4 * (j & 3) + 3]};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(slot + j * 64), "v"(v) : "memory");
        }
        {
            const f32x4 v = {m2, l_run, 0.f, 0.f};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(slot + 8 * 64), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (a.sk_mode == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const int ticket = __hip_atomic_fetch_add(a.sk_cnt + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == w_last - w_first;
            if (last) {
                if (a.sk_mode == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_store(a.sk_cnt + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            }
            sk_last = last;
        }
        __syncthreads();
        write_out = sk_last != 0;
        if (write_out && a.sk_mode != 2) {
#pragma unroll 1
            for (int w2 = w_first; w2 <= w_last; ++w2) {
                const f32x4* const sl = a.sk_ws + (size_t)(2 * w2 + (w2 * a.sk_tpw < unit * TU ? 1 : 0)) * GD_SK_SLOT_F4 + (size_t)qb * 9 * 64 + lane;
                f32x4 pv[9];
                // device-scope loads (served by L2, never by this CU's L1): one batch in flight, one wait
                asm volatile("global_load_dwordx4 %0, %9, off sc0 sc1\n\t"
                             "global_load_dwordx4 %1, %9, off offset:1024 sc0 sc1\n\t"
                             "global_load_dwordx4 %2, %9, off offset:2048 sc0 sc1\n\t"
                             "global_load_dwordx4 %3, %9, off offset:3072 sc0 sc1\n\t"
                             "global_load_dwordx4 %4, %10, off sc0 sc1\n\t"
                             "global_load_dwordx4 %5, %10, off offset:1024 sc0 sc1\n\t"
                             "global_load_dwordx4 %6, %10, off offset:2048 sc0 sc1\n\t"
                             "global_load_dwordx4 %7, %10, off offset:3072 sc0 sc1\n\t"
                             "global_load_dwordx4 %8, %11, off sc0 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]), "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]),
                               "=&v"(pv[8])
                             : "v"(sl), "v"(sl + 4 * 64), "v"(sl + 8 * 64)
                             : "memory");
                if (w2 == w_first) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[j >> 2][4 * (j & 3) + i] = pv[j][i];
                    m2 = pv[8][0]; l_run = pv[8][1];
                } else {
                    const float mk = pv[8][0], mn = fmaxf(m2, mk);
                    const float a0 = __builtin_amdgcn_exp2f(m2 - mn), a1 = __builtin_amdgcn_exp2f(mk - mn);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[j >> 2][4 * (j & 3) + i] = __builtin_fmaf(o[j >> 2][4 * (j & 3) + i], a0, pv[j][i] * a1);
                    l_run = __builtin_fmaf(l_run, a0, pv[8][1] * a1);
                    m2 = mn;
                }
            }
        }
    }

    MP_STAMP(4)
    if (write_out) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (qok) {
            T* __restrict__ op = (T*)sg.out + ooff + (size_t)qrow * rs;
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
            if (sg.lse && h == 0) sg.lse[(size_t)bh * Nq + qrow] = m2 * 0.6931471805599453f + __logf(l_tot);
        }
    }
    MP_STAMP(5)
#if GD_MP_DBG & 32
    ++dbg_seg;
#endif
    if (SK) {
        lin += Tg;
        if (lin < lin_end) __syncthreads();           // the next segment's first stores reuse the tile rings
    }
  } while (SK && lin < lin_end);
#undef MP_STAMP
}

// =================================================================================================================================
// k_attn_fwd_w64 — the same modulo-scheduled iteration for TWO 32-query blocks per wave, one wave per SIMD.
//
// Why.  Timing-only builds of k_attn_fwd_mp (tools/build_dbg.sh, profiles/r03_attn_fwd_dbg.md) showed where a 64^2 launch spends its
// time: 32 heads 138 us; without the K/V staging loads + LDS stores 111 us; without the LDS FRAGMENT READS 85 us; without both 60 us
// (2.3 PFLOP/s: the MFMA + softmax schedule itself runs at the vector-issue bound).  A 32-query wave reads the whole K and V tile from
// LDS as MFMA A operands for 16 MFMAs: 16 KB per wave and tile, 128 KB per CU and tile round = half the LDS bandwidth at the full MFMA
// rate before the 32 KB of ds_write_b128 staging — the LDS queue, not the matrix or vector pipe, paces the loop.  Here a wave owns 64
// queries (two 32-row blocks A / B that share every K / V fragment: LDS fragment traffic per MFMA halves, a fragment is live for 8
// MFMA gaps instead of 4 so twice the LDS latency is covered) and the K / V tiles arrive by direct-to-LDS loads (buffer_load ... lds:
// no staging registers, no ds_write) into two 4-tile sets, one tile pair ahead.  ~360 VGPRs: one
// wave per SIMD, one 256-query workgroup per CU.  Per tile: 32 gaps of 1 MFMA + 2 probabilities:
//     gaps  1- 8   O_A, O_B += V(t-1)[keys 32..63] P(t-1)     | exp2 of S'_A(t)[keys  0..31]   | read K(t+1)[rows  0..31] fragments
//     gaps  9-16   S'_A, S'_B(t+1)[keys  0..31] = K(t+1) Q'^T | exp2 of S'_B(t)[keys  0..31]   | read V(t)[keys  0..31] (transposed)
//     gaps 17-24   O_A, O_B += V(t)[keys  0..31] P(t)         | exp2 of S'_A(t)[keys 32..63]   | read K(t+1)[rows 32..63]
//     gaps 25-32   S'_A, S'_B(t+1)[keys 32..63]               | exp2 of S'_B(t)[keys 32..63]   | read V(t)[keys 32..63]
// one direct-to-LDS piece (1 KB) per 8 gaps.  Segments, token-major rows, fused query warp, query row lists as in k_attn_fwd_mp;
// units are 256 queries.  What differs:
//   * softmax reference: PRE (queries carry scale*log2 e): the first key tile's row maximum, fixed; a segment whose half-step sums ever
//     exceed 2^60 (fp16: 2^14, the range of its probabilities) is repeated with exact row maxima (see the attempt loop).  !PRE (exact scale: the optimisation pass): raw scores in
//     the tiles and k_attn_fwd_mp's in-loop rescue, which only touches O, l and the reference (w64_rescue);
//   * SK: the linear range of k_attn_fwd_mp, or — launches of at most 128 units — every unit in 2-4 parts, one per workgroup
//     (FwdArgs::sk_parts); hand-off: the holder of a unit's first part merges without storing its own (see the epilogue);
//   * the output goes through LDS so that the global stores are whole rows.
// =================================================================================================================================

#define W64_SUM_LIMIT 1.152921504606847e18f       // 2^60

// (w64_dma: attn_common.hpp)

// Score MFMAs of the 64-query kernel.  With 512 registers per wave hipcc selects the accumulator-register form for every MFMA builtin
// (D and C in a[...]); scores are consumed by v_exp_f32, which cannot read a[...]: 16 v_accvgpr_read per score tile and 16
// v_accvgpr_write per initial accumulator (measured: +352 vector instructions per 64 MFMAs).  These two keep D / C in v[...] (the
// instruction's other encoding) with the K and query fragments in a[...] (ds_read writes a[...] directly); the O products stay on the builtin
// (accumulators in a[...], never touched by vector instructions in the loop).  Hazards: nothing reads D for >= 16 MFMA gaps; chains on
// one D are two gaps apart; C of a chain head (negmu) is only written on the cold path, which pads its own wait states.
template <typename T> struct score_mfma;
template <> struct score_mfma<bf16_t> {
    static __device__ __forceinline__ void head(f32x16& d, const bf16x8 a, const bf16x8 b, const f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
    }
    // exact-scale variant: no bias tile (C = the inline constant 0), and the query fragments live in v[...] — the registers the bias
    // tiles occupy in the pre-scaled variant; with an "a" operand the compiler kept them in v[...] anyway and copied them into a[...]
    // in front of every use (96 v_accvgpr_write per 128 MFMAs, each one the hazard described above)
    static __device__ __forceinline__ void head_zero(f32x16& d, const bf16x8 a, const bf16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(d) : "a"(a), "v"(b));
    }
    static __device__ __forceinline__ void acc_vb(f32x16& d, const bf16x8 a, const bf16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
    }
    static __device__ __forceinline__ void acc(f32x16& d, const bf16x8 a, const bf16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
    }
    static __device__ __forceinline__ void acc_pad(f32x16& d, const bf16x8 a, const bf16x8 b) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
    }
    // padded forms of the loop's MFMAs, for the peeled iterations behind the loop (see acc_pad)
    static __device__ __forceinline__ void head_pad(f32x16& d, const bf16x8 a, const bf16x8 b, const f32x16& c) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
    }
    static __device__ __forceinline__ void head_zero_pad(f32x16& d, const bf16x8 a, const bf16x8 b) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(d) : "a"(a), "v"(b));
    }
    static __device__ __forceinline__ void acc_vb_pad(f32x16& d, const bf16x8 a, const bf16x8 b) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
    }
};
// (acc_pad: for the code outside the loop, where the compiler may have just copied a fragment into a[...] with v_accvgpr_write — an
// MFMA must not read such a register in the next wait states, and the hazard recogniser does not pad in front of asm.  Seen as NaNs in
// query block B only: its fragment was written immediately before the instruction.  In the loop the fragments are resident in a[...]
// (the instruction-mix check of tools/loopstat shows no v_accvgpr traffic) and K fragments arrive by ds_read, which s_waitcnt covers.)
template <> struct score_mfma<f16_t> {
    static __device__ __forceinline__ void head(f32x16& d, const f16x8 a, const f16x8 b, const f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
    }
    // exact-scale variant: no bias tile (C = the inline constant 0), and the query fragments live in v[...] — the registers the bias
    // tiles occupy in the pre-scaled variant; with an "a" operand the compiler kept them in v[...] anyway and copied them into a[...]
    // in front of every use (96 v_accvgpr_write per 128 MFMAs, each one the hazard described above)
    static __device__ __forceinline__ void head_zero(f32x16& d, const f16x8 a, const f16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(d) : "a"(a), "v"(b));
    }
    static __device__ __forceinline__ void acc_vb(f32x16& d, const f16x8 a, const f16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
    }
    static __device__ __forceinline__ void acc(f32x16& d, const f16x8 a, const f16x8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
    }
    static __device__ __forceinline__ void acc_pad(f32x16& d, const f16x8 a, const f16x8 b) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
    }
    // padded forms of the loop's MFMAs, for the peeled iterations behind the loop (see acc_pad)
    static __device__ __forceinline__ void head_pad(f32x16& d, const f16x8 a, const f16x8 b, const f32x16& c) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
    }
    static __device__ __forceinline__ void head_zero_pad(f32x16& d, const f16x8 a, const f16x8 b) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(d) : "a"(a), "v"(b));
    }
    static __device__ __forceinline__ void acc_vb_pad(f32x16& d, const f16x8 a, const f16x8 b) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
    }
};

// Two scores -> two probabilities.  Plain fp32 VALU on purpose: the packed forms (v_pk_fma_f32 for the exact-scale variant's
// scale-and-shift, v_pk_add_f32 for the row sums) halve the issue slots but run on the pipe the MFMAs use — measured with them the
// 15-head exact-scale launch went 69 -> 86 us and the 20-head pre-scaled one 98 -> 133 us (profiles/r03_packed_fp32.log).
#define W64_EG2(SRC, i, DST, j, QI)                                                                              \
    {                                                                                                            \
        const float p0_ = __builtin_amdgcn_exp2f(PRE ? SRC[i] : __builtin_fmaf(SRC[i], c, -m2[QI]));             \
        const float p1_ = __builtin_amdgcn_exp2f(PRE ? SRC[(i) + 1] : __builtin_fmaf(SRC[(i) + 1], c, -m2[QI])); \
        DST[(j) >> 1] = pack2<T>(p0_, p1_);                                                                      \
        if (!LSUM) { ps0 += p0_; ps1 += p1_; }                                                                   \
    }
// Cold path of the exact-scale variant (!PRE: the optimisation pass): a half step's probability sum left [0, 2^14] — raise the reference
// by the half's maximum exponent, rescale O and l, recompute the half's probabilities, exactly like softmax_rescue of k_attn_fwd_mp.
// With raw scores in the tiles (p = exp2(c S - mu), mu a scalar) nothing but O, l and mu changes: no score or bias tile is touched, which
// is what made the in-place rescue unaffordable for the pre-scaled variant (see the kernel's header).  It also keeps the property the
// parity tests of the optimisation pass were calibrated on: after a rescue the reference sits exactly on a row maximum, so a DOMINANT
// probability is exactly 1.0 in 16 bits.
template <typename T>
__device__ __forceinline__ void w64_rescue(const f32x16& H, f32x16 (&o)[2], float& m2, float& l_run, const float c, u32x4& pf0, u32x4& pf1,
                                           float& ps) {
    float mx = H[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, H[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fmaxf(__builtin_fmaf(mx, c, -m2), 0.f);
    float alpha = __builtin_amdgcn_exp2f(-mx);
    l_run *= alpha;
    m2 += mx;
    GD_TILE_OP("v_mul_f32", o[0], alpha)
    GD_TILE_OP("v_mul_f32", o[1], alpha)
    ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(H[i], c, -m2));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(H[i + 1], c, -m2));
        const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(H[8 + i], c, -m2));
        const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(H[8 + i + 1], c, -m2));
        pf0[i >> 1] = pack2<T>(p0, p1);
        pf1[i >> 1] = pack2<T>(p2, p3);
        ps += (p0 + p1) + (p2 + p3);
    }
}

// end of a half step: row sums; pre-scaled variant: the running maximum of the half-step sums (checked once per segment, see
// k_attn_fwd_w64); exact-scale variant: the in-loop rescue above
#define W64_CHECK(H, QI, PF0, PF1)                                                                               \
    if (!LSUM) {                                                                                                 \
        float ps = ps0 + ps1;                                                                                    \
        if (!PRE) {                                                                                              \
            if (!(GD_MP_DBG & 4) && __builtin_expect(__builtin_amdgcn_ballot_w64(!(ps <= P_SUM_LIMIT)) != 0, 0)) \
                w64_rescue<T>(H, o[QI], m2[QI], l_run[QI], c, PF0, PF1, ps);                                     \
        } else {                                                                                                 \
            chk = fmaxf(chk, ps);                                                                                \
        }                                                                                                        \
        l_run[QI] += ps;                                                                                         \
        ps0 = 0.f; ps1 = 0.f;                                                                                    \
    }

// what one iteration stages for a later pair: two pieces of a K tile and two of a V tile (DMA = false: nothing)
struct W64Stage {
    __amdgpu_buffer_rsrc_t kb, vb;
    uint32_t voff[2];
    int ksoff, vsoff;          // byte offsets of the K / V tile inside the segment's rows
    char* kdst; char* vdst;    // this wave's first piece of the destination tiles
};

// LSUM (pre-scaled variant only): the row sums come from the matrix pipe — l^T += 1 P^T, one MFMA per 16-key step and query block whose A
// operand is a fragment of ones and whose B operand is the SAME 16-bit probability fragment the O products consume.  Two things follow:
//   * the two v_add_f32 per pair of probabilities leave the vector pipe, which paces this loop (MFMA issue 8 + 2 x v_exp 16 + 2 x v_add 8
//     + pack 4.5 = 36.5-40.5 cycles per 32-cycle MFMA gap, profiles/r03_ub_gap.log): 28.5 per gap, at the price of 8 more MFMAs per 64;
//   * numerator and denominator are sums over the same ROUNDED probabilities: out = sum p~ v / sum p~, so a dominant probability gives
//     out = v exactly whatever the softmax reference is — the property the exact-scale rescue variant was kept for in the optimisation
//     pass (DESIGN 4a'), now without the rescue.
// score MFMA of the iteration: PAD = the padded asm forms (peeled iterations: the compiler may copy a fragment into a[...] right in front of
// a use there — seen with the row-sum variant's register pressure — and does not pad in front of asm)
template <typename T, bool PRE, bool PAD>
__device__ __forceinline__ void w64_head(f32x16& d, const typename elem_traits<T>::vec8 a, const typename elem_traits<T>::vec8 b, const f32x16& c) {
    if (PRE) { if (PAD) score_mfma<T>::head_pad(d, a, b, c); else score_mfma<T>::head(d, a, b, c); }
    else { if (PAD) score_mfma<T>::head_zero_pad(d, a, b); else score_mfma<T>::head_zero(d, a, b); }
}
template <typename T, bool PRE, bool PAD>
__device__ __forceinline__ void w64_acc(f32x16& d, const typename elem_traits<T>::vec8 a, const typename elem_traits<T>::vec8 b) {
    if (PRE) { if (PAD) score_mfma<T>::acc_pad(d, a, b); else score_mfma<T>::acc(d, a, b); }
    else { if (PAD) score_mfma<T>::acc_vb_pad(d, a, b); else score_mfma<T>::acc_vb(d, a, b); }
}
template <typename T, bool LAST, bool PRE, bool DMA, bool LSUM, bool PAD = false>
__device__ __forceinline__ void w64_iter(const char* lk, const char* lv, const FragOffs& fo, const typename elem_traits<T>::vec8 (&qf)[2][4],
                                         f32x16 (&X0)[2], f32x16 (&X1)[2], f32x16 (&Y0)[2], f32x16 (&Y1)[2], f32x16 (&negmu)[2],
                                         f32x16 (&o)[2][2], float (&m2)[2], float (&l_run)[2], const float c, u32x4 (&pb0)[2],
                                         u32x4 (&pb1)[2], typename elem_traits<T>::vec8 (&vc)[4], const W64Stage& sg, float& chk,
                                         f32x16 (&lacc)[2], const typename elem_traits<T>::vec8& ones) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    u32x4 pa0[2], pa1[2];
    V8 kf[4], va[4];
    float ps0 = 0.f, ps1 = 0.f;
    // ---- gaps 1-8: second half of tile t-1 into O | probabilities of block A, keys 0..31 | K(t+1) rows 0..31 ----
    if (DMA && !(GD_MP_DBG & 1)) w64_dma(sg.kb, sg.kdst, sg.voff[0], sg.ksoff);
    if (!LAST) kf[0] = ((GD_MP_DBG & 8) ? qf[0][0] : rd_row<T>(lk, fo, 0, 0));
    // (row-sum MFMAs FIRST in their group: an MFMA that is the LAST reader of a probability fragment, directly followed by the v_exp_f32 that
    //  reuses the fragment's register, read the new value — wrong sums for query block B, bit-exact for block A; no such pair is left in any
    //  instantiation: tests/test_cabi.py checks the assembly)
    if (LSUM) { lacc[0] = TR::mfma32(ones, as_frag<T>(pb0[0]), lacc[0]); GD_SB(); lacc[1] = TR::mfma32(ones, as_frag<T>(pb0[1]), lacc[1]); GD_SB();
                 lacc[0] = TR::mfma32(ones, as_frag<T>(pb1[0]), lacc[0]); GD_SB(); lacc[1] = TR::mfma32(ones, as_frag<T>(pb1[1]), lacc[1]); GD_SB(); }
    o[0][0] = TR::mfma32(vc[0], as_frag<T>(pb0[0]), o[0][0]); W64_EG2(X0[0], 0, pa0[0], 0, 0); GD_SB();
    o[1][0] = TR::mfma32(vc[0], as_frag<T>(pb0[1]), o[1][0]); W64_EG2(X0[0], 2, pa0[0], 2, 0); GD_SB();
    if (!LAST) kf[1] = ((GD_MP_DBG & 8) ? qf[0][1] : rd_row<T>(lk, fo, 0, 1));
    o[0][1] = TR::mfma32(vc[1], as_frag<T>(pb0[0]), o[0][1]); W64_EG2(X0[0], 4, pa0[0], 4, 0); GD_SB();
    o[1][1] = TR::mfma32(vc[1], as_frag<T>(pb0[1]), o[1][1]); W64_EG2(X0[0], 6, pa0[0], 6, 0); GD_SB();
    if (!LAST) kf[2] = ((GD_MP_DBG & 8) ? qf[0][2] : rd_row<T>(lk, fo, 0, 2));
    o[0][0] = TR::mfma32(vc[2], as_frag<T>(pb1[0]), o[0][0]); W64_EG2(X0[0], 8, pa1[0], 0, 0); GD_SB();
    o[1][0] = TR::mfma32(vc[2], as_frag<T>(pb1[1]), o[1][0]); W64_EG2(X0[0], 10, pa1[0], 2, 0); GD_SB();
    if (!LAST) kf[3] = ((GD_MP_DBG & 8) ? qf[0][3] : rd_row<T>(lk, fo, 0, 3));
    o[0][1] = TR::mfma32(vc[3], as_frag<T>(pb1[0]), o[0][1]); W64_EG2(X0[0], 12, pa1[0], 4, 0); GD_SB();
    o[1][1] = TR::mfma32(vc[3], as_frag<T>(pb1[1]), o[1][1]); W64_EG2(X0[0], 14, pa1[0], 6, 0); GD_SB();
    W64_CHECK(X0[0], 0, pa0[0], pa1[0])
    GD_SB();
    // ---- gaps 9-16: scores of tile t+1, keys 0..31 | probabilities of block B, keys 0..31 | V(t) keys 0..31 ----
    if (DMA && !(GD_MP_DBG & 1)) w64_dma(sg.kb, sg.kdst + 1024, sg.voff[1], sg.ksoff);
    va[0] = ((GD_MP_DBG & 8) ? qf[1][0] : rd_tr<T>(lv, fo, 0, 0));
    if (!LAST) w64_head<T, PRE, PAD>(Y0[0], kf[0], qf[0][0], negmu[0]);
    W64_EG2(X0[1], 0, pa0[1], 0, 1); GD_SB();
    if (!LAST) w64_head<T, PRE, PAD>(Y0[1], kf[0], qf[1][0], negmu[1]);
    W64_EG2(X0[1], 2, pa0[1], 2, 1); GD_SB();
    va[1] = ((GD_MP_DBG & 8) ? qf[1][0] : rd_tr<T>(lv, fo, 1, 0));
    if (!LAST) w64_acc<T, PRE, PAD>(Y0[0], kf[1], qf[0][1]);
    W64_EG2(X0[1], 4, pa0[1], 4, 1); GD_SB();
    if (!LAST) w64_acc<T, PRE, PAD>(Y0[1], kf[1], qf[1][1]);
    W64_EG2(X0[1], 6, pa0[1], 6, 1); GD_SB();
    va[2] = ((GD_MP_DBG & 8) ? qf[1][1] : rd_tr<T>(lv, fo, 0, 1));
    if (!LAST) w64_acc<T, PRE, PAD>(Y0[0], kf[2], qf[0][2]);
    W64_EG2(X0[1], 8, pa1[1], 0, 1); GD_SB();
    if (!LAST) w64_acc<T, PRE, PAD>(Y0[1], kf[2], qf[1][2]);
    W64_EG2(X0[1], 10, pa1[1], 2, 1); GD_SB();
    va[3] = ((GD_MP_DBG & 8) ? qf[1][1] : rd_tr<T>(lv, fo, 1, 1));
    if (!LAST) w64_acc<T, PRE, PAD>(Y0[0], kf[3], qf[0][3]);
    W64_EG2(X0[1], 12, pa1[1], 4, 1); GD_SB();
    if (!LAST) w64_acc<T, PRE, PAD>(Y0[1], kf[3], qf[1][3]);
    W64_EG2(X0[1], 14, pa1[1], 6, 1); GD_SB();
    W64_CHECK(X0[1], 1, pa0[1], pa1[1])
    GD_SB();
    // ---- gaps 17-24: first half of tile t into O | probabilities of block A, keys 32..63 | K(t+1) rows 32..63 ----
    if (DMA && !(GD_MP_DBG & 1)) w64_dma(sg.vb, sg.vdst, sg.voff[0], sg.vsoff);
    if (!LAST) kf[0] = ((GD_MP_DBG & 8) ? qf[0][0] : rd_row<T>(lk, fo, 1, 0));
    if (LSUM) { lacc[0] = TR::mfma32(ones, as_frag<T>(pa0[0]), lacc[0]); GD_SB(); lacc[1] = TR::mfma32(ones, as_frag<T>(pa0[1]), lacc[1]); GD_SB();
                 lacc[0] = TR::mfma32(ones, as_frag<T>(pa1[0]), lacc[0]); GD_SB(); lacc[1] = TR::mfma32(ones, as_frag<T>(pa1[1]), lacc[1]); GD_SB(); }
    o[0][0] = TR::mfma32(va[0], as_frag<T>(pa0[0]), o[0][0]); W64_EG2(X1[0], 0, pb0[0], 0, 0); GD_SB();
    o[1][0] = TR::mfma32(va[0], as_frag<T>(pa0[1]), o[1][0]); W64_EG2(X1[0], 2, pb0[0], 2, 0); GD_SB();
    if (!LAST) kf[1] = ((GD_MP_DBG & 8) ? qf[0][1] : rd_row<T>(lk, fo, 1, 1));
    o[0][1] = TR::mfma32(va[1], as_frag<T>(pa0[0]), o[0][1]); W64_EG2(X1[0], 4, pb0[0], 4, 0); GD_SB();
    o[1][1] = TR::mfma32(va[1], as_frag<T>(pa0[1]), o[1][1]); W64_EG2(X1[0], 6, pb0[0], 6, 0); GD_SB();
    if (!LAST) kf[2] = ((GD_MP_DBG & 8) ? qf[0][2] : rd_row<T>(lk, fo, 1, 2));
    o[0][0] = TR::mfma32(va[2], as_frag<T>(pa1[0]), o[0][0]); W64_EG2(X1[0], 8, pb1[0], 0, 0); GD_SB();
    o[1][0] = TR::mfma32(va[2], as_frag<T>(pa1[1]), o[1][0]); W64_EG2(X1[0], 10, pb1[0], 2, 0); GD_SB();
    if (!LAST) kf[3] = ((GD_MP_DBG & 8) ? qf[0][3] : rd_row<T>(lk, fo, 1, 3));
    o[0][1] = TR::mfma32(va[3], as_frag<T>(pa1[0]), o[0][1]); W64_EG2(X1[0], 12, pb1[0], 4, 0); GD_SB();
    o[1][1] = TR::mfma32(va[3], as_frag<T>(pa1[1]), o[1][1]); W64_EG2(X1[0], 14, pb1[0], 6, 0); GD_SB();
    W64_CHECK(X1[0], 0, pb0[0], pb1[0])
    GD_SB();
    // ---- gaps 25-32: scores of tile t+1, keys 32..63 | probabilities of block B, keys 32..63 | V(t) keys 32..63 ----
    if (DMA && !(GD_MP_DBG & 1)) w64_dma(sg.vb, sg.vdst + 1024, sg.voff[1], sg.vsoff);
    vc[0] = ((GD_MP_DBG & 8) ? qf[1][2] : rd_tr<T>(lv, fo, 0, 2));
    if (!LAST) w64_head<T, PRE, PAD>(Y1[0], kf[0], qf[0][0], negmu[0]);
    W64_EG2(X1[1], 0, pb0[1], 0, 1); GD_SB();
    if (!LAST) w64_head<T, PRE, PAD>(Y1[1], kf[0], qf[1][0], negmu[1]);
    W64_EG2(X1[1], 2, pb0[1], 2, 1); GD_SB();
    vc[1] = ((GD_MP_DBG & 8) ? qf[1][2] : rd_tr<T>(lv, fo, 1, 2));
    if (!LAST) w64_acc<T, PRE, PAD>(Y1[0], kf[1], qf[0][1]);
    W64_EG2(X1[1], 4, pb0[1], 4, 1); GD_SB();
    if (!LAST) w64_acc<T, PRE, PAD>(Y1[1], kf[1], qf[1][1]);
    W64_EG2(X1[1], 6, pb0[1], 6, 1); GD_SB();
    vc[2] = ((GD_MP_DBG & 8) ? qf[1][3] : rd_tr<T>(lv, fo, 0, 3));
    if (!LAST) w64_acc<T, PRE, PAD>(Y1[0], kf[2], qf[0][2]);
    W64_EG2(X1[1], 8, pb1[1], 0, 1); GD_SB();
    if (!LAST) w64_acc<T, PRE, PAD>(Y1[1], kf[2], qf[1][2]);
    W64_EG2(X1[1], 10, pb1[1], 2, 1); GD_SB();
    vc[3] = ((GD_MP_DBG & 8) ? qf[1][3] : rd_tr<T>(lv, fo, 1, 3));
    if (!LAST) w64_acc<T, PRE, PAD>(Y1[0], kf[3], qf[0][3]);
    W64_EG2(X1[1], 12, pb1[1], 4, 1); GD_SB();
    if (!LAST) w64_acc<T, PRE, PAD>(Y1[1], kf[3], qf[1][3]);
    W64_EG2(X1[1], 14, pb1[1], 6, 1); GD_SB();
    W64_CHECK(X1[1], 1, pb0[1], pb1[1])
}

template <typename T, bool PRE, bool SK, bool LSUM = false>
__global__ void __launch_bounds__(256, 1)
k_attn_fwd_w64(const FwdArgs a) {
    static_assert(PRE || !LSUM, "row sums on the matrix pipe: pre-scaled variant only (the rescue variant needs the half-step sums on the vector pipe)");
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    // set = {Ka, Kb, Va, Vb} of one tile pair.  Two separate arrays: the compiler can then prove that the direct-to-LDS stores into one
    // set do not alias the fragment reads of the other (with one array and a runtime set index it waits vmcnt(0) before every ds_read)
    __shared__ __attribute__((aligned(16))) char ldsA[4][ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) char ldsB[4][ATT_TILE_BYTES];
    __shared__ int sk_last, w_abort;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const int TU = a.M / ATT_BN;
    // debug bit 32 (timing builds, tools/w64_phases.py): 100 MHz timestamps of the phases of every segment, 1 MB behind the part slots of the workspace
#if GD_MP_DBG & 32
    int dbg_seg = 0;
#define W64_STAMP(E) { if (tid == 0 && a.sk_ws) ((unsigned long long*)((char*)a.sk_ws + GD_SK_SLOT_BYTES + (1u << 20)))[(wg * 4 + dbg_seg) * 8 + (E)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define W64_STAMP(E) {}
#endif
    // SK, two ways of dealing the key tiles out: one linear range (a workgroup walks one or more segments, see k_attn_fwd_mp), or — few
    // units — every unit in P = sk_parts equal runs, one per workgroup (workgroup = unit * P + part: one segment, one prologue)
    const int P = SK ? a.sk_parts : 0;
    int lin = SK ? (P ? 0 : wg * a.sk_tpw) : 0;
    const int lin_end = SK ? (P ? 1 : (lin + a.sk_tpw < a.sk_total ? lin + a.sk_tpw : a.sk_total)) : 0;
    const int lin_first = lin;
    const FragOffs fo = make_frag_offs(lane);
#pragma unroll 1
  do {
    W64_STAMP(0)
    int unit = wg, t_first = 0, Tg = TU;
    if (SK && P) {
        // parts 1 .. P-1 hold sk_tpw tiles each, part 0 the rest (>= sk_tpw): the holder of the unit's FIRST part finishes last, finds
        // the other parts in place and merges without storing its own (see the hand-off below)
        // (units below sk_unit0 are not cut: one workgroup, the whole key range, no hand-off)
        if (wg >= a.sk_unit0) {
            const int cw = wg - a.sk_unit0;
            unit = a.sk_unit0 + cw / P;
            const int part = cw - (unit - a.sk_unit0) * P, first_len = TU - (P - 1) * a.sk_tpw;
            t_first = part ? first_len + (part - 1) * a.sk_tpw : 0;
            Tg = part ? a.sk_tpw : first_len;
        }
    } else if (SK) {
        unit = lin / TU;
        t_first = lin - unit * TU;
        Tg = TU - t_first < lin_end - lin ? TU - t_first : lin_end - lin;
    }
    int sidx = 0, bh, tile;
    const bool compact = a.cseg >= 0 && unit >= a.units_full;      // the segment with a query row list (see k_attn_fwd_mp)
    if (compact) {
        compact_unit(a, unit, sidx, bh, tile);
    } else {
        const int gbh = unit / a.tiles;
        tile = unit - gbh * a.tiles;
        if (a.n_order > 0) {
            const int code = a.order[gbh];
            sidx = code >> 12;
            bh = code & 4095;
        } else {
#pragma unroll
            for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
                if (i < a.nseg - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
            bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
        }
    }
    const gd_attn_seg_t sg = a.seg[sidx];
    const int N = a.N, M = a.M;
    const int Nq = compact ? sg.q_rows_len : N;
    const int rs = sg.heads > 0 ? sg.heads * ATT_D : ATT_D;
    size_t qoff, koff, ooff;
    if (sg.heads > 0) {
        const int b = bh / sg.heads, hh = bh - b * sg.heads;
        qoff = (size_t)b * N * rs + (size_t)hh * ATT_D;
        koff = (size_t)b * M * rs + (size_t)hh * ATT_D;
        ooff = (size_t)b * Nq * rs + (size_t)hh * ATT_D;
    } else {
        qoff = (size_t)bh * N * ATT_D;
        koff = (size_t)bh * M * ATT_D;
        ooff = (size_t)bh * Nq * ATT_D;
    }
    const T* __restrict__ qp = (const T*)sg.q + qoff;
    W64Stage st;
    st.kb = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)sg.k + koff + (size_t)t_first * ATT_BN * rs), 0, 0x7FFFFFFF, 0x00020000);
    st.vb = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)sg.v + koff + (size_t)t_first * ATT_BN * rs), 0, 0x7FFFFFFF, 0x00020000);
    const int tB = ATT_BN * rs * (int)sizeof(T);                            // bytes from one key tile to the next
    // this wave's two 1-KB pieces of a tile: rows 16 w .. 16 w + 15; the lane that fills 16-byte slot (row, c') of the swizzled image
    // fetches global chunk c' ^ g(row) of that row (the image's XOR moves to the source address)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wave + j) * 8 + (lane >> 3);
        const int x = (row >> 1) & 7, g = (x & 2) | ((x & 1) << 2) | ((x >> 2) & 1);
        st.voff[j] = (uint32_t)(((unsigned)row * (unsigned)rs + (unsigned)(((lane & 7) ^ g) * 8)) * (unsigned)sizeof(T));
    }
    // pair p (tiles 2p, 2p+1) reads Ka = K(2p+1), Kb = K(2p+2), Va = V(2p), Vb = V(2p+1); even pairs live in set A, odd pairs in set B;
    // K(0) borrows set B's Ka until the first pair ends
#define W64_ISSUE_TILE(RS, TILE_IDX, DSTTILE)                                                       \
    {                                                                                               \
        w64_dma(RS, (DSTTILE) + wave * 2048, st.voff[0], (TILE_IDX) * tB);                          \
        w64_dma(RS, (DSTTILE) + wave * 2048 + 1024, st.voff[1], (TILE_IDX) * tB);                   \
    }
    // K(0) on its way before the query loads (with warp tables: dependent round trips, slow in the launch's opening burst) start; the
    // other four tiles of the pipeline's fill behind them: the reference value (first tile's scores) needs K(0) and the queries only,
    // and memory operations complete in order — `vmcnt(8)` below = everything but the last eight pieces has landed
    W64_ISSUE_TILE(st.kb, 0, ldsB[0])
    const int qrowA = tile * 256 + wave * 64 + (lane & 31), qrowB = qrowA + 32;      // rows of the output (= of the row list, if any)
    const int nq_ok = compact ? sg.q_rows_n[0] : N;
    const int pix[2] = {compact ? sg.q_rows[qrowA < Nq ? qrowA : Nq - 1] : (qrowA < N ? qrowA : N - 1),
                        compact ? sg.q_rows[qrowB < Nq ? qrowB : Nq - 1] : (qrowB < N ? qrowB : N - 1)};   // rows of q they are computed from
    V8 qf[2][4];
    if (sg.warp_idx) {               // warped, blended queries of both blocks built together (composite_chunks2: fewer dependent round trips)
        const int coff[4] = {8 * h, 16 + 8 * h, 32 + 8 * h, 48 + 8 * h};
        composite_chunks2<T, 4>(qp, (size_t)rs, coff, sg.warp_idx, sg.warp_w, sg.warp_m, pix, sg.warp_K, qf);
    } else {
        load_q_frags<T>(sg, qp, rs, pix[0], h, qf[0]);
        load_q_frags<T>(sg, qp, rs, pix[1], h, qf[1]);
    }
    W64_ISSUE_TILE(st.kb, 1, ldsA[0]) W64_ISSUE_TILE(st.kb, 2, ldsA[1]) W64_ISSUE_TILE(st.vb, 0, ldsA[2]) W64_ISSUE_TILE(st.vb, 1, ldsA[3])
    if (PRE && !a.q_prescaled) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[b][s4][j] = TR::from_f32(TR::to_f32(qf[b][s4][j]) * a.c);
    }
    const float c = (PRE || a.q_prescaled) ? 1.0f : a.c;

    f32x16 o[2][2], X0[2], X1[2], Y0[2], Y1[2], negmu[2];
    f32x16 lacc[2];                   // LSUM: row sums as MFMA accumulators (every row of the 32 x 32 tile holds the same sum)
    V8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = TR::from_f32(1.0f);
    float m2[2], l_run[2];
    u32x4 pb0[2], pb1[2];
    // wait states around the asm score MFMAs outside the loop.  The X tiles are OPERANDS of the pad: an asm with only a memory clobber
    // does not stop the compiler from moving the vector instructions that write / read X across it (seen: row maxima read before the
    // MFMAs had landed — a timing-dependent reference value, outputs one ulp apart run to run)
#define W64_PAD(NOPS) asm volatile(NOPS : "+v"(X0[0]), "+v"(X0[1]), "+v"(X1[0]), "+v"(X1[1]));
    // Reference value of the softmax.  There is no per-step maximum and — unlike k_attn_fwd_mp — no in-loop rescue (its in-place updates of
    // seven tiles per query block, inlined at 16 check points, made the register allocator park the bias tiles in a[...] and copy them
    // back for every score MFMA).  bf16 probabilities and f32 sums are exact in RELATIVE terms for any reference as long as nothing
    // overflows, so: attempt 0 takes the row maximum of the segment's first tile and only tracks the largest half-step probability sum;
    // if that ever exceeded 2^60 (scores > 40 nats above the first tile's maximum: never seen outside the adversarial tests) the whole
    // workgroup repeats the segment with the EXACT row maxima from a scores-only pre-pass.  Exact for every input either way.
#pragma unroll 1
  for (int attempt = (GD_MP_DBG & 16) ? 1 : 0;; ++attempt) {          // debug bit 4: always take the exact row maxima
    if (tid == 0) w_abort = 0;
    float mxe[2] = {-INFINITY, -INFINITY};
    if (attempt) {
#pragma unroll 1
        for (int t = 0; t < Tg; ++t) {
            W64_ISSUE_TILE(st.kb, t, ldsB[0])
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) { X0[b][i] = 0.f; X1[b][i] = 0.f; }
            W64_PAD("s_nop 7")
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const V8 f0 = rd_row<T>(ldsB[0], fo, 0, s4), f1 = rd_row<T>(ldsB[0], fo, 1, s4);
                score_mfma<T>::acc_pad(X0[0], f0, qf[0][s4]); score_mfma<T>::acc_pad(X0[1], f0, qf[1][s4]);
                score_mfma<T>::acc_pad(X1[0], f1, qf[0][s4]); score_mfma<T>::acc_pad(X1[1], f1, qf[1][s4]);
            }
            W64_PAD("s_nop 15\n\ts_nop 15")
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) mxe[b] = fmaxf(mxe[b], fmaxf(X0[b][i], X1[b][i]));
            __syncthreads();
        }
    }
    if (attempt) {                    // (attempt 0: issued before the query loads)
        W64_ISSUE_TILE(st.kb, 0, ldsB[0])
        W64_ISSUE_TILE(st.kb, 1, ldsA[0]) W64_ISSUE_TILE(st.kb, 2, ldsA[1]) W64_ISSUE_TILE(st.vb, 0, ldsA[2]) W64_ISSUE_TILE(st.vb, 1, ldsA[3])
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // K(0) (and, by the compiler's own count, the queries where they are used)
    __syncthreads();

#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[b][0][i] = 0.f; o[b][1][i] = 0.f; X0[b][i] = 0.f; X1[b][i] = 0.f; Y0[b][i] = 0.f; Y1[b][i] = 0.f; lacc[b][i] = 0.f; }
    }
    W64_PAD("s_nop 7")
    {
        const char* k0 = ldsB[0];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const V8 f0 = rd_row<T>(k0, fo, 0, s4), f1 = rd_row<T>(k0, fo, 1, s4);
            score_mfma<T>::acc_pad(X0[0], f0, qf[0][s4]); score_mfma<T>::acc_pad(X0[1], f0, qf[1][s4]);
            score_mfma<T>::acc_pad(X1[0], f1, qf[0][s4]); score_mfma<T>::acc_pad(X1[1], f1, qf[1][s4]);
        }
    }
    W64_PAD("s_nop 15\n\ts_nop 15")                          // the asm MFMAs above have landed before vector instructions read X
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float mx = mxe[b];
        if (!attempt) {               // the row maximum of the segment's first tile (log2 domain after * c)
            mx = X0[b][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, X0[b][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, X1[b][i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
        m2[b] = mx;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PRE) { X0[b][i] -= mx; X1[b][i] -= mx; }
            negmu[b][i] = PRE ? -mx : 0.f;
        }
        l_run[b] = 0.f;
        pb0[b] = u32x4{0u, 0u, 0u, 0u}; pb1[b] = u32x4{0u, 0u, 0u, 0u};
    }
    float chk = 0.f;
    V8 vc[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) vc[j][i] = TR::from_f32(0.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the rest of the fill
    __syncthreads();                  // (and every wave has read K(0) before the next pair's pieces land on its tile)
    W64_STAMP(1)

#define W64_ITER_P(LAST_, DMA_, PAD_, LK, LV, SA0, SA1, SB0, SB1) \
    w64_iter<T, LAST_, PRE, DMA_, LSUM, PAD_>(LK, LV, fo, qf, SA0, SA1, SB0, SB1, negmu, o, m2, l_run, c, pb0, pb1, vc, st, chk, lacc, ones)
#define W64_ITER(LAST_, DMA_, LK, LV, SA0, SA1, SB0, SB1) W64_ITER_P(LAST_, DMA_, false, LK, LV, SA0, SA1, SB0, SB1)
#define W64_STAGE(KT, VT, KDST, VDST) { st.ksoff = (KT) * tB; st.vsoff = (VT) * tB; st.kdst = (KDST) + wave * 2048; st.vdst = (VDST) + wave * 2048; }
    int t = 0;                        // first tile of the pair on set A
#pragma unroll 1
    for (; t + 4 < Tg; t += 4) {
        // the next pair is staged while this one is consumed: each iteration carries two pieces of one K and one V tile
        W64_STAGE(t + 3, t + 2, ldsB[0], ldsB[2]) W64_ITER(false, true, ldsA[0], ldsA[2], X0, X1, Y0, Y1);
        W64_STAGE(t + 4, t + 3, ldsB[1], ldsB[3]) W64_ITER(false, true, ldsA[1], ldsA[3], Y0, Y1, X0, X1);
        if (!(GD_MP_DBG & 2)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        W64_STAGE(t + 5, t + 4, ldsA[0], ldsA[2]) W64_ITER(false, true, ldsB[0], ldsB[2], X0, X1, Y0, Y1);
        W64_STAGE(t + 6, t + 5, ldsA[1], ldsA[3]) W64_ITER(false, true, ldsB[1], ldsB[3], Y0, Y1, X0, X1);
        if (!(GD_MP_DBG & 2)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    // (the four iterations behind the loop: padded score MFMAs)
    W64_STAGE(t + 3, t + 2, ldsB[0], ldsB[2]) W64_ITER_P(false, true, true, ldsA[0], ldsA[2], X0, X1, Y0, Y1);
    W64_STAGE(t + 3, t + 3, ldsB[1], ldsB[3]) W64_ITER_P(false, true, true, ldsA[1], ldsA[3], Y0, Y1, X0, X1);   // no K(Tg): a harmless repeat of K(Tg-1)
    if (!(GD_MP_DBG & 2)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    W64_ITER_P(false, false, true, ldsB[0], ldsB[2], X0, X1, Y0, Y1);
    W64_ITER_P(true, false, true, ldsB[1], ldsB[3], Y0, Y1, X0, X1);
#undef W64_ITER
#undef W64_ITER_P
#undef W64_STAGE
    // The bias tiles stay LIVE to here.  Their last reader is an asm score MFMA of the second-to-last iteration, and the compiler — which
    // treats an asm as complete when it is issued — may hand a tile's registers to the very next vector instruction while the MFMA is
    // still reading them as SrcC (16 passes): seen in the first row-sum build, whose register pressure made it reuse negmu[1] for the
    // packed probabilities two instructions behind the MFMA — wrong scores for the last tile of query block B only, exact for block A
    // (tests/test_cabi.py now scans every instantiation for a vector write into an asm MFMA's sources behind it).
    asm volatile("" ::"v"(negmu[0]), "v"(negmu[1]));
    // second half of the last tile
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        o[b][0] = TR::mfma32(vc[0], as_frag<T>(pb0[b]), o[b][0]);
        o[b][1] = TR::mfma32(vc[1], as_frag<T>(pb0[b]), o[b][1]);
        o[b][0] = TR::mfma32(vc[2], as_frag<T>(pb1[b]), o[b][0]);
        o[b][1] = TR::mfma32(vc[3], as_frag<T>(pb1[b]), o[b][1]);
        if (LSUM) {
            lacc[b] = TR::mfma32(ones, as_frag<T>(pb0[b]), lacc[b]);
            lacc[b] = TR::mfma32(ones, as_frag<T>(pb1[b]), lacc[b]);
        }
    }
    if (LSUM) {
        // every lane holds the full row sum (both key halves of a k-step are summed by the MFMA); the code below adds the two lane
        // halves' l_run, so each gets half.  A sum that left [0, 2^60] or is NaN sends the segment to the exact-maxima attempt.
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            l_run[b] = 0.5f * lacc[b][0];
            chk = fmaxf(chk, fabsf(lacc[b][0]) == fabsf(lacc[b][0]) ? lacc[b][0] : __builtin_inff());
        }
    }
    W64_STAMP(2)
    if (attempt) break;
    // fp16 probabilities: a half-step sum of at most 2^14 keeps every one of its probabilities finite in 16 bits (vector-pipe sums); with the
    // row sums on the matrix pipe an overflowed probability is an infinite (or NaN) sum.  bf16 has fp32's range: only the sums can overflow.
    const float sum_limit = (__is_same(T, f16_t) && !LSUM) ? P_SUM_LIMIT : W64_SUM_LIMIT;
    if (__builtin_amdgcn_ballot_w64(!(chk <= sum_limit)) != 0 && lane == 0) w_abort = 1;
    __syncthreads();
    if (!w_abort) break;
    __syncthreads();                  // everyone has seen the flag before the next attempt clears it
  }

    bool write_out = true;
    if (SK && Tg != TU) {
        // Part of a unit (slots, counters, write-through stores as in k_attn_fwd_mp; a slot holds 4 waves x 2 query blocks x 9 chunks).
        // Two things differ from that kernel's hand-off (tools/w64_phases.py: store + ticket 3.3 us, merge 3.5-5.5 us per unit):
        //  * the workgroup that holds a unit's FIRST part first LOOKS at the unit's counter: if every other part is already there it
        //    is the merger and never stores its own part — its O tiles are the start of the accumulation anyway (with the linear range
        //    the workgroup that finishes a unit last holds its head, and that one is on the launch's critical path).  Any other
        //    merger reads all parts back, its own included: the fold runs in part order whoever merges, so the result does not depend
        //    on the arrival order;
        //  * both query blocks of a part are fetched in one round trip.
        const int w_first = P ? a.sk_unit0 + (unit - a.sk_unit0) * P : (unit * TU) / a.sk_tpw;
        const int w_last = P ? w_first + P - 1 : ((unit + 1) * TU - 1) / a.sk_tpw;
        const int others = w_last - w_first;
#define W64_SLOT(W2) (a.sk_ws + (size_t)(2 * (W2) + ((!P && (W2) * a.sk_tpw < unit * TU) ? 1 : 0)) * (2 * GD_SK_SLOT_F4) + (size_t)wave * 18 * 64 + lane)
        bool merger = false, published = false;
        // sk_mode 0 (gd_attn_cfg_t::handoff = 0): the same protocol under the memory model's terms — an agent-scope RELEASE fence between a
        // part's stores and its ticket, an agent-scope ACQUIRE fence between the observation "every other part has arrived" and the loads of
        // those parts.  Mode 1 (default) relies on the cache-policy bits alone (write-through sc0 sc1 stores drained with vmcnt(0) before
        // the ticket, sc0 sc1 loads served by L2): bit-identical results (tests/test_hip_kernels.py), 2-3 us per launch cheaper.
        if (wg == w_first) {
            if (tid == 0) {
                sk_last = a.sk_mode != 2 && __hip_atomic_fetch_add(a.sk_cnt + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == others;
                if (sk_last && a.sk_mode == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            merger = sk_last != 0;
            __syncthreads();                                   // everyone has read the flag before the ticket below rewrites it
        }
        if (!merger) {
            f32x4* const slot = W64_SLOT(wg);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 v = {o[b][j >> 2][4 * (j & 3)], o[b][j >> 2][4 * (j & 3) + 1], o[b][j >> 2][4 * (j & 3) + 2], o[b][j >> 2][4 * (j & 3) + 3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(slot + (b * 9 + j) * 64), "v"(v) : "memory");
                }
                const f32x4 v = {m2[b], l_run[b], 0.f, 0.f};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(slot + (b * 9 + 8) * 64), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (a.sk_mode == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const int ticket = a.sk_mode == 2 ? -1 : __hip_atomic_fetch_add(a.sk_cnt + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sk_last = ticket == others;
                if (sk_last && a.sk_mode == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            merger = sk_last != 0;
            published = true;
        }
        W64_STAMP(3)
        write_out = merger;
        if (merger) {
            if (tid == 0) __hip_atomic_store(a.sk_cnt + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero for the next launch
#pragma unroll 1
            for (int w2 = published ? w_first : w_first + 1; w2 <= w_last; ++w2) {
                const f32x4* const sl = W64_SLOT(w2);
                f32x4 pv[2][9];
                // device-scope loads of both query blocks, then one wait; the registers they write are operands of the wait (nothing
                // that reads them may be scheduled in front of it)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    asm volatile("global_load_dwordx4 %0, %9, off sc0 sc1\n\t"
                                 "global_load_dwordx4 %1, %9, off offset:1024 sc0 sc1\n\t"
                                 "global_load_dwordx4 %2, %9, off offset:2048 sc0 sc1\n\t"
                                 "global_load_dwordx4 %3, %9, off offset:3072 sc0 sc1\n\t"
                                 "global_load_dwordx4 %4, %10, off sc0 sc1\n\t"
                                 "global_load_dwordx4 %5, %10, off offset:1024 sc0 sc1\n\t"
                                 "global_load_dwordx4 %6, %10, off offset:2048 sc0 sc1\n\t"
                                 "global_load_dwordx4 %7, %10, off offset:3072 sc0 sc1\n\t"
                                 "global_load_dwordx4 %8, %11, off sc0 sc1"
                                 : "=&v"(pv[b][0]), "=&v"(pv[b][1]), "=&v"(pv[b][2]), "=&v"(pv[b][3]), "=&v"(pv[b][4]), "=&v"(pv[b][5]),
                                   "=&v"(pv[b][6]), "=&v"(pv[b][7]), "=&v"(pv[b][8])
                                 : "v"(sl + b * 9 * 64), "v"(sl + b * 9 * 64 + 4 * 64), "v"(sl + b * 9 * 64 + 8 * 64)
                                 : "memory");
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(pv[0][0]), "+v"(pv[0][1]), "+v"(pv[0][2]), "+v"(pv[0][3]), "+v"(pv[0][4]), "+v"(pv[0][5]), "+v"(pv[0][6]),
                               "+v"(pv[0][7]), "+v"(pv[0][8]), "+v"(pv[1][0]), "+v"(pv[1][1]), "+v"(pv[1][2]), "+v"(pv[1][3]), "+v"(pv[1][4]),
                               "+v"(pv[1][5]), "+v"(pv[1][6]), "+v"(pv[1][7]), "+v"(pv[1][8])
                             :
                             : "memory");
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (w2 == w_first) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[b][j >> 2][4 * (j & 3) + i] = pv[b][j][i];
                        m2[b] = pv[b][8][0]; l_run[b] = pv[b][8][1];
                    } else {
                        const float mk = pv[b][8][0], mn = fmaxf(m2[b], mk);
                        const float a0 = __builtin_amdgcn_exp2f(m2[b] - mn), a1 = __builtin_amdgcn_exp2f(mk - mn);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                o[b][j >> 2][4 * (j & 3) + i] = __builtin_fmaf(o[b][j >> 2][4 * (j & 3) + i], a0, pv[b][j][i] * a1);
                        l_run[b] = __builtin_fmaf(l_run[b], a0, pv[b][8][1] * a1);
                        m2[b] = mn;
                    }
                }
            }
        }
#undef W64_SLOT
    }

    W64_STAMP(4)
    if (write_out) {
        // Through LDS, one 32-query block at a time, so that the global stores are whole 128-byte rows: in the accumulator layout a lane
        // holds 4 columns of one row — written from there, every store instruction is 64 separate 8-byte pieces and the output phase took
        // 3.2 us of a 70 us launch (tools/w64_phases.py); 8 lanes x 16 bytes per row it takes well under 1 us.  Set A's tiles are free
        // here (last read one barrier ago; a split workgroup's next segment starts behind the barrier at the end of this one).  Row
        // pitch 144 B: the 8-byte writes and the 16-byte reads are both bank-conflict free.
        char* const stg = ldsA[0] + wave * (32 * 144);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int qrow = b ? qrowB : qrowA;
            const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typename TR::vec4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(o[b][dblk][4 * g + j] * inv);
                    *(typename TR::vec4*)(stg + (lane & 31) * 144 + (dblk * 32 + 8 * g + 4 * h) * (int)sizeof(T)) = w;
                }
            if (sg.lse && h == 0 && qrow < nq_ok) sg.lse[(size_t)bh * Nq + qrow] = m2[b] * 0.6931471805599453f + __logf(l_tot);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own writes, other lanes' rows: in order, no barrier
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3);
                const int row = tile * 256 + wave * 64 + b * 32 + r;
                const u32x4 val = *(const u32x4*)(stg + r * 144 + (lane & 7) * 16);
                if (row < nq_ok) *(u32x4*)((char*)((T*)sg.out + ooff + (size_t)row * rs) + (lane & 7) * 16) = val;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // block A's rows are in registers before block B overwrites them
        }
    }
    W64_STAMP(5)
#if GD_MP_DBG & 32
    ++dbg_seg;
#endif
    if (SK) {
        lin = P ? lin_end : lin + Tg;
        if (lin < lin_end) __syncthreads();
    }
  } while (SK && lin < lin_end);
#undef W64_ISSUE_TILE
#undef W64_PAD
#undef W64_STAMP
}

size_t gd_attn_sk_workspace_bytes(int tot_bh, int N, int M) {
    const size_t units = (size_t)tot_bh * ((N + 127) / 128);
    return GD_SK_SLOT_BYTES + units * sizeof(int);
}

// units of a launch: `unit_rows` queries of one head; a segment with a query row list has its own (shorter) row count
static int set_units(FwdArgs& a, int tot, int unit_rows) {
    a.tiles = (a.N + unit_rows - 1) / unit_rows;
    int bh_c = 0, cu = 0;
    a.ncseg = 0;
    for (int i = 0; i < a.nseg; ++i) {
        if (!a.seg[i].q_rows) continue;
        const int tc = (a.seg[i].q_rows_len + unit_rows - 1) / unit_rows;
        cu += tc * a.seg[i].bh;
        bh_c += a.seg[i].bh;
        a.tiles_cs[a.ncseg] = tc;
        a.cu_end[a.ncseg] = cu;
        a.cseg_of[a.ncseg] = (unsigned char)i;
        ++a.ncseg;
    }
    a.units_full = a.tiles * (tot - bh_c);
    return a.units_full + cu;
}

// sk_ws != NULL (set by the caller from its workspace): deal the key tiles out evenly where that pays
static bool sk_plan(FwdArgs& a, int tot, int qb, int ks) {
    const int TU = a.M / ATT_BN;
    // measured equal to the in-workgroup key ranges at 5 heads (31.9 vs 32.2 us) and behind k_attn_fwd_w64 from 10 heads up: only on request
    if (!a.sk_ws || !a.sk_force || qb != 4 || TU % 4 != 0 || TU < 16) return false;
    const long long total = (long long)a.nwg * TU;
    int tpw = (int)((total + GD_SK_SLOTS - 1) / GD_SK_SLOTS);
    tpw = (tpw + 3) & ~3;
    if (tpw < 16 || total > 0x7FFFFFFF) return false;                      // short runs: prologue / merge would dominate
    a.sk_tpw = tpw;
    a.sk_total = (int)total;
    a.nwg = (int)((total + tpw - 1) / tpw);
    return true;
}

// 64 queries per wave (k_attn_fwd_w64): units of 256 queries, one workgroup per CU
static int w64_launch(FwdArgs a, int tot, int pre, int dtype, hipStream_t st, bool sk_force) {
    const int TU = a.M / ATT_BN;
    GD_REQUIRE(a.M % (4 * ATT_BN) == 0, GD_EINVAL, "gd_attn_fwd: the 64-query kernel needs a multiple of four full key tiles (M=%d)", a.M);
    a.nwg = set_units(a, tot, 256);
    bool sk = false;
    // Even split where it beats whole units.  Cost model in key tiles per workgroup: unsplit = rounds of 256 workgroups x TU; split =
    // the equal share + ~24 tiles' worth of hand-off (a second prologue, the partial tiles through the workspace, ticket, merge: 17 us on
    // the pre-scaled variant, more on the exact-scale one).  Measured: 20 heads = 320 units, 128 vs 80 + 24 (109.9 -> 86.2 us); 15 heads
    // = 240 units, 64 vs 60 + 24 (59.1 unsplit, 70.0 split); 30 heads, 128 vs 120 + 24 (117.1 / 126.2); the optimisation pass's 165
    // units (two dense segments + the edit rows), 64 vs 42 + 24 (70.0 unsplit, 73.4 split: tools/bench_opt15.py).
    const long long tiles_unsplit = (long long)((a.nwg + 255) / 256) * TU, tiles_split = ((long long)a.nwg * TU + 255) / 256 + 24;
    // Few units (at most half the CUs: the 5-head inversion launch = 80 units): every unit in P equal runs, one per workgroup — one
    // prologue per workgroup and at most P - 1 parts to fetch for the merger (tools/w64_phases.py: with the linear range a quarter of the
    // workgroups walk two segments and a unit has up to five parts)
    if (a.sk_ws && a.nwg <= 128 && TU >= 48) {
        int P = 256 / a.nwg;
        if (P > 4) P = 4;
        const int tpp = (TU / P) & ~3;                 // parts 1 .. P-1; part 0 takes the rest (64 tiles in 3 parts: 24 + 20 + 20)
        if (P >= 2 && tpp >= 16) {
            a.sk_parts = P;
            a.sk_tpw = tpp;
            a.sk_total = a.nwg * TU;
            a.nwg *= P;
            sk = true;
        }
    }
    // One round of whole units plus a segment with a query row list (the CFG / optimisation pass of an edit: 240 + 5 units): the
    // workgroups of that segment build warped queries in their prologue (dependent gathers, ~5 us) and then walk the same 64 tiles as
    // everyone else — they finish last and the launch waits for them.  Where the CUs left over allow it, only THEIR units are cut
    // into parts (two or three workgroups each, 20-32 tiles): prologue + hand-off then end long before the dense workgroups.
    if (!sk && a.sk_ws && a.cseg >= 0 && a.nwg <= 256 && TU >= 48 && !sk_force) {
        const int cu = a.nwg - a.units_full;                    // units of the row-list segment
        int P = cu > 0 ? 1 + (256 - a.nwg) / cu : 1;
        if (P > 3) P = 3;
        const int tpp = (TU / (P > 1 ? P : 1)) & ~3;
        if (P >= 2 && tpp >= 16) {
            a.sk_parts = P;
            a.sk_unit0 = a.units_full;
            a.sk_tpw = tpp;
            a.sk_total = a.nwg * TU;
            a.nwg = a.units_full + cu * P;
            sk = true;
        }
    }
    if (!sk && a.sk_ws && TU >= 16 && (sk_force || tiles_split < tiles_unsplit)) {
        const long long total = (long long)a.nwg * TU;
        int tpw = (int)((total + 255) / 256);
        tpw = (tpw + 3) & ~3;
        if (tpw >= 16 && total <= 0x7FFFFFFF) {
            a.sk_tpw = tpw;
            a.sk_total = (int)total;
            a.nwg = (int)((total + tpw - 1) / tpw);
            sk = true;
        }
    }
#define GD_W64_LAUNCH(T_, PRE_, SK_) k_attn_fwd_w64<T_, PRE_, SK_><<<a.nwg, 256, 0, st>>>(a)
#define GD_W64_LAUNCH_L(T_, SK_) k_attn_fwd_w64<T_, true, SK_, true><<<a.nwg, 256, 0, st>>>(a)
    // pre-scaled variant with the row sums on the matrix pipe: where the caller asks for it (gd_attn_seg_t::q_scaled == 2: the
    // optimisation pass, whose L1 losses need numerator and denominator over the same rounded probabilities).  Measured (tools/
    // bench_lsum.py): no faster than the vector-pipe sums on the no-grad launches (15 heads 59 -> 62-66 us: the 8 extra MFMAs per tile cost
    // what the 32 v_add_f32 saved), 10-11 us faster than the exact-scale rescue variant it replaces in the optimisation pass (71 -> 60).
    const bool lsum = a.lsum != 0;
#define GD_W64_T(T_)                                                          \
    {                                                                         \
        if (pre && lsum) { if (sk) GD_W64_LAUNCH_L(T_, true); else GD_W64_LAUNCH_L(T_, false); }   \
        else if (pre) { if (sk) GD_W64_LAUNCH(T_, true, true); else GD_W64_LAUNCH(T_, true, false); }   \
        else { if (sk) GD_W64_LAUNCH(T_, false, true); else GD_W64_LAUNCH(T_, false, false); }     \
    }
    // fp16 (r06): the same variants.  The fixed reference lets a probability grow past fp16's range where bf16 has fp32's, so the segment's
    // check is tighter (see sum_limit in the kernel: half-step sums <= 2^14, or a finite matrix-pipe row sum) and a segment that fails it is
    // repeated with the exact row maxima like any other — exact for every input; r05 ran fp16 on the exact-scale rescue variant with a
    // multiplier of 1 (one fma per score more: 57.9 vs 50.4 us per launch over the bench's mix)
    if (dtype == GD_F16) GD_W64_T(f16_t)
    else GD_W64_T(bf16_t)
#undef GD_W64_T
#undef GD_W64_LAUNCH_L
#undef GD_W64_LAUNCH
    GD_CHECK_LAUNCH("gd_attn_fwd");
    return GD_OK;
}

int gd_attn_fwd_mp_launch(FwdArgs a, int qb, int ks, int dtype, hipStream_t st) {
    int tot = 0;
    for (int i = 0; i < a.nseg; ++i) tot = a.bh_end[i];
    {
        // below 160 units the 64-query kernel needs the even split's workspace (unit parts)
        const long long blocks = (long long)((a.N + 31) / 32) * tot;
        const int T = a.M / ATT_BN;
        if (qb == 8 && blocks < 1280 && !a.sk_ws) {
            qb = 4;
            ks = (blocks < 1280 && T % 4 == 0) ? 2 : 1;
        }
    }
    a.nwg = set_units(a, tot, 32 * qb);
    // Head order of the grid.  The 1-D grid is cut into 8 contiguous chunks, one per XCD (xcd_remap), and an XCD runs its chunk in
    // ascending order.  Segment after segment, the heads whose workgroups build warped queries in their prologue (~8 us of dependent
    // gathers) all land on two XCDs, which then finish last (measured: +17 us on a 106 us launch).  Interleaving the segments by
    // relative position spreads them over the XCDs, puts a warped head before the plain heads of its neighbourhood, and places the
    // edit_out / replace_out heads that share k_base / v_base on the same XCD's L2.
    a.n_order = 0;
    GD_REQUIRE(a.cseg < 0 || a.nseg == 1 || tot <= GD_ATTN_MAX_ORDER, GD_EUNSUPPORTED,
               "gd_attn_fwd: a query row list needs <= %d heads per launch (%d)", GD_ATTN_MAX_ORDER, tot);
    if (a.nseg > 1 && tot <= GD_ATTN_MAX_ORDER) {
        float key[GD_ATTN_MAX_ORDER];
        int n = 0, start = 0;
        for (int sgi = 0; sgi < a.nseg; ++sgi) {
            const int cnt = a.bh_end[sgi] - start;
            for (int i = 0; i < cnt && !a.seg[sgi].q_rows; ++i) {              // (the row-list segments' units come after the others')
                key[n] = ((float)i + (a.seg[sgi].warp_idx ? 0.25f : 0.5f)) / (float)cnt + 1e-4f * (float)sgi;
                a.order[n++] = (unsigned short)((sgi << 12) | i);
            }
            start = a.bh_end[sgi];
        }
        for (int i = 1; i < n; ++i) {                                        // insertion sort by key (n <= GD_ATTN_MAX_ORDER)
            const float kx = key[i];
            const unsigned short ox = a.order[i];
            int j = i - 1;
            for (; j >= 0 && key[j] > kx; --j) { key[j + 1] = key[j]; a.order[j + 1] = a.order[j]; }
            key[j + 1] = kx; a.order[j + 1] = ox;
        }
        a.n_order = n;
    }
    const int pre = a.q_prescaled ? 1 : 0;
    if (qb == 8) return w64_launch(a, tot, pre, dtype, st, a.sk_force != 0);
    if (sk_plan(a, tot, qb, ks)) {
#define GD_SK_LAUNCH(T_, PRE_) k_attn_fwd_mp<T_, 4, 1, 2, PRE_, true><<<a.nwg, 256, 0, st>>>(a)
        if (dtype == GD_F16) { if (pre) GD_SK_LAUNCH(f16_t, true); else GD_SK_LAUNCH(f16_t, false); }
        else { if (pre) GD_SK_LAUNCH(bf16_t, true); else GD_SK_LAUNCH(bf16_t, false); }
#undef GD_SK_LAUNCH
        GD_CHECK_LAUNCH("gd_attn_fwd");
        return GD_OK;
    }
    const int Tg = (a.M / ATT_BN) / ks;
    GD_REQUIRE(a.M % ATT_BN == 0 && (a.M / ATT_BN) % ks == 0 && Tg >= 2 && Tg % 2 == 0, GD_EINVAL,
               "gd_attn_fwd: the pipelined kernel needs an even number of full key tiles per key range (M=%d, KS=%d)", a.M, ks);
    const int nt = (ks <= 2 && Tg % 4 == 0) ? 2 : 1;                      // two tiles per barrier where LDS (64 KB per key range) allows
#define GD_MP_LAUNCH(QB_, KS_, NT_, PRE_)                                                                                \
    {                                                                                                                    \
        if (dtype == GD_F16) k_attn_fwd_mp<f16_t, QB_, KS_, NT_, PRE_, false><<<a.nwg, QB_ * KS_ * 64, 0, st>>>(a);              \
        else k_attn_fwd_mp<bf16_t, QB_, KS_, NT_, PRE_, false><<<a.nwg, QB_ * KS_ * 64, 0, st>>>(a);                            \
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
