// The gather-composite + query blend of R3b/R7 for 8-channel chunks of one pixel, shared by the stand-alone splat kernel
// (k_composite_tok, splat.hip) and by the attention kernels' fused warped-query prologue (attn_fwd.hip): ONE implementation, so the
// fused path is bit-identical to the two-launch path by construction.
//
//   out = src[pix] * (1 - m) + m * half(sum_k w[pix,k] * src[idx[pix,k]])        (m == NULL: out = half(sum))
//
// Replaces pytorch3d alpha_composite + `.to(torch.half)` (GeoDiffuser/utils/warp_utils.py:156-176) and the blend at
// GeoDiffuser/utils/attention_processors.py:424,544 (op by op in the query dtype, as torch does on 16-bit tensors).
#pragma once
#include "common.hpp"

// One pixel's K <= 16 table slots (idx and w rows of K entries).  K == 15 (the splat's points_per_pixel on this path): four wide loads per
// table (4 + 4 + 4 + 3 entries; rows are only 4-byte aligned, which global loads accept) instead of 15 single ones — as single loads
// every instruction touched ~30 cache lines per wave (lane stride 60 B), 64 instructions per lane pair of tables: +5 us on the CFG
// pass's launch from the tables alone (tools/bench_cfg20.py).  Other K: one load per slot.  Slots >= K come back as idx -1 / w 0.
__device__ __forceinline__ void load_table_row(const int32_t* __restrict__ idx, const float* __restrict__ w, size_t pix, int K, int k0,
                                               int (&pk)[16], float (&wk)[16]) {
    if (K == 15 && k0 == 0) {
        typedef __attribute__((ext_vector_type(4))) int i32x4_u __attribute__((aligned(4)));
        typedef __attribute__((ext_vector_type(4))) float f32x4_u __attribute__((aligned(4)));
        typedef __attribute__((ext_vector_type(3))) int i32x3_u __attribute__((aligned(4)));
        typedef __attribute__((ext_vector_type(3))) float f32x3_u __attribute__((aligned(4)));
        const int32_t* ip = idx + pix * 15;
        const float* wp = w + pix * 15;
        const i32x4_u i0 = *(const i32x4_u*)ip, i1 = *(const i32x4_u*)(ip + 4), i2 = *(const i32x4_u*)(ip + 8);
        const i32x3_u i3 = *(const i32x3_u*)(ip + 12);
        const f32x4_u w0 = *(const f32x4_u*)wp, w1 = *(const f32x4_u*)(wp + 4), w2 = *(const f32x4_u*)(wp + 8);
        const f32x3_u w3 = *(const f32x3_u*)(wp + 12);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pk[j] = i0[j]; pk[4 + j] = i1[j]; pk[8 + j] = i2[j]; wk[j] = w0[j]; wk[4 + j] = w1[j]; wk[8 + j] = w2[j]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { pk[12 + j] = i3[j]; wk[12 + j] = w3[j]; }
        pk[15] = -1; wk[15] = 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kk = k0 + j;
            const int kc = kk < K ? kk : K - 1;                              // branch-free: all table loads issue together
            const int p = idx[pix * K + kc];
            const float wv = w[pix * K + kc];                                // unconditional: no index -> weight dependency
            pk[j] = kk < K ? p : -1;
            wk[j] = wv;
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) wk[j] = pk[j] >= 0 ? wk[j] : 0.0f;          // weight 0 for empty slots: fma(0, x, acc) == acc
}

// sb: first row of the cloud (batch / head offset applied), rs: row stride in elements, coff[c]: channel offset of chunk c.
template <typename T, int NCH>
__device__ __forceinline__ void composite_chunks(const T* __restrict__ sb, size_t rs, const int (&coff)[NCH],
                                                 const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                 const float* __restrict__ m, int pix, int K,
                                                 typename elem_traits<T>::vec8 (&out)[NCH]) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    // Everything that does not depend on another load is issued up front, in one round trip: the mask value, this pixel's own row
    // (needed for the blend) and the first 16 index / weight slots (K = 15 on this path).
    const float mm = m ? m[pix] : 1.0f;
    V8 qown[NCH];
    if (m) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) qown[c] = *(const V8*)(sb + (size_t)pix * rs + coff[c]);      // npix == P on this path
    }
    int pk[16];
    float wk[16];
    load_table_row(idx, w, (size_t)pix, K, 0, pk, wk);
    // Pixels outside the warped mask (m == 0: ~90 % of an attention map) keep their own row: q*1 + 0*splat == q, so the K gathers are
    // skipped for them (exec-masked; most waves skip entirely).  [With m == 0 the blend can differ from this only in the sign of a
    // zero (-0 + +0) or when the splat is non-finite; both kernels that use this function take the same shortcut.]
    if (m && mm == 0.0f) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) out[c] = qown[c];
        return;
    }
    float acc[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[c][i] = 0.0f;
    // Slots are filled front to back (the K nearest points in depth order, then -1): nvu = the wave's last filled slot + 1, wave-uniform
    // (ballots count active lanes only).  Slots from nvu on are empty in EVERY lane: their row loads and their fma(0, x, acc) — which
    // leaves acc unchanged bit for bit (acc is never -0: the weights are >= 0 and it starts at +0) — are skipped.  A 1.3-pixel splat fills
    // ~5 of the 15 slots: a third of the gathers and of the 15 x 64 multiply-adds per row (the fused prologue of the CFG pass's attention
    // launch, tools/bench_cfg20.py).

    // The K gathers of a pixel are independent and latency-bound: the row loads of two 8-channel chunks for all 16 slots are issued
    // together (32 loads in flight; empty slots fetch row 0 with weight 0 — a predicated load per slot would serialise the round
    // trips), then accumulated in slot order — the summation order per element never changes.
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (k0 > 0) load_table_row(idx, w, (size_t)pix, K, k0, pk, wk);      // K > 16: further table batches
        int nvu = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (__builtin_amdgcn_ballot_w64(pk[j] >= 0) != 0) nvu = j + 1;
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += 2) {
            V8 f[2][16];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                if (c0 + cc < NCH) {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < nvu) f[cc][j] = *(const V8*)(sb + (size_t)(pk[j] < 0 ? 0 : pk[j]) * rs + coff[c0 + cc]);
                }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                if (c0 + cc < NCH) {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < nvu) {
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                acc[c0 + cc][i] = __builtin_fmaf(wk[j], TR::to_f32(f[cc][j][i]), acc[c0 + cc][i]);
                        }
                }
        }
    }
    if (m) {
        const float one_m = TR::to_f32(TR::from_f32(1.0f - mm));
        const float m_t = TR::to_f32(TR::from_f32(mm));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const V8 q = qown[c];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float s16 = (float)(f16_t)acc[c][i];                   // `.to(torch.half)` U/warp_utils.py:176
                const float t1 = TR::to_f32(TR::from_f32(TR::to_f32(q[i]) * one_m));
                const float t2b = TR::to_f32(TR::from_f32(m_t * s16));
                out[c][i] = TR::from_f32(t1 + t2b);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) out[c][i] = TR::from_f32((float)(f16_t)acc[c][i]);
    }
}

// Two pixels per lane at once (the 64-query-per-wave attention kernel's fused prologue: query blocks A and B).  Same arithmetic per
// element as composite_chunks — acc = fma(w_j, x_j, acc) over the slots in ascending order, then the blend — so the result is bit-identical
// (tested against gd_splat_composite); what changes is the order of the LOADS: the tables of both pixels in one round trip, then the
// gathers of both pixels and all NCH chunks four slots at a time (32 row loads in flight, 2 batches for the ~5 filled slots of a
// 1.3-pixel splat).  Called per pixel this prologue was six dependent round trips (tables, chunk pair 0, chunk pair 1, twice): +16 us on
// the CFG pass's 64^2 launch once the even split had removed the slack that used to hide it (tools/bench_cfg20.py).
template <typename T, int NCH>
__device__ __forceinline__ void composite_chunks2(const T* __restrict__ sb, size_t rs, const int (&coff)[NCH],
                                                  const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                  const float* __restrict__ m, const int (&pix)[2], int K,
                                                  typename elem_traits<T>::vec8 (&out)[2][NCH]) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    float mm[2];
    V8 qown[2][NCH];
    int pk[2][16];
    float wk[2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        mm[p] = m ? m[pix[p]] : 1.0f;
        if (m) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) qown[p][c] = *(const V8*)(sb + (size_t)pix[p] * rs + coff[c]);
        }
    }
    bool need[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) need[p] = !(m && mm[p] == 0.0f);             // m == 0: the pixel keeps its own row (see composite_chunks)
    float acc[2][NCH][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[p][c][i] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load_table_row(idx, w, (size_t)pix[p], K, k0, pk[p], wk[p]);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                pk[p][j] = need[p] ? pk[p][j] : -1;
                wk[p][j] = need[p] ? wk[p][j] : 0.0f;
            }
        }
        int nvu = 0;                                                          // last slot filled in any lane / pixel of the wave, + 1
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (__builtin_amdgcn_ballot_w64(pk[0][j] >= 0 || pk[1][j] >= 0) != 0) nvu = j + 1;
#pragma unroll
        for (int b4 = 0; b4 < 16; b4 += 4) {
            if (b4 < nvu) {
                V8 f[2][NCH][4];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            f[p][c][jj] = *(const V8*)(sb + (size_t)(pk[p][b4 + jj] < 0 ? 0 : pk[p][b4 + jj]) * rs + coff[c]);
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                acc[p][c][i] = __builtin_fmaf(wk[p][b4 + jj], TR::to_f32(f[p][c][jj][i]), acc[p][c][i]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (m) {
            const float one_m = TR::to_f32(TR::from_f32(1.0f - mm[p]));
            const float m_t = TR::to_f32(TR::from_f32(mm[p]));
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const V8 q = qown[p][c];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float s16 = (float)(f16_t)acc[p][c][i];
                    const float t1 = TR::to_f32(TR::from_f32(TR::to_f32(q[i]) * one_m));
                    const float t2b = TR::to_f32(TR::from_f32(m_t * s16));
                    out[p][c][i] = need[p] ? TR::from_f32(t1 + t2b) : q[i];
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) out[p][c][i] = TR::from_f32((float)(f16_t)acc[p][c][i]);
        }
    }
}
