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
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int kc = j < K ? j : K - 1;                                    // branch-free: all table loads issue together
        const int p = idx[(size_t)pix * K + kc];
        const float wv = w[(size_t)pix * K + kc];                            // unconditional: no index -> weight dependency
        pk[j] = j < K ? p : -1;
        wk[j] = pk[j] >= 0 ? wv : 0.0f;                                      // weight 0 for empty slots: fma(0, x, acc) == acc
    }
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
    // The K gathers of a pixel are independent and latency-bound: the row loads of two 8-channel chunks for all 16 slots are issued
    // together (32 loads in flight; empty slots fetch row 0 with weight 0 — a predicated load per slot would serialise the round
    // trips), then accumulated in slot order — the summation order per element never changes.
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (k0 > 0) {                                                        // K > 16: further table batches
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int kk = k0 + j;
                const int kc = kk < K ? kk : K - 1;
                const int p = idx[(size_t)pix * K + kc];
                const float wv = w[(size_t)pix * K + kc];
                pk[j] = kk < K ? p : -1;
                wk[j] = pk[j] >= 0 ? wv : 0.0f;
            }
        }
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += 2) {
            V8 f[2][16];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                if (c0 + cc < NCH) {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        f[cc][j] = *(const V8*)(sb + (size_t)(pk[j] < 0 ? 0 : pk[j]) * rs + coff[c0 + cc]);
                }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                if (c0 + cc < NCH) {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            acc[c0 + cc][i] = __builtin_fmaf(wk[j], TR::to_f32(f[cc][j][i]), acc[c0 + cc][i]);
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
