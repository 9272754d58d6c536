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
    float acc[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[c][i] = 0.0f;
    // the K gathers of a pixel are independent: fetch the index / weight slots 8 at a time, issue the row loads together
    // (a dependent idx -> row chain per slot is latency-bound), then accumulate in slot order
    for (int k0 = 0; k0 < K; k0 += 8) {
        int pk[8];
        float wk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = k0 + j;
            const int kc = kk < K ? kk : K - 1;                              // branch-free: all 16 table loads issue together
            const int p = idx[(size_t)pix * K + kc];
            pk[j] = kk < K ? p : -1;
            wk[j] = pk[j] >= 0 ? w[(size_t)pix * K + kc] : 0.0f;             // weight 0 for empty slots: fma(0, x, acc) == acc
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            V8 f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = *(const V8*)(sb + (size_t)(pk[j] < 0 ? 0 : pk[j]) * rs + coff[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[c][i] = __builtin_fmaf(wk[j], TR::to_f32(f[j][i]), acc[c][i]);
        }
    }
    if (m) {
        const float mm = m[pix];
        const float one_m = TR::to_f32(TR::from_f32(1.0f - mm));
        const float m_t = TR::to_f32(TR::from_f32(mm));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const V8 q = *(const V8*)(sb + (size_t)pix * rs + coff[c]);      // npix == P on this path
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
