// R3b — splat weights and the gather-composite (+ fused query blend).
//
// Replaces GeoDiffuser/utils/warp_utils.py:131-176 (alpha from dist2, pytorch3d alpha_composite,
// `.to(torch.half)`) and the blend at GeoDiffuser/utils/attention_processors.py:424,544.
// HBM-bound: per call it reads src once, idx/w once (shared by all heads) and writes out once:
// algorithmic bytes = 2*B*P*C*sizeof(T) + npix*K*8  (5.7 MB at 64^2, f=5, D=64).
#include "common.hpp"
#include "splat_common.hpp"

__global__ void k_splat_weights(const int32_t* __restrict__ idx, const float* __restrict__ dist2, int npix, int K,
                                float inv_rpow, float tau, float* __restrict__ w) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    float cum = 1.0f;
    for (int k = 0; k < K; ++k) {
        const size_t o = (size_t)pix * K + k;
        float wk = 0.0f;
        if (idx[o] >= 0) {
            float d = dist2[o] * inv_rpow;
            d = fminf(fmaxf(d, 1e-3f), 1.0f);
            float a = 1.0f - sqrtf(d);
            if (tau != 1.0f) a = powf(a, tau);
            wk = cum * a;
            cum = cum * (1.0f - a);
        }
        w[o] = wk;
    }
}

extern "C" int gd_splat_weights(const int32_t* idx, const float* dist2, int npix, int K,
                                float radius_ndc, float rad_pow, float tau, float* w, void* stream) {
    GD_REQUIRE(idx && dist2 && w, GD_EINVAL, "gd_splat_weights: null pointer");
    GD_REQUIRE(npix > 0 && K > 0, GD_EINVAL, "gd_splat_weights: bad sizes");
    // dist / pow(radius, rad_pow) — the division is kept as a division by the f32-rounded power
    const float rp = powf(radius_ndc, rad_pow);
    k_splat_weights<<<(npix + 255) / 256, 256, 0, as_stream(stream)>>>(idx, dist2, npix, K, 1.0f / rp, tau, w);
    GD_CHECK_LAUNCH("gd_splat_weights");
    return GD_OK;
}

// ---- token-major: src [B, P, C], one thread handles 8 consecutive channels of one (b, pix) ----------
template <typename T>
__global__ void k_composite_tok(const T* __restrict__ src, const int32_t* __restrict__ idx, const float* __restrict__ w,
                                const float* __restrict__ m, int B, int P, int C, int npix, int K, T* __restrict__ out) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const int cv = C >> 3;                                  // 16-byte chunks per row
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * npix * cv;
    if (gid >= total) return;
    const int ch = (int)(gid % cv);
    const long long t2 = gid / cv;
    const int pix = (int)(t2 % npix);
    const int b = (int)(t2 / npix);
    const T* sb = src + (size_t)b * P * C;
    const int coff[1] = {ch * 8};
    V8 o1[1];
    composite_chunks<T, 1>(sb, (size_t)C, coff, idx, w, m, pix, K, o1);      // splat_common.hpp (shared with the attention prologue)
    const V8 o = o1[0];
    *(V8*)(out + ((size_t)b * npix + pix) * C + ch * 8) = o;
}

// ---- channel-major: src [B, C, P] (latents, masks, images); one thread per (b, c, pix) ---------------
template <typename T>
__global__ void k_composite_chan(const T* __restrict__ src, const int32_t* __restrict__ idx, const float* __restrict__ w,
                                 const float* __restrict__ m, int B, int P, int C, int npix, int K, T* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * C * npix;
    if (gid >= total) return;
    const int pix = (int)(gid % npix);
    const long long bc = gid / npix;
    const T* sp = src + (size_t)bc * P;
    float acc = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 8) {
        int pk[8];
        float wk[8], fv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = k0 + j;
            const int p = kk < K ? idx[(size_t)pix * K + kk] : -1;
            pk[j] = p;
            wk[j] = (kk < K && p >= 0) ? w[(size_t)pix * K + kk] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) fv[j] = (float)sp[pk[j] < 0 ? 0 : pk[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (pk[j] >= 0) acc = __builtin_fmaf(wk[j], fv[j], acc);
    }
    float r = (float)(f16_t)acc;
    if (m) {
        const float mm = m[pix];
        r = (float)sp[pix] * (1.0f - mm) + mm * r;
    }
    out[(size_t)bc * npix + pix] = (T)r;
}

extern "C" int gd_splat_composite(const void* src, const int32_t* idx, const float* w, const float* m,
                                  int B, int P, int C, int npix, int K, int layout, void* out, int dtype, void* stream) {
    GD_REQUIRE(src && idx && w && out, GD_EINVAL, "gd_splat_composite: null pointer");
    GD_REQUIRE(B > 0 && P > 0 && C > 0 && npix > 0 && K > 0, GD_EINVAL, "gd_splat_composite: bad sizes");
    GD_REQUIRE(!m || P == npix, GD_EINVAL, "gd_splat_composite: blend needs P == npix");
    hipStream_t st = as_stream(stream);
    if (layout == GD_TOKEN_MAJOR) {
        GD_REQUIRE((C & 7) == 0, GD_EINVAL, "gd_splat_composite: token-major needs C %% 8 == 0 (C=%d)", C);
        GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_splat_composite: token-major is 16-bit only");
        const long long total = (long long)B * npix * (C >> 3);
        const int blocks = (int)((total + 255) / 256);
        if (dtype == GD_F16)
            k_composite_tok<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)src, idx, w, m, B, P, C, npix, K, (f16_t*)out);
        else
            k_composite_tok<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, idx, w, m, B, P, C, npix, K, (bf16_t*)out);
    } else if (layout == GD_CHANNEL_MAJOR) {
        const long long total = (long long)B * C * npix;
        const int blocks = (int)((total + 255) / 256);
        if (dtype == GD_F16)
            k_composite_chan<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)src, idx, w, m, B, P, C, npix, K, (f16_t*)out);
        else if (dtype == GD_BF16)
            k_composite_chan<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, idx, w, m, B, P, C, npix, K, (bf16_t*)out);
        else if (dtype == GD_F32)
            k_composite_chan<float><<<blocks, 256, 0, st>>>((const float*)src, idx, w, m, B, P, C, npix, K, (float*)out);
        else
            GD_REQUIRE(false, GD_EINVAL, "gd_splat_composite: bad dtype %d", dtype);
    } else {
        GD_REQUIRE(false, GD_EINVAL, "gd_splat_composite: bad layout %d", layout);
    }
    GD_CHECK_LAUNCH("gd_splat_composite");
    return GD_OK;
}
