// UNet plumbing around the hot path (not rows of SURVEY 8a): element-wise fusions for the no-grad passes of the SD-shaped UNet
// harness.  One UNet pass at batch 1-3 is ~550 kernels of 5-20 us; the GPU is launch-latency-bound there, so what counts is
// the NUMBER of kernels between the convolutions / GEMMs, not their bandwidth.  All tensors are 16-bit, channels-last / token-major
// ([rows, C] with C % 8 == 0), 16-byte accesses.
//
//   gd_bias_residual : y = x + bias[c] (+ res)         — convolution epilogue (MIOpen adds the bias as a separate kernel and the
//                                                        residual add is another one)
//   gd_geglu         : y = x[:, :C] * gelu(x[:, C:])   — GEGLU gate of the transformer feed-forward (exact erf GELU)
//   gd_add_layer_norm: s = a + b;  y = LayerNorm(s)    — residual add fused with the next LayerNorm (s is also written)
#include "common.hpp"

template <typename T>
__global__ void k_bias_residual(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ res, long long nvec, int cv,
                                T* __restrict__ y) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const int k = (int)(i % cv);
    const V8 a = *(const V8*)(x + i * 8), b = *(const V8*)(bias + k * 8);
    V8 o;
    if (res) {
        const V8 r = *(const V8*)(res + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // same rounding points as the unfused ops: conv + bias -> T, then residual add -> T
            const float t = TR::to_f32(TR::from_f32(TR::to_f32(a[j]) + TR::to_f32(b[j])));
            o[j] = TR::from_f32(t + TR::to_f32(r[j]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = TR::from_f32(TR::to_f32(a[j]) + TR::to_f32(b[j]));
    }
    *(V8*)(y + i * 8) = o;
}

extern "C" int gd_bias_residual(const void* x, const void* bias, const void* res, int64_t rows, int C, void* y, int dtype, void* stream) {
    GD_REQUIRE(x && bias && y, GD_EINVAL, "gd_bias_residual: null pointer");
    GD_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0, GD_EINVAL, "gd_bias_residual: need C %% 8 == 0 (C=%d)", C);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_bias_residual: dtype must be f16/bf16");
    const long long nvec = (long long)rows * (C >> 3);
    const int blocks = (int)((nvec + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_bias_residual<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, (const f16_t*)bias, (const f16_t*)res, nvec, C >> 3, (f16_t*)y);
    else
        k_bias_residual<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)bias, (const bf16_t*)res, nvec, C >> 3, (bf16_t*)y);
    GD_CHECK_LAUNCH("gd_bias_residual");
    return GD_OK;
}

template <typename T>
__global__ void k_geglu(const T* __restrict__ x, long long nvec, int cv, T* __restrict__ y) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const long long row = i / cv;
    const int k = (int)(i - row * cv);
    const T* xr = x + row * (long long)cv * 16;                 // row of 2C elements
    const V8 h = *(const V8*)(xr + k * 8), g = *(const V8*)(xr + (cv + k) * 8);
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gf = TR::to_f32(g[j]);
        const float ge = TR::to_f32(TR::from_f32(0.5f * gf * (1.0f + erff(gf * 0.70710678118654752440f))));   // F.gelu, rounded like torch
        o[j] = TR::from_f32(TR::to_f32(h[j]) * ge);
    }
    *(V8*)(y + i * 8) = o;
}

extern "C" int gd_geglu(const void* x, int64_t rows, int C, void* y, int dtype, void* stream) {
    GD_REQUIRE(x && y, GD_EINVAL, "gd_geglu: null pointer");
    GD_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0, GD_EINVAL, "gd_geglu: need C %% 8 == 0 (C=%d)", C);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_geglu: dtype must be f16/bf16");
    const long long nvec = (long long)rows * (C >> 3);
    const int blocks = (int)((nvec + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) k_geglu<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, nvec, C >> 3, (f16_t*)y);
    else k_geglu<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, nvec, C >> 3, (bf16_t*)y);
    GD_CHECK_LAUNCH("gd_geglu");
    return GD_OK;
}

// backward of y = h * gelu(gate) w.r.t. x = [h | gate] (the optimisation pass differentiates through the feed-forward layers): one launch
// instead of autograd's mul / gelu_backward / mul / cat.  Rounding points as autograd has them: gelu(gate) and dy * h are 16-bit tensors there.
//   dh = dy * gelu(gate);  dgate = (dy * h) * gelu'(gate),  gelu'(g) = 0.5 (1 + erf(g / sqrt 2)) + g exp(-g^2 / 2) / sqrt(2 pi)
template <typename T>
__global__ void k_geglu_bwd(const T* __restrict__ x, const T* __restrict__ dy, long long nvec, int cv, T* __restrict__ dx) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const long long row = i / cv;
    const int k = (int)(i - row * cv);
    const T* xr = x + row * (long long)cv * 16;
    const V8 h = *(const V8*)(xr + k * 8), g = *(const V8*)(xr + (cv + k) * 8), d = *(const V8*)(dy + i * 8);
    V8 oh, og;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gf = TR::to_f32(g[j]), df = TR::to_f32(d[j]);
        const float cdf = 0.5f * (1.0f + erff(gf * 0.70710678118654752440f));
        const float ge = TR::to_f32(TR::from_f32(gf * cdf));
        oh[j] = TR::from_f32(df * ge);
        const float dge = TR::to_f32(TR::from_f32(df * TR::to_f32(h[j])));
        og[j] = TR::from_f32(dge * (cdf + gf * 0.39894228040143267794f * __expf(-0.5f * gf * gf)));
    }
    T* dr = dx + row * (long long)cv * 16;
    *(V8*)(dr + k * 8) = oh;
    *(V8*)(dr + (cv + k) * 8) = og;
}

extern "C" int gd_geglu_bwd(const void* x, const void* dy, int64_t rows, int C, void* dx, int dtype, void* stream) {
    GD_REQUIRE(x && dy && dx, GD_EINVAL, "gd_geglu_bwd: null pointer");
    GD_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0, GD_EINVAL, "gd_geglu_bwd: need C %% 8 == 0 (C=%d)", C);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_geglu_bwd: dtype must be f16/bf16");
    const long long nvec = (long long)rows * (C >> 3);
    const int blocks = (int)((nvec + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) k_geglu_bwd<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, (const f16_t*)dy, nvec, C >> 3, (f16_t*)dx);
    else k_geglu_bwd<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, nvec, C >> 3, (bf16_t*)dx);
    GD_CHECK_LAUNCH("gd_geglu_bwd");
    return GD_OK;
}

// one wave per row; C <= 8 * 64 * LN_MAXV
#define LN_MAXV 4
template <typename T>
__global__ void __launch_bounds__(256)
k_add_layer_norm(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ gamma, const T* __restrict__ beta,
                 long long rows, int C, float eps, T* __restrict__ sum_out, T* __restrict__ y) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int cv = C >> 3;
    const T* ar = a + row * C;
    const T* br = b ? b + row * C : nullptr;
    float v[LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int k = lane + 64 * t;
        if (k < cv) {
            const V8 x = *(const V8*)(ar + k * 8);
            if (br) {
                const V8 z = *(const V8*)(br + k * 8);
                V8 sm;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sm[j] = TR::from_f32(TR::to_f32(x[j]) + TR::to_f32(z[j]));      // the residual stream stays 16-bit, as unfused
                    v[t][j] = TR::to_f32(sm[j]);
                }
                if (sum_out) *(V8*)(sum_out + row * C + k * 8) = sm;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[t][j] = TR::to_f32(x[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[t][j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int k = lane + 64 * t;
        if (k < cv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[t][j] - mean; q = __builtin_fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int k = lane + 64 * t;
        if (k < cv) {
            const V8 ga = *(const V8*)(gamma + k * 8), be = *(const V8*)(beta + k * 8);
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = TR::from_f32((v[t][j] - mean) * rstd * TR::to_f32(ga[j]) + TR::to_f32(be[j]));
            *(V8*)(y + row * C + k * 8) = o;
        }
    }
}

// backward of y = LayerNorm(s) * gamma + beta w.r.t. s for frozen gamma / beta, plus the gradient that reaches s directly (s is also
// the residual stream): ds = rstd * (g - mean(g) - xh * mean(g * xh)) + gs,  g = gy * gamma,  xh = (s - mean) * rstd.  One wave per row,
// mean / rstd recomputed from s (nothing saved by the forward but s itself).
template <typename T>
__global__ void __launch_bounds__(256)
k_layer_norm_bwd(const T* __restrict__ s_in, const T* __restrict__ gamma, const T* __restrict__ gy, const T* __restrict__ gs,
                 long long rows, int C, float eps, T* __restrict__ ds) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int cv = C >> 3;
    float v[LN_MAXV][8], g[LN_MAXV][8];
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int k = lane + 64 * t;
        if (k < cv) {
            const V8 x = *(const V8*)(s_in + row * C + k * 8);
            const V8 d = *(const V8*)(gy + row * C + k * 8);
            const V8 ga = *(const V8*)(gamma + k * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[t][j] = TR::to_f32(x[j]);
                g[t][j] = TR::to_f32(d[j]) * TR::to_f32(ga[j]);
                sm += v[t][j];
            }
        }
    }
    const float mean = wave_sum(sm) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t)
        if (lane + 64 * t < cv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[t][j] - mean; q = __builtin_fmaf(d, d, q); }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t)
        if (lane + 64 * t < cv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[t][j] = (v[t][j] - mean) * rstd;                     // xh
                a1 += g[t][j];
                a2 = __builtin_fmaf(g[t][j], v[t][j], a2);
            }
        }
    const float m1 = wave_sum(a1) / (float)C, m2 = wave_sum(a2) / (float)C;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int k = lane + 64 * t;
        if (k < cv) {
            V8 o;
            if (gs) {
                const V8 u = *(const V8*)(gs + row * C + k * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = TR::from_f32(rstd * (g[t][j] - m1 - v[t][j] * m2) + TR::to_f32(u[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = TR::from_f32(rstd * (g[t][j] - m1 - v[t][j] * m2));
            }
            *(V8*)(ds + row * C + k * 8) = o;
        }
    }
}

extern "C" int gd_layer_norm_bwd(const void* s, const void* gamma, const void* gy, const void* gs, int64_t rows, int C, float eps, void* ds,
                                 int dtype, void* stream) {
    GD_REQUIRE(s && gamma && gy && ds, GD_EINVAL, "gd_layer_norm_bwd: null pointer");
    GD_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0 && C <= 8 * 64 * LN_MAXV, GD_EINVAL, "gd_layer_norm_bwd: need C %% 8 == 0 and C <= %d (C=%d)",
               8 * 64 * LN_MAXV, C);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_layer_norm_bwd: dtype must be f16/bf16");
    const int blocks = (int)((rows + 3) / 4);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_layer_norm_bwd<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)s, (const f16_t*)gamma, (const f16_t*)gy, (const f16_t*)gs, rows, C, eps, (f16_t*)ds);
    else
        k_layer_norm_bwd<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)s, (const bf16_t*)gamma, (const bf16_t*)gy, (const bf16_t*)gs, rows, C, eps, (bf16_t*)ds);
    GD_CHECK_LAUNCH("gd_layer_norm_bwd");
    return GD_OK;
}

extern "C" int gd_add_layer_norm(const void* a, const void* b, const void* gamma, const void* beta, int64_t rows, int C, float eps,
                                 void* sum_out, void* y, int dtype, void* stream) {
    GD_REQUIRE(a && gamma && beta && y, GD_EINVAL, "gd_add_layer_norm: null pointer");
    GD_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0 && C <= 8 * 64 * LN_MAXV, GD_EINVAL,
               "gd_add_layer_norm: need C %% 8 == 0 and C <= %d (C=%d)", 8 * 64 * LN_MAXV, C);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_add_layer_norm: dtype must be f16/bf16");
    const int blocks = (int)((rows + 3) / 4);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_add_layer_norm<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)a, (const f16_t*)b, (const f16_t*)gamma, (const f16_t*)beta, rows, C, eps,
                                                        (f16_t*)sum_out, (f16_t*)y);
    else
        k_add_layer_norm<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)gamma, (const bf16_t*)beta, rows, C, eps,
                                                         (bf16_t*)sum_out, (bf16_t*)y);
    GD_CHECK_LAUNCH("gd_add_layer_norm");
    return GD_OK;
}
