// Fused GroupNorm (+ SiLU) for channels-last 16-bit activations — UNet plumbing around the hot path.
//
// Not part of the reference's attention-sharing path itself: it replaces torch.nn.GroupNorm + F.silu inside the SD-shaped
// UNet harness (geodiffuser_amd/unet_sd21.py) on no-grad passes.  PyTorch-ROCm's GroupNorm works on NCHW-contiguous data, so
// in a channels-last network (MIOpen's igemm convolutions are NHWC) every norm costs two layout copies plus four kernels; the
// profile of an edit showed ~18 % of the GPU time there.  HBM-bound: reads x twice, writes y once.
//
//   k_gn_stats : one workgroup per (batch, pixel slab): coalesced 16-B reads of whole channel rows, per-channel partial sums
//                in registers, per-group reduction through LDS atomics, one f32 global atomic per (group, moment)
//   k_gn_apply : y = (x - mean) * rstd * gamma + beta, optional SiLU, 8 channels (16 B) per thread
#include "common.hpp"

#define GN_MAX_G 64
#define GN_PIX 32          // pixels per workgroup in the stats pass

template <typename T>
__global__ void __launch_bounds__(256)
k_gn_stats(const T* __restrict__ x, int HW, int C, int G, float* __restrict__ stats) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ float s_g[GN_MAX_G * 2];
    const int b = blockIdx.y, p0 = blockIdx.x * GN_PIX, tid = threadIdx.x;
    const int cv = C >> 3, cpg = C / G;
    const T* xb = x + (size_t)b * HW * C;
    if (tid < G * 2) s_g[tid] = 0.f;
    __syncthreads();
    const int p1 = (p0 + GN_PIX) < HW ? (p0 + GN_PIX) : HW;
    // Every thread owns one 8-channel column k of the slab and walks pixels with a fixed stride, so its partial sums stay in
    // registers; consecutive threads read consecutive 16-B chunks (coalesced).  cv <= 256: (256 / cv) pixel rows in flight;
    // cv > 256: a thread takes columns tid and tid + 256.
    const int rows = cv <= 256 ? 256 / cv : 1;
    for (int k = (cv <= 256 ? tid % cv : tid); k < cv; k += 256) {
        const int r = cv <= 256 ? tid / cv : 0;
        if (r < rows) {
            float sum[8], sq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sum[i] = 0.f; sq[i] = 0.f; }
            for (int p = p0 + r; p < p1; p += rows) {
                const V8 v = *(const V8*)(xb + (size_t)p * C + k * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float f = TR::to_f32(v[i]);
                    sum[i] += f;
                    sq[i] = __builtin_fmaf(f, f, sq[i]);
                }
            }
            // an 8-channel column touches at most two groups (C / G >= 8)
            const int c0 = k * 8, g0 = c0 / cpg, g1 = (c0 + 7) / cpg, split = g1 * cpg - c0;
            float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (g1 != g0 && i >= split) { a1 += sum[i]; q1 += sq[i]; } else { a0 += sum[i]; q0 += sq[i]; }
            }
            atomicAdd(&s_g[g0 * 2], a0); atomicAdd(&s_g[g0 * 2 + 1], q0);
            if (g1 != g0) { atomicAdd(&s_g[g1 * 2], a1); atomicAdd(&s_g[g1 * 2 + 1], q1); }
        }
        if (cv <= 256) break;
    }
    __syncthreads();
    if (tid < G * 2) atomicAdd(&stats[(size_t)b * G * 2 + tid], s_g[tid]);
}

template <typename T, bool SILU>
__global__ void k_gn_apply(const T* __restrict__ x, const float* __restrict__ stats, const T* __restrict__ gamma,
                           const T* __restrict__ beta, int B, int HW, int C, int G, float eps, T* __restrict__ y) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const int cv = C >> 3, cpg = C / G;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * HW * cv) return;
    const int k = (int)(gid % cv);
    const int b = (int)(gid / ((long long)HW * cv));
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    const int c0 = k * 8;
    const int g0 = c0 / cpg, g1 = (c0 + 7) / cpg;
    const float* st = stats + (size_t)b * G * 2;
    const float m0 = st[g0 * 2] * inv_n, m1 = st[g1 * 2] * inv_n;
    const float r0 = rsqrtf(fmaxf(st[g0 * 2 + 1] * inv_n - m0 * m0, 0.f) + eps);
    const float r1 = rsqrtf(fmaxf(st[g1 * 2 + 1] * inv_n - m1 * m1, 0.f) + eps);
    const int split = (g1 * cpg) - c0;            // elements i >= split belong to g1
    const V8 v = *(const V8*)(x + gid * 8);
    const V8 ga = *(const V8*)(gamma + c0), be = *(const V8*)(beta + c0);
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool hi = (g1 != g0) && (i >= split);
        const float mean = hi ? m1 : m0, rstd = hi ? r1 : r0;
        float t = (TR::to_f32(v[i]) - mean) * rstd * TR::to_f32(ga[i]) + TR::to_f32(be[i]);
        if (SILU) t = t / (1.0f + __expf(-t));
        o[i] = TR::from_f32(t);
    }
    *(V8*)(y + gid * 8) = o;
}

extern "C" int gd_group_norm_nhwc(const void* x, const void* gamma, const void* beta, int B, int HW, int C, int G, float eps,
                                  int silu, float* stats /* [B,G,2] f32 scratch */, void* y, int dtype, void* stream) {
    GD_REQUIRE(x && gamma && beta && stats && y, GD_EINVAL, "gd_group_norm_nhwc: null pointer");
    GD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && G <= GN_MAX_G && C % G == 0 && (C & 7) == 0 && C / G >= 8, GD_EINVAL,
               "gd_group_norm_nhwc: unsupported shape B=%d HW=%d C=%d G=%d (need C %% 8 == 0, C/G >= 8, G <= %d)", B, HW, C, G, GN_MAX_G);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_group_norm_nhwc: dtype must be f16/bf16");
    hipStream_t st = as_stream(stream);
    gd_zero_async(stats, (size_t)B * G * 2 * sizeof(float), st);
    dim3 sgrid((HW + GN_PIX - 1) / GN_PIX, B);
    const long long total = (long long)B * HW * (C >> 3);
    const int ablocks = (int)((total + 255) / 256);
    if (dtype == GD_F16) {
        k_gn_stats<f16_t><<<sgrid, 256, 0, st>>>((const f16_t*)x, HW, C, G, stats);
        if (silu) k_gn_apply<f16_t, true><<<ablocks, 256, 0, st>>>((const f16_t*)x, stats, (const f16_t*)gamma, (const f16_t*)beta, B, HW, C, G, eps, (f16_t*)y);
        else k_gn_apply<f16_t, false><<<ablocks, 256, 0, st>>>((const f16_t*)x, stats, (const f16_t*)gamma, (const f16_t*)beta, B, HW, C, G, eps, (f16_t*)y);
    } else {
        k_gn_stats<bf16_t><<<sgrid, 256, 0, st>>>((const bf16_t*)x, HW, C, G, stats);
        if (silu) k_gn_apply<bf16_t, true><<<ablocks, 256, 0, st>>>((const bf16_t*)x, stats, (const bf16_t*)gamma, (const bf16_t*)beta, B, HW, C, G, eps, (bf16_t*)y);
        else k_gn_apply<bf16_t, false><<<ablocks, 256, 0, st>>>((const bf16_t*)x, stats, (const bf16_t*)gamma, (const bf16_t*)beta, B, HW, C, G, eps, (bf16_t*)y);
    }
    GD_CHECK_LAUNCH("gd_group_norm_nhwc");
    return GD_OK;
}
