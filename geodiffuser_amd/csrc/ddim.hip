// R10-R12 — scheduler / latent arithmetic: CFG combine + DDIM closed form, masked latent step, norm-preserve.
//
// Replaces GeoDiffuser/utils/diffusion.py:45-46,52 (CFG + DDIMScheduler.step, eta = 0), the inverse step
// GeoDiffuser/utils/inversion.py:57-65,185-190, GeoDiffuser/utils/optimization.py:216-231 and
// GeoDiffuser/utils/editor.py:219,316.  Tiny HBM-bound elementwise kernels (3 n * sizeof(T) bytes); their point is
// to remove ~10 separate torch launches per step from a launch-bound loop.
#include "common.hpp"

// VPRED: the model output is v = sqrt(a_t) eps - sqrt(1-a_t) x0 (SD2.1-768, BASELINE configs[3]); isa_t then carries sqrt(a_t).
template <typename T, bool VPRED>
__global__ void k_ddim_step(const T* __restrict__ x, const T* __restrict__ eu, const T* __restrict__ ec, float g,
                            float sb_t, float isa_t, float sa_to, float sb_to, T* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e = (float)eu[i];
    if (ec) e = e + g * ((float)ec[i] - e);
    float x0;
    if (VPRED) {
        const float xv = (float)x[i], v = e;
        x0 = isa_t * xv - sb_t * v;
        e = isa_t * v + sb_t * xv;
    } else {
        x0 = ((float)x[i] - sb_t * e) * isa_t;
    }
    out[i] = (T)(sa_to * x0 + sb_to * e);
}

template <bool VPRED>
static int ddim_step_launch(const void* x, const void* eps_u, const void* eps_c, float guidance, float a_t, float a_to,
                            void* out, int64_t n, int dtype, void* stream) {
    GD_REQUIRE(x && eps_u && out && n > 0, GD_EINVAL, "gd_ddim_step: null pointer or n<=0");
    GD_REQUIRE(a_t > 0.f && a_t <= 1.f && a_to > 0.f && a_to <= 1.f, GD_EINVAL, "gd_ddim_step: alphas out of (0,1]");
    const float sb_t = sqrtf(1.f - a_t), isa_t = VPRED ? sqrtf(a_t) : 1.f / sqrtf(a_t), sa_to = sqrtf(a_to), sb_to = sqrtf(1.f - a_to);
    const int blocks = (int)((n + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F32)
        k_ddim_step<float, VPRED><<<blocks, 256, 0, st>>>((const float*)x, (const float*)eps_u, (const float*)eps_c, guidance, sb_t, isa_t, sa_to, sb_to, (float*)out, n);
    else if (dtype == GD_F16)
        k_ddim_step<f16_t, VPRED><<<blocks, 256, 0, st>>>((const f16_t*)x, (const f16_t*)eps_u, (const f16_t*)eps_c, guidance, sb_t, isa_t, sa_to, sb_to, (f16_t*)out, n);
    else if (dtype == GD_BF16)
        k_ddim_step<bf16_t, VPRED><<<blocks, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)eps_u, (const bf16_t*)eps_c, guidance, sb_t, isa_t, sa_to, sb_to, (bf16_t*)out, n);
    else
        GD_REQUIRE(false, GD_EINVAL, "gd_ddim_step: bad dtype %d", dtype);
    GD_CHECK_LAUNCH("gd_ddim_step");
    return GD_OK;
}

extern "C" int gd_ddim_step(const void* x, const void* eps_u, const void* eps_c, float guidance, float a_t, float a_to,
                            void* out, int64_t n, int dtype, void* stream) {
    return ddim_step_launch<false>(x, eps_u, eps_c, guidance, a_t, a_to, out, n, dtype, stream);
}

extern "C" int gd_ddim_step_v(const void* x, const void* v_u, const void* v_c, float guidance, float a_t, float a_to,
                              void* out, int64_t n, int dtype, void* stream) {
    return ddim_step_launch<true>(x, v_u, v_c, guidance, a_t, a_to, out, n, dtype, stream);
}

__global__ void k_masked_update(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ m,
                                float step, int C, int hw, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * hw) return;
    float gv = g[i];
    if (!(gv == gv) || fabsf(gv) == INFINITY) gv = 0.f;          // torch.nan_to_num(nan=0, posinf=0, neginf=0)
    const float mm = m[i % hw];
    // two chained updates of U/optimization.py:230-231, kept as two roundings
    const float x1 = x[i] - 2.0f * mm * step * gv;
    out[i] = x1 - (1.0f - mm) * step * gv;
}

extern "C" int gd_masked_latent_update(const float* x, const float* g, const float* m, float step, int C, int hw,
                                       float* out, void* stream) {
    GD_REQUIRE(x && g && m && out && C > 0 && hw > 0, GD_EINVAL, "gd_masked_latent_update: bad argument");
    k_masked_update<<<(C * hw + 255) / 256, 256, 0, as_stream(stream)>>>(x, g, m, step, C, hw, out);
    GD_CHECK_LAUNCH("gd_masked_latent_update");
    return GD_OK;
}

// Deterministic: ONE workgroup, fixed summation order (thread-strided partials -> wave tree -> the 16 wave sums in index order), no
// atomics.  The latents this is called on are 16-37 K elements, so a single workgroup costs nothing against the launch itself.
__global__ void __launch_bounds__(1024)
k_sumsq(const float* __restrict__ x, long long n, float* __restrict__ acc) {
    __shared__ float part[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        s = __builtin_fmaf(v, v, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w];
        acc[0] += t;
    }
}

extern "C" int gd_sumsq(const float* x, int64_t n, float* sumsq, void* stream) {
    GD_REQUIRE(x && sumsq && n > 0, GD_EINVAL, "gd_sumsq: bad argument");
    k_sumsq<<<1, 1024, 0, as_stream(stream)>>>(x, n, sumsq);
    GD_CHECK_LAUNCH("gd_sumsq");
    return GD_OK;
}

__global__ void k_norm_rescale(const float* __restrict__ x, const float* __restrict__ num, const float* __restrict__ den,
                               long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = sqrtf(num[0] + 1e-12f) / sqrtf(den[0] + 1e-12f);
    out[i] = x[i] * s;
}

extern "C" int gd_norm_rescale(const float* x, const float* num_sumsq, const float* den_sumsq, int64_t n, float* out, void* stream) {
    GD_REQUIRE(x && num_sumsq && den_sumsq && out && n > 0, GD_EINVAL, "gd_norm_rescale: bad argument");
    k_norm_rescale<<<(int)((n + 255) / 256), 256, 0, as_stream(stream)>>>(x, num_sumsq, den_sumsq, n, out);
    GD_CHECK_LAUNCH("gd_norm_rescale");
    return GD_OK;
}
