// Fused GroupNorm (+ SiLU) for channels-last 16-bit activations — UNet plumbing around the hot path.
//
// Not part of the reference's attention-sharing path itself: it replaces torch.nn.GroupNorm + F.silu inside the SD-shaped
// UNet harness (geodiffuser_amd/unet_sd21.py) on no-grad passes.  PyTorch-ROCm's GroupNorm works on NCHW-contiguous data, so
// in a channels-last network (MIOpen's igemm convolutions are NHWC) every norm costs two layout copies plus four kernels; the
// profile of an edit showed ~18 % of the GPU time there.  HBM-bound: reads x twice, writes y once.
//
//   k_gn_stats : one workgroup per (batch, pixel slab): coalesced 16-B reads of whole channel rows, per-channel partial sums
//                in registers, per-group reduction through LDS in a FIXED order (column slots, then the columns of a group: no float
//                atomics anywhere, the kernels are bit-reproducible), ONE plain store of the slab's [G,2] partial moments
//   k_gn_apply : every workgroup first folds the <= 64 slab partials of its batch entry into mean / rstd in LDS, then
//                y = (x - mean) * rstd * gamma + beta, optional SiLU, 8 channels (16 B) per thread
// Optional add_bc [B,C]: the norm is taken of x + add_bc[b,c] (the ResNet block's time-embedding add folded in).
#include "common.hpp"

#define GN_MAX_G 64
#define GN_MAX_SLABS 64

// Sum the slab partials of batch entry b into s_out[G*2] (LDS): all 256 threads take part — (256 / (2G)) threads per value, each
// adding every (256 / 2G)-th slab with independent loads (a single thread per value walked <= 64 dependent-looking loads: 6 us).
__device__ __forceinline__ void gn_fold_partials(const float* __restrict__ partial, int nslab, int G, int b, float* s_out) {
    // Deterministic: the `parts` partial sums of a value go through LDS and are added in ascending order by ONE thread (no LDS float
    // atomics: their arrival order changed the last bits from launch to launch, and a random-init UNet amplifies that to 1e-2).
    __shared__ float s_fold[256];
    const int tid = threadIdx.x, nv = G * 2, parts = 256 / nv;          // G <= 64 -> parts >= 2
    __syncthreads();                                                    // s_fold may still be read by a previous fold
    if (tid < nv * parts) {
        const int v = tid % nv, p = tid / nv;
        const float* pp = partial + (size_t)b * nslab * nv + v;
        float a0 = 0.f, a1 = 0.f;
        int s = p;
        for (; s + parts < nslab; s += 2 * parts) { a0 += pp[(size_t)s * nv]; a1 += pp[(size_t)(s + parts) * nv]; }
        if (s < nslab) a0 += pp[(size_t)s * nv];
        s_fold[p * nv + v] = a0 + a1;
    }
    __syncthreads();
    if (tid < nv) {
        float t = 0.f;
        for (int p = 0; p < parts; ++p) t += s_fold[p * nv + tid];
        s_out[tid] = t;
    }
    __syncthreads();
}

// Deterministic per-group reduction of the threads' column sums (replaces atomicAdd(&s_g[...]) from every thread).  A thread owns the
// 8-channel column k of pixel-row slot r and holds (a0, q0) for the column's first group g0 and (a1, q1) for its second group g1 (an
// 8-channel column touches at most two groups).  The slots of a column are added in ascending r, the columns of a group in ascending k.
#define GN_MAX_CV 512
struct GnRed {
    float rc[256][4];            // per-thread contributions (cv <= 256)
    float col[GN_MAX_CV][4];     // per-column totals
};

__device__ __forceinline__ void gn_store_contrib(GnRed& red, int cv, int k, float a0, float q0, float a1, float q1) {
    float* d = cv <= 256 ? red.rc[threadIdx.x] : red.col[k];
    d[0] = a0; d[1] = q0; d[2] = a1; d[3] = q1;
}

__device__ __forceinline__ void gn_group_reduce(GnRed& red, int cv, int rows, int cpg, int G, float* s_g) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (cv <= 256) {
        if (tid < cv) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            for (int r = 0; r < rows; ++r) {
                const float* d = red.rc[r * cv + tid];
                t0 += d[0]; t1 += d[1]; t2 += d[2]; t3 += d[3];
            }
            red.col[tid][0] = t0; red.col[tid][1] = t1; red.col[tid][2] = t2; red.col[tid][3] = t3;
        }
        __syncthreads();
    }
    if (tid < G * 2) {
        const int g = tid >> 1, w = tid & 1;
        const int k_lo = (g * cpg) >> 3, k_hi = ((g + 1) * cpg - 1) >> 3;
        float t = 0.f;
        for (int k = k_lo; k <= k_hi; ++k) {
            const int c0 = k * 8, g0 = c0 / cpg, g1 = (c0 + 7) / cpg;
            if (g0 == g) t += red.col[k][w];
            if (g1 != g0 && g1 == g) t += red.col[k][2 + w];
        }
        s_g[tid] = t;
    }
    __syncthreads();
}

static inline int gn_pix_per_slab(int HW) {
    int p = (HW + GN_MAX_SLABS - 1) / GN_MAX_SLABS;
    return p < 4 ? 4 : p;             // small maps: many thin slabs rather than two workgroups walking 32 pixels each
}

template <typename T>
__global__ void __launch_bounds__(256)
k_gn_stats(const T* __restrict__ x, const T* __restrict__ add_bc, int add_ld, int HW, int C, int G, int pix, float* __restrict__ partial) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ float s_g[GN_MAX_G * 2];
    __shared__ GnRed red;
    const int b = blockIdx.y, p0 = blockIdx.x * pix, tid = threadIdx.x;
    const int cv = C >> 3, cpg = C / G;
    const T* xb = x + (size_t)b * HW * C;
    if (cv <= 256) gn_store_contrib(red, cv, 0, 0.f, 0.f, 0.f, 0.f);       // threads without a column contribute zeros
    const int p1 = (p0 + pix) < HW ? (p0 + pix) : HW;
    // Every thread owns one 8-channel column k of the slab and walks pixels with a fixed stride, so its partial sums stay in
    // registers; consecutive threads read consecutive 16-B chunks (coalesced).  cv <= 256: (256 / cv) pixel rows in flight;
    // cv > 256: a thread takes columns tid and tid + 256.
    const int rows = cv <= 256 ? 256 / cv : 1;
    for (int k = (cv <= 256 ? tid % cv : tid); k < cv; k += 256) {
        const int r = cv <= 256 ? tid / cv : 0;
        if (r < rows) {
            float sum[8], sq[8], ad[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sum[i] = 0.f; sq[i] = 0.f; ad[i] = 0.f; }
            if (add_bc) {
                const V8 av = *(const V8*)(add_bc + (size_t)b * add_ld + k * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) ad[i] = TR::to_f32(av[i]);
            }
            for (int p = p0 + r; p < p1; p += 4 * rows) {            // four pixel rows in flight per thread
                V8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = p + u * rows;
                    v[u] = *(const V8*)(xb + (size_t)(pp < p1 ? pp : p) * C + k * 8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (p + u * rows >= p1) break;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        // x + add is rounded to the storage type first, as the unfused `h + temb[:, :, None, None]` does
                        const float f = add_bc ? TR::to_f32(TR::from_f32(TR::to_f32(v[u][i]) + ad[i])) : TR::to_f32(v[u][i]);
                        sum[i] += f;
                        sq[i] = __builtin_fmaf(f, f, sq[i]);
                    }
                }
            }
            // an 8-channel column touches at most two groups (C / G >= 8)
            const int c0 = k * 8, g0 = c0 / cpg, g1 = (c0 + 7) / cpg, split = g1 * cpg - c0;
            float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (g1 != g0 && i >= split) { a1 += sum[i]; q1 += sq[i]; } else { a0 += sum[i]; q0 += sq[i]; }
            }
            gn_store_contrib(red, cv, k, a0, q0, a1, q1);
        }
        if (cv <= 256) break;
    }
    gn_group_reduce(red, cv, rows, cpg, G, s_g);
    if (tid < G * 2) partial[((size_t)b * gridDim.x + blockIdx.x) * G * 2 + tid] = s_g[tid];
}

template <typename T, bool SILU>
__global__ void __launch_bounds__(256)
k_gn_apply(const T* __restrict__ x, const T* __restrict__ add_bc, int add_ld, const float* __restrict__ partial, int nslab,
           const T* __restrict__ gamma, const T* __restrict__ beta, int HW, int C, int G, float eps, T* __restrict__ y) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ float s_m[GN_MAX_G * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cv = C >> 3, cpg = C / G;
    gn_fold_partials(partial, nslab, G, b, s_m);
    const long long idx = (long long)blockIdx.x * 256 + tid;          // vector index inside batch entry b
    if (idx >= (long long)HW * cv) return;
    const int k = (int)(idx % cv);
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    const int c0 = k * 8;
    const int g0 = c0 / cpg, g1 = (c0 + 7) / cpg;
    const float m0 = s_m[g0 * 2] * inv_n, m1 = s_m[g1 * 2] * inv_n;
    const float r0 = rsqrtf(fmaxf(s_m[g0 * 2 + 1] * inv_n - m0 * m0, 0.f) + eps);
    const float r1 = rsqrtf(fmaxf(s_m[g1 * 2 + 1] * inv_n - m1 * m1, 0.f) + eps);
    const int split = (g1 * cpg) - c0;            // elements i >= split belong to g1
    const size_t off = ((size_t)b * HW * cv + idx) * 8;
    const V8 v = *(const V8*)(x + off);
    const V8 ga = *(const V8*)(gamma + c0), be = *(const V8*)(beta + c0);
    V8 av;
    if (add_bc) av = *(const V8*)(add_bc + (size_t)b * add_ld + c0);
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool hi = (g1 != g0) && (i >= split);
        const float mean = hi ? m1 : m0, rstd = hi ? r1 : r0;
        const float f = add_bc ? TR::to_f32(TR::from_f32(TR::to_f32(v[i]) + TR::to_f32(av[i]))) : TR::to_f32(v[i]);
        float t = (f - mean) * rstd * TR::to_f32(ga[i]) + TR::to_f32(be[i]);
        if (SILU) t = t / (1.0f + __expf(-t));
        o[i] = TR::from_f32(t);
    }
    *(V8*)(y + off) = o;
}

// ---- single-launch variant for maps whose (batch entry, group) slice fits one workgroup's LDS -------------------------------
// One workgroup of 1024 threads per (group, batch entry): the slice (HW pixels x C/G channels, <= 40 KB) is read ONCE, rounded
// (x + add) values are parked in LDS while the moments are summed, then normalised from LDS.  One launch instead of two and no
// partial-moment round trip: at the UNet's sizes both kernels of the two-launch form sit on the launch floor (5-7 us each), so this
// is what a norm costs less (tools/bench_gn.py).  Addressing: a pixel's group segment is C/G * 2 bytes (20 ... 160 B, 4-byte aligned),
// handled in 2-channel (4-byte) units; LP = the power of two >= units per pixel lanes walk one pixel, so a thread's unit index j —
// and with it its add / gamma / beta values — is loop-invariant and no division is needed.  Reductions in a fixed order (wave
// shuffles, then one thread over the 16 wave sums): bit-reproducible.  The slab-partial scratch of the two-launch form is filled
// compatibly (totals in slab 0, zeros elsewhere) for the backward kernels.
#define GN1_THREADS 1024
#define GN1_MAX_LDS (128 * 1024)

template <typename T, bool SILU>
__global__ void __launch_bounds__(GN1_THREADS)
k_gn_fused(const T* __restrict__ x, const T* __restrict__ add_bc, int add_ld, const T* __restrict__ gamma, const T* __restrict__ beta,
           int HW, int C, int G, int LP, float eps, float* __restrict__ partial, int nslab, T* __restrict__ y) {
    using TR = elem_traits<T>;
    extern __shared__ __attribute__((aligned(16))) uint32_t gn1_lds[];        // [HW][U] units of two 16-bit values
    __shared__ float s_w[2][GN1_THREADS / 64];
    __shared__ float s_mr[2];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G, U = cpg >> 1;
    const int j = tid & (LP - 1), slot = tid / LP, nslot = GN1_THREADS / LP;
    const bool act = j < U;
    const size_t base = (size_t)b * HW * C + (size_t)g * cpg + 2 * (act ? j : 0);
    float ad0 = 0.f, ad1 = 0.f;
    if (add_bc && act) {
        const uint32_t aw = *(const uint32_t*)(add_bc + (size_t)b * add_ld + g * cpg + 2 * j);
        T a2[2];
        __builtin_memcpy(a2, &aw, 4);
        ad0 = TR::to_f32(a2[0]); ad1 = TR::to_f32(a2[1]);
    }
    float sum = 0.f, sq = 0.f;
    if (act) {
        int p = slot;
        for (; p + 3 * nslot < HW; p += 4 * nslot) {                        // four pixels in flight per thread
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = *(const uint32_t*)(x + base + (size_t)(p + u * nslot) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                T v2[2];
                __builtin_memcpy(v2, &w[u], 4);
                if (add_bc) {           // x + add is rounded to the storage type first, as the unfused `h + temb[:, :, None, None]` does
                    v2[0] = TR::from_f32(TR::to_f32(v2[0]) + ad0); v2[1] = TR::from_f32(TR::to_f32(v2[1]) + ad1);
                    __builtin_memcpy(&w[u], v2, 4);
                }
                const float f0 = TR::to_f32(v2[0]), f1 = TR::to_f32(v2[1]);
                sum += f0 + f1;
                sq = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, sq));
                gn1_lds[(size_t)(p + u * nslot) * U + j] = w[u];
            }
        }
        for (; p < HW; p += nslot) {
            uint32_t w = *(const uint32_t*)(x + base + (size_t)p * C);
            T v2[2];
            __builtin_memcpy(v2, &w, 4);
            if (add_bc) {
                v2[0] = TR::from_f32(TR::to_f32(v2[0]) + ad0); v2[1] = TR::from_f32(TR::to_f32(v2[1]) + ad1);
                __builtin_memcpy(&w, v2, 4);
            }
            const float f0 = TR::to_f32(v2[0]), f1 = TR::to_f32(v2[1]);
            sum += f0 + f1;
            sq = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, sq));
            gn1_lds[(size_t)p * U + j] = w;
        }
    }
    sum = wave_sum(sum); sq = wave_sum(sq);
    if ((tid & 63) == 0) { s_w[0][tid >> 6] = sum; s_w[1][tid >> 6] = sq; }
    __syncthreads();
    if (tid == 0) {
        float ts = 0.f, tq = 0.f;
        for (int w = 0; w < GN1_THREADS / 64; ++w) { ts += s_w[0][w]; tq += s_w[1][w]; }
        const float inv_n = 1.0f / ((float)HW * (float)cpg);
        const float m = ts * inv_n;
        s_mr[0] = m;
        s_mr[1] = rsqrtf(fmaxf(tq * inv_n - m * m, 0.f) + eps);
        partial[((size_t)b * nslab) * G * 2 + g * 2] = ts;                   // what k_gn_stats would have left, folded: slab 0 = totals
        partial[((size_t)b * nslab) * G * 2 + g * 2 + 1] = tq;
    } else if (tid < nslab) {
        partial[((size_t)b * nslab + tid) * G * 2 + g * 2] = 0.f;
        partial[((size_t)b * nslab + tid) * G * 2 + g * 2 + 1] = 0.f;
    }
    __syncthreads();
    if (!act) return;
    const float mean = s_mr[0], rstd = s_mr[1];
    float ga0, ga1, be0, be1;
    {
        T t2[2];
        const uint32_t gw = *(const uint32_t*)(gamma + g * cpg + 2 * j), bw = *(const uint32_t*)(beta + g * cpg + 2 * j);
        __builtin_memcpy(t2, &gw, 4); ga0 = TR::to_f32(t2[0]); ga1 = TR::to_f32(t2[1]);
        __builtin_memcpy(t2, &bw, 4); be0 = TR::to_f32(t2[0]); be1 = TR::to_f32(t2[1]);
    }
    for (int p = slot; p < HW; p += nslot) {
        const uint32_t w = gn1_lds[(size_t)p * U + j];
        T v2[2];
        __builtin_memcpy(v2, &w, 4);
        // same association as k_gn_apply: ((f - mean) * rstd) * gamma + beta
        float t0 = (TR::to_f32(v2[0]) - mean) * rstd * ga0 + be0, t1 = (TR::to_f32(v2[1]) - mean) * rstd * ga1 + be1;
        if (SILU) { t0 = t0 / (1.0f + __expf(-t0)); t1 = t1 / (1.0f + __expf(-t1)); }
        v2[0] = TR::from_f32(t0); v2[1] = TR::from_f32(t1);
        uint32_t o;
        __builtin_memcpy(&o, v2, 4);
        *(uint32_t*)(y + base + (size_t)p * C) = o;
    }
}

// single_launch: the caller's flag (-1 / 1: the rule below, 0: always the two-launch form — benchmarks / tests compare both)
static bool gn_single_ok(int HW, int C, int G, int single_launch) {
    const int cpg = C / G;
    // measured (tools/bench_gn.py, batch 1 and 3): 8^2 and 16^2 maps 7.5-13 -> 4-10 us, 32^2 a tie around 40 KB slices, 64^2 x 320
    // (80 KB slices on 32-96 workgroups) 12 -> 21 us: one launch up to 40 KB per (batch entry, group)
    return single_launch != 0 && (cpg & 1) == 0 && cpg <= 128 && (size_t)HW * cpg * 2 <= 40 * 1024;
}

extern "C" int64_t gd_group_norm_nhwc_scratch_floats(int B, int HW, int G) {
    if (B <= 0 || HW <= 0 || G <= 0) return 0;
    const int pix = gn_pix_per_slab(HW);
    return (int64_t)B * ((HW + pix - 1) / pix) * G * 2;
}

extern "C" int gd_group_norm_nhwc(const void* x, const void* add_bc, int add_ld, const void* gamma, const void* beta, int B, int HW, int C, int G,
                                  float eps, int silu, int single_launch, float* scratch, void* y, int dtype, void* stream) {
    GD_REQUIRE(x && gamma && beta && scratch && y, GD_EINVAL, "gd_group_norm_nhwc: null pointer");
    // an 8-channel vector may touch at most two groups: C/G >= 8, or exactly 4 (the VAE's 128-channel norms)
    GD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && G <= GN_MAX_G && C % G == 0 && (C & 7) == 0 && (C / G >= 8 || C / G == 4) && C <= 8 * GN_MAX_CV, GD_EINVAL,
               "gd_group_norm_nhwc: unsupported shape B=%d HW=%d C=%d G=%d (need C %% 8 == 0, C <= %d, C/G >= 8 or == 4, G <= %d)", B, HW, C, G, 8 * GN_MAX_CV, GN_MAX_G);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_group_norm_nhwc: dtype must be f16/bf16");
    GD_REQUIRE(!add_bc || add_ld == 0 || (add_ld >= C && (add_ld & 7) == 0), GD_EINVAL, "gd_group_norm_nhwc: add_ld must be >= C and a multiple of 8");
    if (add_ld == 0) add_ld = C;
    hipStream_t st = as_stream(stream);
    const int pix = gn_pix_per_slab(HW);
    const int nslab = (HW + pix - 1) / pix;
    if (gn_single_ok(HW, C, G, single_launch)) {
        const int U = (C / G) >> 1;
        int LP = 1;
        while (LP < U) LP <<= 1;
        const size_t lds = (size_t)HW * U * 4;
        dim3 grid(G, B);
        static bool attr_set = false;             // all four instantiations at once (the first call must not fall inside a capture)
        if (!attr_set) {
            hipFuncSetAttribute((const void*)k_gn_fused<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GN1_MAX_LDS);
            hipFuncSetAttribute((const void*)k_gn_fused<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GN1_MAX_LDS);
            hipFuncSetAttribute((const void*)k_gn_fused<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GN1_MAX_LDS);
            hipFuncSetAttribute((const void*)k_gn_fused<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GN1_MAX_LDS);
            attr_set = true;
        }
#define GD_GN1(T_, S_)                                                                                                   \
    k_gn_fused<T_, S_><<<grid, GN1_THREADS, lds, st>>>((const T_*)x, (const T_*)add_bc, add_ld, (const T_*)gamma, (const T_*)beta, HW, C, G, LP, \
                                                       eps, scratch, nslab, (T_*)y);
        if (dtype == GD_F16) { if (silu) GD_GN1(f16_t, true) else GD_GN1(f16_t, false) }
        else { if (silu) GD_GN1(bf16_t, true) else GD_GN1(bf16_t, false) }
#undef GD_GN1
        GD_CHECK_LAUNCH("gd_group_norm_nhwc");
        return GD_OK;
    }
    dim3 sgrid(nslab, B);
    dim3 agrid((unsigned)(((long long)HW * (C >> 3) + 255) / 256), B);
    if (dtype == GD_F16) {
        k_gn_stats<f16_t><<<sgrid, 256, 0, st>>>((const f16_t*)x, (const f16_t*)add_bc, add_ld, HW, C, G, pix, scratch);
        if (silu) k_gn_apply<f16_t, true><<<agrid, 256, 0, st>>>((const f16_t*)x, (const f16_t*)add_bc, add_ld, scratch, nslab, (const f16_t*)gamma, (const f16_t*)beta, HW, C, G, eps, (f16_t*)y);
        else k_gn_apply<f16_t, false><<<agrid, 256, 0, st>>>((const f16_t*)x, (const f16_t*)add_bc, add_ld, scratch, nslab, (const f16_t*)gamma, (const f16_t*)beta, HW, C, G, eps, (f16_t*)y);
    } else {
        k_gn_stats<bf16_t><<<sgrid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)add_bc, add_ld, HW, C, G, pix, scratch);
        if (silu) k_gn_apply<bf16_t, true><<<agrid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)add_bc, add_ld, scratch, nslab, (const bf16_t*)gamma, (const bf16_t*)beta, HW, C, G, eps, (bf16_t*)y);
        else k_gn_apply<bf16_t, false><<<agrid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)add_bc, add_ld, scratch, nslab, (const bf16_t*)gamma, (const bf16_t*)beta, HW, C, G, eps, (bf16_t*)y);
    }
    GD_CHECK_LAUNCH("gd_group_norm_nhwc");
    return GD_OK;
}


// ---- backward (optimisation passes; weights are frozen, so only dx) -------------------------------------------------------
//   f = x (+ add);  xh = (f - mean) * rstd;  z = xh * gamma + beta;  y = SILU ? z * sigmoid(z) : z
//   dz = dy * (SILU ? sigmoid(z) * (1 + z * (1 - sigmoid(z))) : 1);  g = dz * gamma
//   dx = rstd * (g - S1 / n - xh * S2 / n),   S1 = sum_group g,  S2 = sum_group g * xh,   n = HW * C / G
// Same two-kernel shape as the forward: slab partials of (S1, S2), then the apply kernel folds them.  mean / rstd come from the
// forward's slab partials (kept by the caller), so x is read twice and dy twice, nothing else.
template <typename T>
__device__ __forceinline__ void gn_fold_fwd_stats(const float* __restrict__ fwd_partial, int nslab, int G, int b, float inv_n, float eps,
                                                  float* s_mr /* [G][2] mean, rstd */) {
    const int tid = threadIdx.x;
    gn_fold_partials(fwd_partial, nslab, G, b, s_mr);                  // (sum, sum of squares) per group
    float m = 0.f, r = 0.f;
    if (tid < G) {
        m = s_mr[tid * 2] * inv_n;
        r = rsqrtf(fmaxf(s_mr[tid * 2 + 1] * inv_n - m * m, 0.f) + eps);
    }
    __syncthreads();
    if (tid < G) { s_mr[tid * 2] = m; s_mr[tid * 2 + 1] = r; }
    __syncthreads();
}

template <typename T, bool SILU>
__device__ __forceinline__ float gn_g(float xv, float addv, bool has_add, float mean, float rstd, float ga, float be, float dyv, float& xh) {
    using TR = elem_traits<T>;
    const float f = has_add ? TR::to_f32(TR::from_f32(xv + addv)) : xv;
    xh = (f - mean) * rstd;
    float dz = dyv;
    if (SILU) {
        const float z = xh * ga + be;
        const float sg = 1.0f / (1.0f + __expf(-z));
        dz *= sg * (1.0f + z * (1.0f - sg));
    }
    return dz * ga;
}

template <typename T, bool SILU>
__global__ void __launch_bounds__(256)
k_gn_bwd_stats(const T* __restrict__ x, const T* __restrict__ add_bc, int add_ld, const T* __restrict__ gamma, const T* __restrict__ beta,
               const T* __restrict__ dy, const float* __restrict__ fwd_partial, int nslab, int HW, int C, int G, int pix, float eps,
               float* __restrict__ partial) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ float s_mr[GN_MAX_G * 2];
    __shared__ float s_g[GN_MAX_G * 2];
    __shared__ GnRed red;
    const int b = blockIdx.y, p0 = blockIdx.x * pix, tid = threadIdx.x;
    const int cv = C >> 3, cpg = C / G;
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    if (cv <= 256) gn_store_contrib(red, cv, 0, 0.f, 0.f, 0.f, 0.f);
    gn_fold_fwd_stats<T>(fwd_partial, nslab, G, b, inv_n, eps, s_mr);
    const size_t boff = (size_t)b * HW * C;
    const int p1 = (p0 + pix) < HW ? (p0 + pix) : HW;
    const int rows = cv <= 256 ? 256 / cv : 1;
    for (int k = (cv <= 256 ? tid % cv : tid); k < cv; k += 256) {
        const int r = cv <= 256 ? tid / cv : 0;
        if (r < rows) {
            const int c0 = k * 8, g0 = c0 / cpg, g1 = (c0 + 7) / cpg, split = g1 * cpg - c0;
            const V8 gav = *(const V8*)(gamma + c0), bev = *(const V8*)(beta + c0);
            float ad[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ad[i] = 0.f;
            if (add_bc) {
                const V8 av = *(const V8*)(add_bc + (size_t)b * add_ld + c0);
#pragma unroll
                for (int i = 0; i < 8; ++i) ad[i] = TR::to_f32(av[i]);
            }
            float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
            for (int p = p0 + r; p < p1; p += rows) {
                const V8 xv = *(const V8*)(x + boff + (size_t)p * C + c0);
                const V8 dv = *(const V8*)(dy + boff + (size_t)p * C + c0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool hi = (g1 != g0) && (i >= split);
                    float xh;
                    const float g = gn_g<T, SILU>(TR::to_f32(xv[i]), ad[i], add_bc != nullptr, s_mr[(hi ? g1 : g0) * 2], s_mr[(hi ? g1 : g0) * 2 + 1],
                                                  TR::to_f32(gav[i]), TR::to_f32(bev[i]), TR::to_f32(dv[i]), xh);
                    if (hi) { a1 += g; q1 = __builtin_fmaf(g, xh, q1); } else { a0 += g; q0 = __builtin_fmaf(g, xh, q0); }
                }
            }
            gn_store_contrib(red, cv, k, a0, q0, a1, q1);
        }
        if (cv <= 256) break;
    }
    gn_group_reduce(red, cv, rows, cpg, G, s_g);
    if (tid < G * 2) partial[((size_t)b * gridDim.x + blockIdx.x) * G * 2 + tid] = s_g[tid];
}

template <typename T, bool SILU>
__global__ void __launch_bounds__(256)
k_gn_bwd_apply(const T* __restrict__ x, const T* __restrict__ add_bc, int add_ld, const T* __restrict__ gamma, const T* __restrict__ beta,
               const T* __restrict__ dy, const float* __restrict__ fwd_partial, const float* __restrict__ partial, int nslab, int HW, int C,
               int G, float eps, T* __restrict__ dx) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ float s_mr[GN_MAX_G * 2];
    __shared__ float s_s[GN_MAX_G * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cv = C >> 3, cpg = C / G;
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    gn_fold_partials(partial, nslab, G, b, s_s);
    if (tid < G * 2) s_s[tid] *= inv_n;                            // S1 / n, S2 / n  (made visible by the barriers of the next fold)
    gn_fold_fwd_stats<T>(fwd_partial, nslab, G, b, inv_n, eps, s_mr);
    const long long idx = (long long)blockIdx.x * 256 + tid;
    if (idx >= (long long)HW * cv) return;
    const int k = (int)(idx % cv);
    const int c0 = k * 8, g0 = c0 / cpg, g1 = (c0 + 7) / cpg, split = g1 * cpg - c0;
    const size_t off = ((size_t)b * HW * cv + idx) * 8;
    const V8 xv = *(const V8*)(x + off), dv = *(const V8*)(dy + off);
    const V8 gav = *(const V8*)(gamma + c0), bev = *(const V8*)(beta + c0);
    V8 av;
    if (add_bc) av = *(const V8*)(add_bc + (size_t)b * add_ld + c0);
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gi = ((g1 != g0) && (i >= split)) ? g1 : g0;
        float xh;
        const float g = gn_g<T, SILU>(TR::to_f32(xv[i]), add_bc ? TR::to_f32(av[i]) : 0.f, add_bc != nullptr, s_mr[gi * 2], s_mr[gi * 2 + 1],
                                      TR::to_f32(gav[i]), TR::to_f32(bev[i]), TR::to_f32(dv[i]), xh);
        o[i] = TR::from_f32(s_mr[gi * 2 + 1] * (g - s_s[gi * 2] - xh * s_s[gi * 2 + 1]));
    }
    *(V8*)(dx + off) = o;
}

// single-launch backward for the small maps (same eligibility and addressing as k_gn_fused): one workgroup per (group, batch entry)
// parks the slice's x (+ add) and dy in LDS while S1 = sum g and S2 = sum g * xh are reduced, then writes dx from LDS.
template <typename T, bool SILU>
__global__ void __launch_bounds__(GN1_THREADS)
k_gn_bwd_fused(const T* __restrict__ x, const T* __restrict__ add_bc, int add_ld, const T* __restrict__ gamma, const T* __restrict__ beta,
               const T* __restrict__ dy, const float* __restrict__ fwd_partial, int nslab, int HW, int C, int G, int LP, float eps,
               T* __restrict__ dx) {
    using TR = elem_traits<T>;
    extern __shared__ __attribute__((aligned(16))) uint32_t gn1_lds[];        // [2][HW][U]: x units, then dy units
    __shared__ float s_w[2][GN1_THREADS / 64];
    __shared__ float s_c[4];                                                  // mean, rstd, S1 / n, S2 / n
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G, U = cpg >> 1;
    const int j = tid & (LP - 1), slot = tid / LP, nslot = GN1_THREADS / LP;
    const bool act = j < U;
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    // the forward's moments of this (batch entry, group): one thread per slab partial loads (a single thread walking <= 64 slabs paid
    // 64 dependent-looking loads: 25 us), then thread 0 adds them from LDS in ascending order
    __shared__ float s_f[2][GN_MAX_SLABS];
    if (tid < nslab) {
        s_f[0][tid] = fwd_partial[((size_t)b * nslab + tid) * G * 2 + g * 2];
        s_f[1][tid] = fwd_partial[((size_t)b * nslab + tid) * G * 2 + g * 2 + 1];
    }
    __syncthreads();
    if (tid == 0) {
        float ts = 0.f, tq = 0.f;
        for (int sl = 0; sl < nslab; ++sl) { ts += s_f[0][sl]; tq += s_f[1][sl]; }
        const float m = ts * inv_n;
        s_c[0] = m;
        s_c[1] = rsqrtf(fmaxf(tq * inv_n - m * m, 0.f) + eps);
    }
    __syncthreads();
    const float mean = s_c[0], rstd = s_c[1];
    const size_t base = (size_t)b * HW * C + (size_t)g * cpg + 2 * (act ? j : 0);
    float ad0 = 0.f, ad1 = 0.f, ga0 = 0.f, ga1 = 0.f, be0 = 0.f, be1 = 0.f;
    if (act) {
        T t2[2];
        uint32_t w_;
        if (add_bc) {
            w_ = *(const uint32_t*)(add_bc + (size_t)b * add_ld + g * cpg + 2 * j);
            __builtin_memcpy(t2, &w_, 4); ad0 = TR::to_f32(t2[0]); ad1 = TR::to_f32(t2[1]);
        }
        w_ = *(const uint32_t*)(gamma + g * cpg + 2 * j); __builtin_memcpy(t2, &w_, 4); ga0 = TR::to_f32(t2[0]); ga1 = TR::to_f32(t2[1]);
        w_ = *(const uint32_t*)(beta + g * cpg + 2 * j); __builtin_memcpy(t2, &w_, 4); be0 = TR::to_f32(t2[0]); be1 = TR::to_f32(t2[1]);
    }
    uint32_t* const lx = gn1_lds;
    uint32_t* const ld = gn1_lds + (size_t)HW * U;
    float s1 = 0.f, s2 = 0.f;
    if (act) {
        for (int p = slot; p < HW; p += nslot) {
            const uint32_t xw = *(const uint32_t*)(x + base + (size_t)p * C), dw = *(const uint32_t*)(dy + base + (size_t)p * C);
            T x2[2], d2[2];
            __builtin_memcpy(x2, &xw, 4); __builtin_memcpy(d2, &dw, 4);
            float xh0, xh1;
            const float g0 = gn_g<T, SILU>(TR::to_f32(x2[0]), ad0, add_bc != nullptr, mean, rstd, ga0, be0, TR::to_f32(d2[0]), xh0);
            const float g1 = gn_g<T, SILU>(TR::to_f32(x2[1]), ad1, add_bc != nullptr, mean, rstd, ga1, be1, TR::to_f32(d2[1]), xh1);
            s1 += g0 + g1;
            s2 = __builtin_fmaf(g0, xh0, __builtin_fmaf(g1, xh1, s2));
            lx[(size_t)p * U + j] = xw;
            ld[(size_t)p * U + j] = dw;
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((tid & 63) == 0) { s_w[0][tid >> 6] = s1; s_w[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float t1 = 0.f, t2 = 0.f;
        for (int w = 0; w < GN1_THREADS / 64; ++w) { t1 += s_w[0][w]; t2 += s_w[1][w]; }
        s_c[2] = t1 * inv_n;
        s_c[3] = t2 * inv_n;
    }
    __syncthreads();
    if (!act) return;
    const float m1 = s_c[2], m2 = s_c[3];
    for (int p = slot; p < HW; p += nslot) {
        const uint32_t xw = lx[(size_t)p * U + j], dw = ld[(size_t)p * U + j];
        T x2[2], d2[2], o2[2];
        __builtin_memcpy(x2, &xw, 4); __builtin_memcpy(d2, &dw, 4);
        float xh0, xh1;
        const float g0 = gn_g<T, SILU>(TR::to_f32(x2[0]), ad0, add_bc != nullptr, mean, rstd, ga0, be0, TR::to_f32(d2[0]), xh0);
        const float g1 = gn_g<T, SILU>(TR::to_f32(x2[1]), ad1, add_bc != nullptr, mean, rstd, ga1, be1, TR::to_f32(d2[1]), xh1);
        o2[0] = TR::from_f32(rstd * (g0 - m1 - xh0 * m2));
        o2[1] = TR::from_f32(rstd * (g1 - m1 - xh1 * m2));
        uint32_t o;
        __builtin_memcpy(&o, o2, 4);
        *(uint32_t*)(dx + base + (size_t)p * C) = o;
    }
}

template <typename T, bool SILU>
static void gn_bwd_launch(const void* x, const void* add_bc, int add_ld, const void* gamma, const void* beta, const void* dy,
                          const float* fwd_scratch, float* scratch, int B, int HW, int C, int G, float eps, void* dx, int single_launch,
                          hipStream_t st) {
    const int pix = gn_pix_per_slab(HW);
    const int nslab = (HW + pix - 1) / pix;
    if (gn_single_ok(HW, C, G, single_launch)) {
        const int U = (C / G) >> 1;
        int LP = 1;
        while (LP < U) LP <<= 1;
        static bool attr_set = false;             // per instantiation; the first call of each happens in an eager pass (graphs.py warms up)
        if (!attr_set) {
            hipFuncSetAttribute((const void*)k_gn_bwd_fused<T, SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, GN1_MAX_LDS);
            attr_set = true;
        }
        k_gn_bwd_fused<T, SILU><<<dim3(G, B), GN1_THREADS, (size_t)HW * U * 8, st>>>((const T*)x, (const T*)add_bc, add_ld, (const T*)gamma,
                                                                                   (const T*)beta, (const T*)dy, fwd_scratch, nslab, HW, C,
                                                                                   G, LP, eps, (T*)dx);
        return;
    }
    dim3 sgrid(nslab, B);
    dim3 agrid((unsigned)(((long long)HW * (C >> 3) + 255) / 256), B);
    k_gn_bwd_stats<T, SILU><<<sgrid, 256, 0, st>>>((const T*)x, (const T*)add_bc, add_ld, (const T*)gamma, (const T*)beta, (const T*)dy,
                                                  fwd_scratch, nslab, HW, C, G, pix, eps, scratch);
    k_gn_bwd_apply<T, SILU><<<agrid, 256, 0, st>>>((const T*)x, (const T*)add_bc, add_ld, (const T*)gamma, (const T*)beta, (const T*)dy,
                                                  fwd_scratch, scratch, nslab, HW, C, G, eps, (T*)dx);
}

extern "C" int gd_group_norm_nhwc_bwd(const void* x, const void* add_bc, int add_ld, const void* gamma, const void* beta, const void* dy,
                                      int B, int HW, int C, int G, float eps, int silu, int single_launch, const float* fwd_scratch, float* scratch,
                                      void* dx, int dtype, void* stream) {
    GD_REQUIRE(x && gamma && beta && dy && fwd_scratch && scratch && dx, GD_EINVAL, "gd_group_norm_nhwc_bwd: null pointer");
    GD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && G <= GN_MAX_G && C % G == 0 && (C & 7) == 0 && (C / G >= 8 || C / G == 4) && C <= 8 * GN_MAX_CV, GD_EINVAL,
               "gd_group_norm_nhwc_bwd: unsupported shape B=%d HW=%d C=%d G=%d", B, HW, C, G);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_group_norm_nhwc_bwd: dtype must be f16/bf16");
    GD_REQUIRE(!add_bc || add_ld == 0 || (add_ld >= C && (add_ld & 7) == 0), GD_EINVAL, "gd_group_norm_nhwc_bwd: bad add_ld");
    if (add_ld == 0) add_ld = C;
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) {
        if (silu) gn_bwd_launch<f16_t, true>(x, add_bc, add_ld, gamma, beta, dy, fwd_scratch, scratch, B, HW, C, G, eps, dx, single_launch, st);
        else gn_bwd_launch<f16_t, false>(x, add_bc, add_ld, gamma, beta, dy, fwd_scratch, scratch, B, HW, C, G, eps, dx, single_launch, st);
    } else {
        if (silu) gn_bwd_launch<bf16_t, true>(x, add_bc, add_ld, gamma, beta, dy, fwd_scratch, scratch, B, HW, C, G, eps, dx, single_launch, st);
        else gn_bwd_launch<bf16_t, false>(x, add_bc, add_ld, gamma, beta, dy, fwd_scratch, scratch, B, HW, C, G, eps, dx, single_launch, st);
    }
    GD_CHECK_LAUNCH("gd_group_norm_nhwc_bwd");
    return GD_OK;
}
