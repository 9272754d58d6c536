// R8 — feature losses (background / placement / amodal / smoothness), their gradient, the amodal target and
// the attention-sharing blend.
//
// Replaces background_preservation_loss, object_placement_loss_geodiff, amodal_loss_geodiff
// (GeoDiffuser/utils/attention_processors.py:231-246,283-305), get_smoothness_loss (GeoDiffuser/utils/loss.py:29-41),
// interpolate_from_mask (GeoDiffuser/utils/attention_sharing.py:67-105), smooth_attention_features
// (GeoDiffuser/utils/generic_torch.py:145-154) and the blend at U/attention_processors.py:504,619 — about thirty
// torch launches and a dozen [f,N,D] temporaries per layer in the reference — by one pass each.
// HBM-bound: forward reads eo + ro (+ target) once: 2 * H*N*D * sizeof(T) (+ 4 * H*N*D) bytes.
#include <stdlib.h>
#include "common.hpp"
#include "edit_layer.hpp"

// GaussianSmoothing(kernel_size=5): sigma = 5//2*2/6 = 2/3, exponent -((x-mean)/(2 sigma))^2 (U/generic_torch.py:32-54),
// normalised to sum 1.  The 2-D kernel is the outer product of this 1-D profile (centre weight 0.18102).
//   exp(-2.25) = 0.10539922456186433, exp(-0.5625) = 0.569782824730923, 1
#define GW_SUM (1.0 + 2.0 * 0.569782824730923 + 2.0 * 0.10539922456186433)
__device__ constexpr float c_g1[5] = {(float)(0.10539922456186433 / GW_SUM), (float)(0.569782824730923 / GW_SUM),
                                      (float)(1.0 / GW_SUM), (float)(0.569782824730923 / GW_SUM),
                                      (float)(0.10539922456186433 / GW_SUM)};

// ---- amodal target -----------------------------------------------------------------------------------
template <typename T>
__global__ void k_amodal_interp(const T* __restrict__ eo, const int32_t* __restrict__ nn_idx, const float* __restrict__ nn_w,
                                const float* __restrict__ fg, int H, int N, int D, float* __restrict__ tmp) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)H * N * D) return;
    const int d = (int)(gid % D);
    const long long t = gid / D;
    const int n = (int)(t % N), h = (int)(t / N);
    const T* e = eo + (size_t)h * N * D;
    float r;
    if (fg[n] > 0.5f) {
        r = (float)e[(size_t)n * D + d];
    } else {
        float acc = 0.f, ws = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float w = nn_w[n * 4 + i];
            acc += (float)e[(size_t)nn_idx[n * 4 + i] * D + d] * w;
            ws += w;
        }
        r = acc / (ws + 1e-12f);
    }
    tmp[gid] = r;
}

__global__ void k_gauss5(const float* __restrict__ tmp, int H, int S, int D, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = S * S;
    if (gid >= (long long)H * N * D) return;
    const int d = (int)(gid % D);
    const long long t = gid / D;
    const int n = (int)(t % N), h = (int)(t / N);
    const int y = n / S, x = n - y * S;
    const float* base = tmp + (size_t)h * N * D + d;
    float acc = 0.f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= S) continue;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= S) continue;
            acc += (c_g1[dy + 2] * c_g1[dx + 2]) * base[(size_t)(yy * S + xx) * D];
        }
    }
    out[gid] = acc;
}

// Both steps in one launch (round 4): a workgroup owns an 8 x 8 pixel tile of one head, builds the interpolated / overwritten features of
// the 12 x 12 pixels its 5 x 5 windows reach in LDS (zeros outside the image: the stand-alone kernel skips those taps, adding w * 0 is
// the same sum) and runs the 25 taps from there in k_gauss5's order — the same values bit for bit, without the [H,N,D] f32 round trip
// through HBM and its 25 strided re-reads per output (k_gauss5 alone: 20 us at 64^2 x 5 heads).  D <= 64.
#define AM_T 8
template <typename T>
__global__ void __launch_bounds__(256)
k_amodal_fused(const T* __restrict__ eo, const int32_t* __restrict__ nn_idx, const float* __restrict__ nn_w, const float* __restrict__ fg,
               int S, int D, int tiles_x, float* __restrict__ out) {
    // (first form: one channel per thread and step — 2-byte loads, ~1,000 dependent wave loads per workgroup: 50 us, slower than the
    //  pair it replaced.  Now 8 channels per thread: 16-byte feature loads, 32-byte LDS accesses.)
    using V8 = typename elem_traits<T>::vec8;
    __shared__ __attribute__((aligned(16))) float tile[(AM_T + 4) * (AM_T + 4) * 64];
    const int N = S * S;
    const int h = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * AM_T, x0 = tx * AM_T;
    const T* e = eo + (size_t)h * N * D;
    constexpr int W = AM_T + 4;
    const int D8 = D / 8;
    for (int i = threadIdx.x; i < W * W * D8; i += 256) {
        const int c = i % D8, pp = i / D8;
        const int py = pp / W, px = pp - py * W;
        const int y = y0 - 2 + py, x = x0 - 2 + px;
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = 0.f;
        if (y >= 0 && y < S && x >= 0 && x < S) {
            const int n = y * S + x;
            if (fg[n] > 0.5f) {
                const V8 v = *(const V8*)(e + (size_t)n * D + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (float)v[j];
            } else {
                const f32x4 w4 = *(const f32x4*)(nn_w + n * 4);
                float ws = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const V8 v = *(const V8*)(e + (size_t)nn_idx[n * 4 + k] * D + c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] += (float)v[j] * w4[k];
                    ws += w4[k];
                }
                const float den = ws + 1e-12f;
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = r[j] / den;
            }
        }
        float* dst = tile + pp * D + c * 8;
        *(f32x4*)dst = f32x4{r[0], r[1], r[2], r[3]};
        *(f32x4*)(dst + 4) = f32x4{r[4], r[5], r[6], r[7]};
    }
    __syncthreads();
    for (int i = threadIdx.x; i < AM_T * AM_T * D8; i += 256) {
        const int c = i % D8, pp = i / D8;
        const int oy = pp / AM_T, ox = pp - oy * AM_T;
        const int y = y0 + oy, x = x0 + ox;
        if (y >= S || x >= S) continue;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= S) continue;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= S) continue;
                const float g = c_g1[dy + 2] * c_g1[dx + 2];
                const float* src = tile + ((oy + dy + 2) * W + (ox + dx + 2)) * D + c * 8;
                const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] += g * a[j]; acc[4 + j] += g * b[j]; }
            }
        }
        float* o = out + ((size_t)h * N + (size_t)y * S + x) * D + c * 8;
        *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
}

extern "C" int gd_amodal_target(const void* eo, const int32_t* nn_idx, const float* nn_w, const float* fg,
                                int H, int S, int D, float* tmp, float* target, int dtype, void* stream) {
    GD_REQUIRE(eo && nn_idx && nn_w && fg && tmp && target, GD_EINVAL, "gd_amodal_target: null pointer");
    GD_REQUIRE(H > 0 && S > 0 && D > 0, GD_EINVAL, "gd_amodal_target: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_amodal_target: dtype must be f16/bf16");
    hipStream_t st = as_stream(stream);
    const int N = S * S;
    if (D <= 64 && D % 8 == 0) {                             // one launch; wider heads: the interpolation + blur pair through `tmp`
        const int tiles_x = (S + AM_T - 1) / AM_T;
        dim3 grid(tiles_x * tiles_x, H);
        if (dtype == GD_F16) k_amodal_fused<f16_t><<<grid, 256, 0, st>>>((const f16_t*)eo, nn_idx, nn_w, fg, S, D, tiles_x, target);
        else k_amodal_fused<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)eo, nn_idx, nn_w, fg, S, D, tiles_x, target);
        GD_CHECK_LAUNCH("gd_amodal_target");
        return GD_OK;
    }
    const long long total = (long long)H * N * D;
    const int blocks = (int)((total + 255) / 256);
    if (dtype == GD_F16) k_amodal_interp<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)eo, nn_idx, nn_w, fg, H, N, D, tmp);
    else k_amodal_interp<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)eo, nn_idx, nn_w, fg, H, N, D, tmp);
    k_gauss5<<<blocks, 256, 0, st>>>(tmp, H, S, D, target);
    GD_CHECK_LAUNCH("gd_amodal_target");
    return GD_OK;
}

// ---- losses forward ----------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }


// Per-thread partial sums of the five feature-loss reductions over a grid-stride range of 8-channel chunks (16-byte loads of eo / ro,
// 32-byte loads of the amodal target).  Shared by k_losses_fwd and k_losses_fused: one summation order, identical sums.
template <typename T>
__device__ __forceinline__ void losses_partial(const T* __restrict__ eo, const T* __restrict__ ro, const float* __restrict__ tgt,
                                               const float* __restrict__ m_wo, const float* __restrict__ m_edit,
                                               const float* __restrict__ w_am, const float* __restrict__ m_amodal, int H, int S, int D,
                                               float (&s)[5]) {
    using V8 = typename elem_traits<T>::vec8;
    const int N = S * S, D8 = D / 8;
    const long long total8 = (long long)H * N * D8;
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = 0.f;
    for (long long c8 = (long long)blockIdx.x * blockDim.x + threadIdx.x; c8 < total8; c8 += (long long)gridDim.x * blockDim.x) {
        const long long t = c8 / D8;
        const int n = (int)(t % N);
        const int y = n / S, x = n - y * S;
        const long long gid = c8 * 8;
        const V8 r8 = *(const V8*)(ro + gid), e8 = *(const V8*)(eo + gid);
        const float mw = m_wo[n], me = m_edit[n];
        float ad = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ad += fabsf((float)e8[j] - (float)r8[j]);
        s[0] += ad * mw;
        s[1] += ad * me;
        if (tgt) {
            const f32x4 t0 = *(const f32x4*)(tgt + gid), t1 = *(const f32x4*)(tgt + gid + 4);
            float am = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) am += fabsf(t0[j] - (float)r8[j]) + fabsf(t1[j] - (float)r8[4 + j]);
            s[2] += am * (w_am[n] * m_amodal[n]);
        }
        if (y < S - 1) {
            const V8 d8 = *(const V8*)(ro + gid + (size_t)S * D);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a += fabsf((float)d8[j] - (float)r8[j]);
            s[3] += a;
        }
        if (x < S - 1) {
            const V8 d8 = *(const V8*)(ro + gid + D);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a += fabsf((float)d8[j] - (float)r8[j]);
            s[4] += a;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_losses_fwd(const T* __restrict__ eo, const T* __restrict__ ro, const float* __restrict__ tgt, const float* __restrict__ m_wo,
             const float* __restrict__ m_edit, const float* __restrict__ w_am, const float* __restrict__ m_amodal,
             int H, int S, int D, float* __restrict__ sums /* [gridDim.x, 5] partials */) {
    float sp[5];
    losses_partial<T>(eo, ro, tgt, m_wo, m_edit, w_am, m_amodal, H, S, D, sp);
    float s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
    __shared__ float part[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { part[wave][0] = s0; part[wave][1] = s1; part[wave][2] = s2; part[wave][3] = s3; part[wave][4] = s4; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        sums[(size_t)blockIdx.x * 5 + threadIdx.x] = v;          // per-workgroup partial (summed in fixed order by k_losses_fold)
    }
}

// sums[k] += sum_b partial[b][k], b in a fixed order: wave k folds column k (lane-strided, then the wave tree).  Together with the
// grid-stride loop above (fixed element -> workgroup assignment) the five loss sums are bit-reproducible from run to run.
__global__ void __launch_bounds__(320)
k_losses_fold(const float* __restrict__ partial, int nblocks, float* __restrict__ sums) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * 5 + k];
    s = wave_sum(s);
    if (lane == 0) sums[k] += s;
}

static int losses_fwd_blocks(int H, int S, int D) {
    const long long total8 = (long long)H * S * S * D / 8;
    int blocks = (int)((total8 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;              // one workgroup per CU: 256 arrival tickets, a 256-row fold in gd_edit_losses_fused's tail
    return blocks;
}

extern "C" size_t gd_edit_losses_fwd_workspace_bytes(int H, int S, int D) {
    return (size_t)losses_fwd_blocks(H, S, D) * 5 * sizeof(float);
}

// ---- losses backward ---------------------------------------------------------------------------------

// (the kernel is k_losses_bwd_rowdot below: the loss backward with the removal row dots as optional extra workgroups)

// ============================================================================================================================
// Fused launches of one hooked optimisation-pass layer (ABI 4, see edit_layer.hpp)
// ============================================================================================================================

// row merge + blend: one pass over the layer's [H,N,D] rows, 8 elements per thread
template <typename T>
__global__ void k_blend_merge(const u32x4* __restrict__ base, const u32x4* __restrict__ act, const int32_t* __restrict__ pos,
                              const u32x4* __restrict__ ro, const float* __restrict__ m, int H, int N, int R, int D8,
                              u32x4* __restrict__ eo_out, u32x4* __restrict__ out) {
#pragma clang fp contract(off)      // each product and the sum are rounded to the tensor dtype: no fused multiply-add across the roundings
    using V8 = typename elem_traits<T>::vec8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;                   // one 16-byte chunk of 8 elements
    if (gid >= (long long)H * N * D8) return;
    const int c = (int)(gid % D8);
    const long long hn = gid / D8;
    const int n = (int)(hn % N), h = (int)(hn / N);
    const int p = act ? pos[n] : -1;
    const u32x4 e = p >= 0 ? act[((long long)h * R + p) * D8 + c] : base[gid];
    if (eo_out) eo_out[gid] = e;
    if (out) {
        // op-by-op in the tensor dtype, as torch evaluates  a*m + b*(1-m)  on 16-bit tensors (k_blend)
        const V8 a8 = __builtin_bit_cast(V8, e), b8 = __builtin_bit_cast(V8, ro[gid]);
        const float mm = (float)(T)m[n];
        const float om = (float)(T)(1.0f - mm);
        V8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t1 = (float)(T)((float)a8[j] * mm);
            const float t2 = (float)(T)((float)b8[j] * om);
            o8[j] = (T)(t1 + t2);
        }
        out[gid] = __builtin_bit_cast(u32x4, o8);
    }
}

extern "C" int gd_blend_merge(const void* base, const void* act, const int32_t* pos, const void* ro, const float* m, int H, int N, int R, int D,
                              void* eo_out, void* out, int dtype, void* stream) {
    GD_REQUIRE(base && (eo_out || out), GD_EINVAL, "gd_blend_merge: null pointer");
    GD_REQUIRE(!act || (pos && R > 0), GD_EINVAL, "gd_blend_merge: act needs pos and R");
    GD_REQUIRE(!out || (ro && m), GD_EINVAL, "gd_blend_merge: out needs ro and m");
    GD_REQUIRE(H > 0 && N > 0 && D > 0 && D % 8 == 0, GD_EINVAL, "gd_blend_merge: bad sizes (D must be a multiple of 8)");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_blend_merge: dtype must be f16/bf16");
    const long long total = (long long)H * N * (D / 8);
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_blend_merge<f16_t><<<blocks, 256, 0, st>>>((const u32x4*)base, (const u32x4*)act, pos, (const u32x4*)ro, m, H, N, R, D / 8, (u32x4*)eo_out, (u32x4*)out);
    else
        k_blend_merge<bf16_t><<<blocks, 256, 0, st>>>((const u32x4*)base, (const u32x4*)act, pos, (const u32x4*)ro, m, H, N, R, D / 8, (u32x4*)eo_out, (u32x4*)out);
    GD_CHECK_LAUNCH("gd_blend_merge");
    return GD_OK;
}

// ---- the layer's layout boundary: token-major [B, rows, heads*D] <-> head-major [B*heads, rows, D], one launch each way ----------------
// one 16-byte chunk (8 x 16-bit) per thread, destination-linear (stores coalesced; loads are 128-byte row pieces)
struct HeadsSplitCum { long long c[GD_HEADS_SPLIT_MAX]; };
__global__ void k_heads_split(const gd_heads_split_t a, const HeadsSplitCum cum) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= cum.c[GD_HEADS_SPLIT_MAX - 1]) return;
    int i = 0;
#pragma unroll
    for (int j = 0; j < GD_HEADS_SPLIT_MAX - 1; ++j) i += gid >= cum.c[j] ? 1 : 0;
    const long long l = gid - (i == 0 ? 0 : cum.c[i - 1]);
    const int D8 = a.D >> 3, rows = a.rows[i], heads = a.heads;
    const int c = (int)(l % D8);
    long long t = l / D8;
    const int r = (int)(t % rows); t /= rows;
    const int h = (int)(t % heads);
    const long long b = t / heads;
    const u32x4* __restrict__ src = (const u32x4*)a.src[i];
    u32x4* __restrict__ dst = (u32x4*)a.dst[i];
    dst[l] = src[((b * rows + r) * heads + h) * D8 + c];
}

extern "C" int gd_heads_split(const gd_heads_split_t* a, int dtype, void* stream) {
    GD_REQUIRE(a && a->n >= 1 && a->n <= GD_HEADS_SPLIT_MAX, GD_EINVAL, "gd_heads_split: 1..%d tensors", GD_HEADS_SPLIT_MAX);
    GD_REQUIRE(a->B > 0 && a->heads > 0 && a->D > 0 && a->D % 8 == 0, GD_EINVAL, "gd_heads_split: bad sizes (D must be a multiple of 8)");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_heads_split: dtype must be f16/bf16");
    HeadsSplitCum cum;
    long long tot = 0;
    for (int i = 0; i < GD_HEADS_SPLIT_MAX; ++i) {
        if (i < a->n) {
            GD_REQUIRE(a->src[i] && a->dst[i] && a->rows[i] > 0, GD_EINVAL, "gd_heads_split: null pointer / no rows");
            tot += (long long)a->B * a->rows[i] * a->heads * (a->D / 8);
        }
        cum.c[i] = tot;
    }
    const int blocks = (int)((tot + 255) / 256);
    k_heads_split<<<blocks, 256, 0, as_stream(stream)>>>(*a, cum);
    GD_CHECK_LAUNCH("gd_heads_split");
    return GD_OK;
}

template <typename T, bool SRC32>
__global__ void k_heads_merge(const gd_heads_merge_t a) {
#pragma clang fp contract(off)      // each product and the sum are rounded to the tensor dtype: no fused multiply-add across the roundings
    using V8 = typename elem_traits<T>::vec8;
    const int D8 = a.D >> 3, rows = a.rows, heads = a.heads;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)a.B * rows * heads * D8) return;
    const int c = (int)(gid % D8);
    long long t = gid / D8;
    const int h = (int)(t % heads); t /= heads;
    const int r = (int)(t % rows);
    const int b = (int)(t / rows);
    const void* sp = a.src[b];
    const long long si = ((long long)h * rows + r) * D8 + c;
    V8 o8;
    if (!sp) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (T)0.f;
    } else if (SRC32) {
        const f32x4 lo = ((const f32x4*)sp)[2 * si], hi = ((const f32x4*)sp)[2 * si + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { o8[j] = (T)lo[j]; o8[4 + j] = (T)hi[j]; }
    } else {
        o8 = __builtin_bit_cast(V8, ((const u32x4*)sp)[si]);
        if (a.blend_b[b]) {              // a*m + b*(1-m), op by op in the tensor dtype
            const V8 b8 = __builtin_bit_cast(V8, ((const u32x4*)a.blend_b[b])[si]);
            const float mm = (float)(T)a.m[b][r];
            const float om = (float)(T)(1.0f - mm);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t1 = (float)(T)((float)o8[j] * mm);
                const float t2 = (float)(T)((float)b8[j] * om);
                o8[j] = (T)(t1 + t2);
            }
        }
    }
    ((u32x4*)a.out)[gid] = __builtin_bit_cast(u32x4, o8);
}

extern "C" int gd_heads_merge(const gd_heads_merge_t* a, int dtype, void* stream) {
    GD_REQUIRE(a && a->out, GD_EINVAL, "gd_heads_merge: null pointer");
    GD_REQUIRE(a->B >= 1 && a->B <= GD_HEADS_MERGE_MAX_ROWS && a->rows > 0 && a->heads > 0 && a->D > 0 && a->D % 8 == 0, GD_EINVAL,
               "gd_heads_merge: bad sizes (B <= %d, D a multiple of 8)", GD_HEADS_MERGE_MAX_ROWS);
    for (int b = 0; b < a->B; ++b)
        GD_REQUIRE(!a->blend_b[b] || (!a->src_f32 && a->m[b] && a->src[b]), GD_EINVAL, "gd_heads_merge: row %d: the blend needs 16-bit sources and a mask", b);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_heads_merge: dtype must be f16/bf16");
    const long long tot = (long long)a->B * a->rows * a->heads * (a->D / 8);
    const int blocks = (int)((tot + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) {
        if (a->src_f32) k_heads_merge<f16_t, true><<<blocks, 256, 0, st>>>(*a);
        else k_heads_merge<f16_t, false><<<blocks, 256, 0, st>>>(*a);
    } else {
        if (a->src_f32) k_heads_merge<bf16_t, true><<<blocks, 256, 0, st>>>(*a);
        else k_heads_merge<bf16_t, false><<<blocks, 256, 0, st>>>(*a);
    }
    GD_CHECK_LAUNCH("gd_heads_merge");
    return GD_OK;
}

// k_losses_fwd's grid; the workgroup whose ticket comes last runs gd_removal_loss_reduce's body, k_losses_fold's sums and gd_loss_assemble
template <typename T>
__global__ void __launch_bounds__(256)
k_losses_fused(const gd_edit_losses_t a) {
    const int S = a.S, H = a.H;
    float sp[5];
    losses_partial<T>((const T*)a.eo, (const T*)a.ro, a.tgt, a.m_wo, a.m_edit, a.w_am, a.m_amodal, H, S, a.D, sp);
    float s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
    __shared__ float part[4][5];
    __shared__ float rpart[4];
    __shared__ float sums[5];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { part[wave][0] = s0; part[wave][1] = s1; part[wave][2] = s2; part[wave][3] = s3; part[wave][4] = s4; }
    __syncthreads();
    if (threadIdx.x < 5)
        a.workspace[(size_t)blockIdx.x * 5 + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: this workgroup's partial is visible at agent scope before its ticket; acquire: the last arriver sees every partial
        const int old = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                            // (per-thread acquire for the loads below)
    // ---- tail: one workgroup ----
    float rm = 0.f;
    if (a.best) rm = removal_reduce_body(a.best, a.rows, a.n_valid, H, a.R, S, a.p_in, a.j_in, a.p_wo, a.j_wo, a.wgt, rpart);
    // k_losses_fold's order: column k lane-strided over the workgroups, then the wave tree (waves 0..3 take columns 0..3, wave 0 also 4)
    const int nblocks = (int)gridDim.x;
    for (int k = wave; k < 5; k += 4) {
        float s = 0.f;
        for (int b = lane; b < nblocks; b += 64) s += a.workspace[(size_t)b * 5 + k];
        s = wave_sum(s);
        if (lane == 0) sums[k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_assemble_body(sums, rm, a.inv5, a.inv_rm, a.wv, a.inv5_bwd, a.use_amodal, a.out12);
        if (a.log_acc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.log_acc[k] = a.log_acc[k] + a.out12[k];
        }
        if (a.loss_out) a.loss_out[0] = (a.loss_in ? a.loss_in[0] : 0.0f) + a.out12[5];
        __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // the next launch finds zero
    }
}

// wv == NULL: the reductions only (two launches, out12[0..4] = the sums); otherwise everything up to the assembled loss in one launch
extern "C" int gd_edit_losses_fwd(const gd_edit_losses_t* a, int dtype, void* stream) {
    GD_REQUIRE(a && a->eo && a->ro && a->m_wo && a->m_edit && a->workspace && a->out12, GD_EINVAL, "gd_edit_losses_fwd: null pointer");
    GD_REQUIRE(!a->tgt || (a->w_am && a->m_amodal), GD_EINVAL, "gd_edit_losses_fwd: tgt needs w_am and m_amodal");
    GD_REQUIRE(a->H > 0 && a->S > 0 && a->D > 0 && a->D % 8 == 0, GD_EINVAL, "gd_edit_losses_fwd: bad sizes (D must be a multiple of 8)");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_edit_losses_fwd: dtype must be f16/bf16");
    const int blocks = losses_fwd_blocks(a->H, a->S, a->D);
    hipStream_t st = as_stream(stream);
    if (!a->wv) {
        if (dtype == GD_F16)
            k_losses_fwd<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)a->eo, (const f16_t*)a->ro, a->tgt, a->m_wo, a->m_edit, a->w_am, a->m_amodal,
                                                        a->H, a->S, a->D, a->workspace);
        else
            k_losses_fwd<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)a->eo, (const bf16_t*)a->ro, a->tgt, a->m_wo, a->m_edit, a->w_am, a->m_amodal,
                                                         a->H, a->S, a->D, a->workspace);
        k_losses_fold<<<1, 320, 0, st>>>(a->workspace, blocks, a->out12);
        GD_CHECK_LAUNCH("gd_edit_losses_fwd");
        return GD_OK;
    }
    GD_REQUIRE(a->ticket && a->inv5 && a->inv_rm && a->inv5_bwd, GD_EINVAL, "gd_edit_losses_fwd: the assembled form needs ticket, inv5, inv_rm, inv5_bwd");
    GD_REQUIRE(!a->best || (a->rows && a->p_in && a->j_in && a->p_wo && a->j_wo && a->wgt && a->R > 0), GD_EINVAL,
               "gd_edit_losses_fwd: best needs rows, R and the five aux outputs");
    if (dtype == GD_F16) k_losses_fused<f16_t><<<blocks, 256, 0, st>>>(*a);
    else k_losses_fused<bf16_t><<<blocks, 256, 0, st>>>(*a);
    GD_CHECK_LAUNCH("gd_edit_losses_fwd");
    return GD_OK;
}

// d(loss)/d(ro) per element, followed (blockIdx >= nb_loss) by k_removal_rowdot's workgroups (4 rows each): independent work, one launch
template <typename T>
__global__ void __launch_bounds__(256)
k_losses_bwd_rowdot(const T* __restrict__ eo, const T* __restrict__ ro, const float* __restrict__ tgt,
                    const float* __restrict__ m_wo, const float* __restrict__ m_edit, const float* __restrict__ w_am,
                    const float* __restrict__ m_amodal, const T* __restrict__ gout, const float* __restrict__ c, const float* __restrict__ gscale, int blend,
                    int H, int S, int D, T* __restrict__ dro, int nb_loss, const gd_removal_bwd_t rm) {
    if ((int)blockIdx.x >= nb_loss) {
        removal_rowdot_body<T>(rm, ((int)blockIdx.x - nb_loss) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), rm.workspace);
        return;
    }
    const int N = S * S;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)H * N * D) return;
    const long long t = gid / D;
    const int n = (int)(t % N);
    const int y = n / S, x = n - y * S;
    const float r = (float)ro[gid], e = (float)eo[gid];
    const float sg = -sgn(e - r);
    const float me = m_edit[n];
    float g = c[0] * sg * m_wo[n] + c[1] * sg * me;
    if (tgt) g += c[2] * (-sgn(tgt[gid] - r)) * w_am[n] * m_amodal[n];
    float gs = 0.f;
    if (y < S - 1) gs -= sgn((float)ro[gid + (size_t)S * D] - r);
    if (y > 0) gs += sgn(r - (float)ro[gid - (size_t)S * D]);
    g += c[3] * gs;
    gs = 0.f;
    if (x < S - 1) gs -= sgn((float)ro[gid + D] - r);
    if (x > 0) gs += sgn(r - (float)ro[gid - D]);
    g += c[4] * gs;
    if (gscale) g *= gscale[0];
    if (gout) {       // blend bit 1: gout is the token-major row [N, H*D] (the layer's boundary, gd_heads_split of it folded in here)
        const long long gi = (blend & 2) ? (((long long)n * H + (int)(t / N)) * D + (int)(gid - t * D)) : gid;
        g += (float)gout[gi] * ((blend & 1) ? (1.0f - me) : 1.0f);
    }
    dro[gid] = (T)g;
}

extern "C" int gd_edit_losses_bwd(const void* eo, const void* ro, const float* tgt, const float* m_wo, const float* m_edit,
                                  const float* w_am, const float* m_amodal, const void* gout, const float* coef_dev, const float* gscale_dev,
                                  int blend, int H, int S, int D, void* dro, const gd_removal_bwd_t* rm, int dtype, void* stream) {
    GD_REQUIRE(eo && ro && m_wo && m_edit && coef_dev && dro, GD_EINVAL, "gd_edit_losses_bwd: null pointer");
    GD_REQUIRE(!tgt || (w_am && m_amodal), GD_EINVAL, "gd_edit_losses_bwd: tgt needs w_am and m_amodal");
    GD_REQUIRE(H > 0 && S > 0 && D > 0, GD_EINVAL, "gd_edit_losses_bwd: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_edit_losses_bwd: dtype must be f16/bf16");
    gd_removal_bwd_t none;
    memset(&none, 0, sizeof(none));
    if (rm) {
        GD_REQUIRE(rm->Pe && rm->Pb && rm->p_in && rm->j_in && rm->p_wo && rm->j_wo && rm->wgt && rm->m_inp && rm->m_wo && rm->workspace, GD_EINVAL,
                   "gd_edit_losses_bwd: removal arguments: null pointer");
        GD_REQUIRE(rm->H > 0 && rm->R > 0 && rm->N > 0 && rm->M > 0 && rm->Mpad >= rm->M, GD_EINVAL, "gd_edit_losses_bwd: removal arguments: bad sizes");
    }
    const long long total = (long long)H * S * S * D;
    const int nb_loss = (int)((total + 255) / 256);
    const int nb_dot = rm ? (rm->H * rm->R + 3) / 4 : 0;          // the removal backward's row dots: extra workgroups of the same launch
    const gd_removal_bwd_t& rr = rm ? *rm : none;
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_losses_bwd_rowdot<f16_t><<<nb_loss + nb_dot, 256, 0, st>>>((const f16_t*)eo, (const f16_t*)ro, tgt, m_wo, m_edit, w_am, m_amodal,
                                                                   (const f16_t*)gout, coef_dev, gscale_dev, blend, H, S, D, (f16_t*)dro, nb_loss, rr);
    else
        k_losses_bwd_rowdot<bf16_t><<<nb_loss + nb_dot, 256, 0, st>>>((const bf16_t*)eo, (const bf16_t*)ro, tgt, m_wo, m_edit, w_am, m_amodal,
                                                                    (const bf16_t*)gout, coef_dev, gscale_dev, blend, H, S, D, (bf16_t*)dro, nb_loss, rr);
    GD_CHECK_LAUNCH("gd_edit_losses_bwd");
    return GD_OK;
}

// dq16[h, n, :] = T( sum_c dq_part[c][h, n, :]  +  (n a live inpaint row ? sum_c rm_part[c][h, slot, :] : 0) ), c ascending, one rounding.
// 4 elements per thread.  The removal partials lie at rm_workspace + H*R (gd_removal_bwd's layout: row dots first).
// part == NULL (the dq kernel did not split its key range and wrote dq directly): only the live inpaint rows are touched, T(float(dq) + s).
template <typename T>
__global__ void k_edit_dq_fold(const float* __restrict__ part, int kchunks, long long cstride4, int BH, int N, int D, const float* __restrict__ rm_part,
                               int msplit, int R, const int32_t* __restrict__ inp_pos, const float* __restrict__ wgt, T* dq) {
    using TR = elem_traits<T>;
    const long long n4 = (long long)BH * N * D / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (part) {
        acc = *(const f32x4*)(part + i * 4);
        for (int c = 1; c < kchunks; ++c) {
            const f32x4 p = *(const f32x4*)(part + ((long long)c * cstride4 + i) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += p[j];
        }
    }
    bool touched = part != nullptr;
    if (rm_part) {
        const int d4 = D / 4;
        const long long hn = i / d4;
        const int dq4 = (int)(i - hn * d4);
        const int n = (int)(hn % N), h = (int)(hn / N);
        const int slot = inp_pos[n];
        if (slot >= 0 && wgt[(size_t)h * R + slot] != 0.f) {
            if (!part) {                 // unsplit dq kernel: the 16-bit gradient is the first term (gd_removal_bwd's in-place add)
                const typename TR::vec4 cur = *(const typename TR::vec4*)(dq + i * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = TR::to_f32(cur[j]);
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < msplit; ++c) {
                const f32x4 p = *(const f32x4*)(rm_part + (((size_t)c * BH + h) * R + slot) * D + dq4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += p[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += s[j];
            touched = true;
        }
    }
    if (!touched) return;
    typename TR::vec4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(acc[j]);
    *(typename TR::vec4*)(dq + i * 4) = w;
}

extern "C" int gd_edit_dq_fold(const float* dq_part, int kchunks, int64_t chunk_stride, int BH, int N, int D, const float* rm_workspace, int M, int R,
                               const int32_t* inp_pos, const float* wgt, void* dq16, int dtype, void* stream) {
    GD_REQUIRE(dq16 && kchunks >= 1 && (dq_part || rm_workspace), GD_EINVAL, "gd_edit_dq_fold: null pointer / kchunks < 1");
    GD_REQUIRE(chunk_stride == 0 || (chunk_stride >= (int64_t)BH * N * D && chunk_stride % 4 == 0), GD_EINVAL,
               "gd_edit_dq_fold: chunk_stride must be 0 (= BH*N*D) or a multiple of 4 >= BH*N*D");
    GD_REQUIRE(BH > 0 && N > 0 && D > 0 && D % 4 == 0, GD_EINVAL, "gd_edit_dq_fold: bad sizes");
    GD_REQUIRE(!rm_workspace || (inp_pos && wgt && R > 0 && M > 0), GD_EINVAL, "gd_edit_dq_fold: removal partials need inp_pos, wgt, R, M");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_edit_dq_fold: dtype must be f16/bf16");
    const int msplit = rm_workspace ? (M + 511) / 512 : 0;           // RM_MCH keys per chunk (removal.hip)
    const float* rm_part = rm_workspace ? rm_workspace + (size_t)BH * R : nullptr;
    const long long n4 = (long long)BH * N * D / 4;
    const int blocks = (int)((n4 + 255) / 256);
    hipStream_t st = as_stream(stream);
    const long long cs4 = (chunk_stride ? chunk_stride : (int64_t)BH * N * D) / 4;
    if (dtype == GD_F16) k_edit_dq_fold<f16_t><<<blocks, 256, 0, st>>>(dq_part, kchunks, cs4, BH, N, D, rm_part, msplit, R, inp_pos, wgt, (f16_t*)dq16);
    else k_edit_dq_fold<bf16_t><<<blocks, 256, 0, st>>>(dq_part, kchunks, cs4, BH, N, D, rm_part, msplit, R, inp_pos, wgt, (bf16_t*)dq16);
    GD_CHECK_LAUNCH("gd_edit_dq_fold");
    return GD_OK;
}

// ---- 4-nearest-foreground table ----------------------------------------------------------------------------
// The mask-only part of interpolate_from_mask (GeoDiffuser/utils/attention_sharing.py:81-83,103): for every pixel the
// k = 4 columns with the largest 1/(dist*256 + 1e5*background + 1e-4).  torch.topk leaves the choice among equal
// values to the implementation; this kernel fixes it to (value descending, index ascending) — the set the CPU
// implementation keeps — by ordering on the exact integer squared pixel distance.  Once per edit per resolution.
// Candidates: with >= 4 foreground pixels (always, on the live path) only foreground pixels can be among the four best keys, so every
// workgroup first compacts the foreground indices into LDS (ascending; ~500 of 4096 at 64^2) and each pixel scans that list (LDS
// broadcast reads) instead of all N pixels from global memory: 630 -> ~30 us at 64^2.  Fewer than 4 foreground pixels, or a map too
// large for the LDS list (S > 128): the full scan.  Same keys, same order: the table is unchanged bit for bit.
#define NN_LIST_MAX (128 * 128)
__global__ void __launch_bounds__(256)
k_nn_table(const float* __restrict__ fg, int S, int32_t* __restrict__ nn_idx, float* __restrict__ nn_w, float* __restrict__ w_dist) {
    __shared__ int32_t list[NN_LIST_MAX];
    __shared__ int wcnt[4];
    __shared__ int total;
    const int N = S * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int F = -1;
    if (N <= NN_LIST_MAX) {
        if (tid == 0) total = 0;
        __syncthreads();
        for (int base = 0; base < N; base += 256) {
            const int i = base + tid;
            const bool is_fg = i < N && fg[i] > 0.5f;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(is_fg);
            if (lane == 0) wcnt[wave] = __builtin_popcountll(b);
            __syncthreads();
            int off = total;
            for (int w2 = 0; w2 < wave; ++w2) off += wcnt[w2];
            if (is_fg) list[off + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
            if (tid == 0) total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        F = total;
    }
    const int n = blockIdx.x * blockDim.x + tid;
    if (n >= N) return;
    const int y = n / S, x = n - y * S;
    // keys: (background << 60) | (r2 << 28) | j   — smaller is better
    unsigned long long best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    if (F >= 4) {
        for (int e = 0; e < F; ++e) {
            const int j = list[e];
            const int yj = j / S, xj = j - yj * S;
            const unsigned long long r2 = (unsigned long long)((x - xj) * (x - xj) + (y - yj) * (y - yj));
            unsigned long long key = (r2 << 28) | (unsigned long long)j;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (key < best[s]) { const unsigned long long t = best[s]; best[s] = key; key = t; }
        }
    } else {
        for (int j = 0; j < N; ++j) {
            const int yj = j / S, xj = j - yj * S;
            const unsigned long long r2 = (unsigned long long)((x - xj) * (x - xj) + (y - yj) * (y - yj));
            const unsigned long long bg = fg[j] > 0.5f ? 0ull : 1ull;
            unsigned long long key = (bg << 60) | (r2 << 28) | (unsigned long long)j;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (key < best[s]) { const unsigned long long t = best[s]; best[s] = key; key = t; }
        }
    }
    float wmax = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int j = (int)(best[s] & 0xFFFFFFFull);
        const int yj = j / S, xj = j - yj * S;
        const float dx = (float)(2 * (x - xj)) / (float)S, dy = (float)(2 * (y - yj)) / (float)S;
        const float dist = sqrtf(dx * dx + dy * dy + 1e-12f);
        const float dnew = dist * 512.0f / 2.0f + 100000.0f * (fg[j] > 0.5f ? 0.0f : 1.0f);
        const float inv = 1.0f / (dnew + 1e-4f);
        nn_idx[n * 4 + s] = j;
        nn_w[n * 4 + s] = inv;
        wmax = fmaxf(wmax, inv);
    }
    w_dist[n] = expf(-(1.0f / wmax) / 5.0f);
}

extern "C" int gd_nn_table(const float* fg, int S, int32_t* nn_idx, float* nn_w, float* w_dist, void* stream) {
    GD_REQUIRE(fg && nn_idx && nn_w && w_dist, GD_EINVAL, "gd_nn_table: null pointer");
    GD_REQUIRE(S > 0 && S <= 16384 && (long long)S * S < (1ll << 28), GD_EINVAL, "gd_nn_table: bad S");
    const int N = S * S;
    k_nn_table<<<(N + 255) / 256, 256, 0, as_stream(stream)>>>(fg, S, nn_idx, nn_w, w_dist);
    GD_CHECK_LAUNCH("gd_nn_table");
    return GD_OK;
}

// ---------------------------------------------------------------------------------------------------
// R8 — the few-float arithmetic between the loss reductions and the backward of a hooked layer, one launch instead of ~11 scalar ones
// (normalise the five sums, pick the terms, weight them, and the coefficients d(loss)/d(sum_i) the backward kernels read):
//   t = sums * inv5;  l_rm = rm * inv_rm
//   terms = [t0 (sim), t1 (movement), l_rm (removal), t3 + t4 (smoothness), use_amodal ? t2 : t1 * 0]     (U/attention_processors.py:
//   loss  = sum_i terms_i * wv_i   (ascending i)                                                            231-305, 479-480, 596-597)
//   coefs = wv[{0, 1, 4, 3, 3}] * inv5_bwd;  rm_coef = wv[2] * inv_rm
// out[12] = terms[0:5], loss[5], coefs[6:11], rm_coef[11].  All operands stay on the device (the adaptive weights wv change between
// replays of a captured optimisation pass).
// ---------------------------------------------------------------------------------------------------
__global__ void k_loss_assemble(const float* __restrict__ sums, const float* __restrict__ rm, const float* __restrict__ inv5,
                                const float* __restrict__ inv_rm, const float* __restrict__ wv, const float* __restrict__ inv5_bwd,
                                int use_amodal, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    float t[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = sums[i] * inv5[i];
    const float l_rm = rm[0] * inv_rm[0];
    float terms[5] = {t[0], t[1], l_rm, t[3] + t[4], use_amodal ? t[2] : t[1] * 0.0f};
    float loss = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { out[i] = terms[i]; loss += terms[i] * wv[i]; }
    out[5] = loss;
    const int perm[5] = {0, 1, 4, 3, 3};
#pragma unroll
    for (int i = 0; i < 5; ++i) out[6 + i] = wv[perm[i]] * inv5_bwd[i];
    out[11] = wv[2] * inv_rm[0];
}

extern "C" int gd_loss_assemble(const float* sums, const float* rm, const float* inv5, const float* inv_rm, const float* wv,
                                const float* inv5_bwd, int use_amodal, float* out12, void* stream) {
    GD_REQUIRE(sums && rm && inv5 && inv_rm && wv && inv5_bwd && out12, GD_EINVAL, "gd_loss_assemble: null pointer");
    k_loss_assemble<<<1, 64, 0, as_stream(stream)>>>(sums, rm, inv5, inv_rm, wv, inv5_bwd, use_amodal, out12);
    GD_CHECK_LAUNCH("gd_loss_assemble");
    return GD_OK;
}
