// R8 — feature losses (background / placement / amodal / smoothness), their gradient, the amodal target and
// the attention-sharing blend.
//
// Replaces background_preservation_loss, object_placement_loss_geodiff, amodal_loss_geodiff
// (GeoDiffuser/utils/attention_processors.py:231-246,283-305), get_smoothness_loss (GeoDiffuser/utils/loss.py:29-41),
// interpolate_from_mask (GeoDiffuser/utils/attention_sharing.py:67-105), smooth_attention_features
// (GeoDiffuser/utils/generic_torch.py:145-154) and the blend at U/attention_processors.py:504,619 — about thirty
// torch launches and a dozen [f,N,D] temporaries per layer in the reference — by one pass each.
// HBM-bound: forward reads eo + ro (+ target) once: 2 * H*N*D * sizeof(T) (+ 4 * H*N*D) bytes.
#include "common.hpp"

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

extern "C" int gd_amodal_target(const void* eo, const int32_t* nn_idx, const float* nn_w, const float* fg,
                                int H, int S, int D, float* tmp, float* target, int dtype, void* stream) {
    GD_REQUIRE(eo && nn_idx && nn_w && fg && tmp && target, GD_EINVAL, "gd_amodal_target: null pointer");
    GD_REQUIRE(H > 0 && S > 0 && D > 0, GD_EINVAL, "gd_amodal_target: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_amodal_target: dtype must be f16/bf16");
    hipStream_t st = as_stream(stream);
    const int N = S * S;
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

template <typename T>
__global__ void __launch_bounds__(256)
k_losses_fwd(const T* __restrict__ eo, const T* __restrict__ ro, const float* __restrict__ tgt, const float* __restrict__ m_wo,
             const float* __restrict__ m_edit, const float* __restrict__ w_am, const float* __restrict__ m_amodal,
             int H, int S, int D, float* __restrict__ sums /* [gridDim.x, 5] partials */) {
    const int N = S * S;
    const long long total = (long long)H * N * D;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const long long t = gid / D;
        const int n = (int)(t % N);
        const int y = n / S, x = n - y * S;
        const float r = (float)ro[gid], e = (float)eo[gid];
        const float ad = fabsf(e - r);
        s0 += ad * m_wo[n];
        s1 += ad * m_edit[n];
        if (tgt) s2 += fabsf(tgt[gid] - r) * w_am[n] * m_amodal[n];
        if (y < S - 1) s3 += fabsf((float)ro[gid + (size_t)S * D] - r);
        if (x < S - 1) s4 += fabsf((float)ro[gid + D] - r);
    }
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
    const long long total = (long long)H * S * S * D;
    int blocks = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    return blocks;
}

extern "C" size_t gd_edit_losses_fwd_workspace_bytes(int H, int S, int D) {
    return (size_t)losses_fwd_blocks(H, S, D) * 5 * sizeof(float);
}

extern "C" int gd_edit_losses_fwd(const void* eo, const void* ro, const float* tgt, const float* m_wo, const float* m_edit,
                                  const float* w_am, const float* m_amodal, int H, int S, int D, float* sums, float* workspace,
                                  int dtype, void* stream) {
    GD_REQUIRE(eo && ro && m_wo && m_edit && sums && workspace, GD_EINVAL, "gd_edit_losses_fwd: null pointer");
    GD_REQUIRE(!tgt || (w_am && m_amodal), GD_EINVAL, "gd_edit_losses_fwd: tgt needs w_am and m_amodal");
    GD_REQUIRE(H > 0 && S > 0 && D > 0, GD_EINVAL, "gd_edit_losses_fwd: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_edit_losses_fwd: dtype must be f16/bf16");
    const int blocks = losses_fwd_blocks(H, S, D);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_losses_fwd<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)eo, (const f16_t*)ro, tgt, m_wo, m_edit, w_am, m_amodal, H, S, D, workspace);
    else
        k_losses_fwd<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)eo, (const bf16_t*)ro, tgt, m_wo, m_edit, w_am, m_amodal, H, S, D, workspace);
    k_losses_fold<<<1, 320, 0, st>>>(workspace, blocks, sums);
    GD_CHECK_LAUNCH("gd_edit_losses_fwd");
    return GD_OK;
}

// ---- losses backward ---------------------------------------------------------------------------------

template <typename T>
__global__ void k_losses_bwd(const T* __restrict__ eo, const T* __restrict__ ro, const float* __restrict__ tgt,
                             const float* __restrict__ m_wo, const float* __restrict__ m_edit, const float* __restrict__ w_am,
                             const float* __restrict__ m_amodal, const T* __restrict__ gout, const float* __restrict__ c, const float* __restrict__ gscale, int blend,
                             int H, int S, int D, T* __restrict__ dro) {
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
    if (gout) g += (float)gout[gid] * (blend ? (1.0f - me) : 1.0f);
    dro[gid] = (T)g;
}

extern "C" int gd_edit_losses_bwd(const void* eo, const void* ro, const float* tgt, const float* m_wo, const float* m_edit,
                                  const float* w_am, const float* m_amodal, const void* gout, const float* coef_dev, const float* gscale_dev,
                                  int blend, int H, int S, int D, void* dro, int dtype, void* stream) {
    GD_REQUIRE(eo && ro && m_wo && m_edit && coef_dev && dro, GD_EINVAL, "gd_edit_losses_bwd: null pointer");
    GD_REQUIRE(!tgt || (w_am && m_amodal), GD_EINVAL, "gd_edit_losses_bwd: tgt needs w_am and m_amodal");
    GD_REQUIRE(H > 0 && S > 0 && D > 0, GD_EINVAL, "gd_edit_losses_bwd: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_edit_losses_bwd: dtype must be f16/bf16");
    const long long total = (long long)H * S * S * D;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16)
        k_losses_bwd<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)eo, (const f16_t*)ro, tgt, m_wo, m_edit, w_am, m_amodal,
                                                    (const f16_t*)gout, coef_dev, gscale_dev, blend, H, S, D, (f16_t*)dro);
    else
        k_losses_bwd<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)eo, (const bf16_t*)ro, tgt, m_wo, m_edit, w_am, m_amodal,
                                                     (const bf16_t*)gout, coef_dev, gscale_dev, blend, H, S, D, (bf16_t*)dro);
    GD_CHECK_LAUNCH("gd_edit_losses_bwd");
    return GD_OK;
}

// ---- blend ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_blend(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ m, int H, int N, int D,
                        T* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)H * N * D) return;
    const int n = (int)((gid / D) % N);
    // op-by-op in the tensor dtype, as torch evaluates  a*m + b*(1-m)  on 16-bit tensors
    const float mm = (float)(T)m[n];
    const float om = (float)(T)(1.0f - mm);
    const float t1 = (float)(T)((float)a[gid] * mm);
    const float t2 = (float)(T)((float)b[gid] * om);
    out[gid] = (T)(t1 + t2);
}

extern "C" int gd_blend_tokens(const void* a, const void* b, const float* m, int H, int N, int D, void* out, int dtype, void* stream) {
    GD_REQUIRE(a && b && m && out, GD_EINVAL, "gd_blend_tokens: null pointer");
    GD_REQUIRE(H > 0 && N > 0 && D > 0, GD_EINVAL, "gd_blend_tokens: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_blend_tokens: dtype must be f16/bf16");
    const long long total = (long long)H * N * D;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) k_blend<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)a, (const f16_t*)b, m, H, N, D, (f16_t*)out);
    else k_blend<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, m, H, N, D, (bf16_t*)out);
    GD_CHECK_LAUNCH("gd_blend_tokens");
    return GD_OK;
}

// ---- full edit-attention output from the reference rows + the rows a q_rows segment computed (gd_attn_seg_t::q_rows) ----
template <typename T>
__global__ void k_rows_merge(const u32x4* __restrict__ base, const u32x4* __restrict__ act, const int32_t* __restrict__ pos, int H, int N,
                             int R, int D8, u32x4* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;                   // one 16-byte chunk of 8 elements
    if (gid >= (long long)H * N * D8) return;
    const int c = (int)(gid % D8);
    const long long hn = gid / D8;
    const int n = (int)(hn % N), h = (int)(hn / N);
    const int p = pos[n];
    out[gid] = p >= 0 ? act[((long long)h * R + p) * D8 + c] : base[gid];
}

extern "C" int gd_rows_merge(const void* base, const void* act, const int32_t* pos, int H, int N, int R, int D, void* out, int dtype,
                             void* stream) {
    GD_REQUIRE(base && act && pos && out, GD_EINVAL, "gd_rows_merge: null pointer");
    GD_REQUIRE(H > 0 && N > 0 && R > 0 && D > 0 && D % 8 == 0, GD_EINVAL, "gd_rows_merge: bad sizes (D must be a multiple of 8)");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_rows_merge: dtype must be f16/bf16");
    const long long total = (long long)H * N * (D / 8);
    k_rows_merge<f16_t><<<(int)((total + 255) / 256), 256, 0, as_stream(stream)>>>((const u32x4*)base, (const u32x4*)act, pos, H, N, R, D / 8,
                                                                                  (u32x4*)out);
    GD_CHECK_LAUNCH("gd_rows_merge");
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
