// N2 — masked per-channel histogram matching of the decoded image (post-process step right after the path).
//
// Replaces GeoDiffuser/utils/image_processing.py:24-77 (`_match_cumulative_cdf` per channel: np.bincount inside the masks,
// cumulative quantiles, np.interp onto the template's quantiles, LUT applied to the whole channel).  Integer part (counts,
// prefix sums) is exact; the quantiles and the interpolation are IEEE binary64 with the operation order of numpy's
// `arr_interp` (slope = (fp[j+1]-fp[j])/(xp[j+1]-xp[j]); slope*(x-xp[j]) + fp[j], no FMA contraction), so the float64
// result is bit-identical to the reference's.  HBM-bound, trivial sizes: 512*512*3 bytes in, 8x that out.
#include "common.hpp"

#pragma clang fp contract(off)

// counts[0][c][v]: source pixels of channel c with value v inside m_src; counts[1][c][v]: template inside m_tmpl
__global__ void __launch_bounds__(256)
k_masked_hist(const uint8_t* __restrict__ src, const uint8_t* __restrict__ tmpl, const uint8_t* __restrict__ m_src,
              const uint8_t* __restrict__ m_tmpl, int npix, int C, uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t s_cnt[];                 // [2][C][256]
    const int tid = threadIdx.x, nb = 2 * C * 256;
    for (int i = tid; i < nb; i += 256) s_cnt[i] = 0;
    __syncthreads();
    for (int p = blockIdx.x * 256 + tid; p < npix; p += gridDim.x * 256) {
        const bool a = m_src[p] != 0, b = m_tmpl[p] != 0;
        for (int c = 0; c < C; ++c) {
            if (a) atomicAdd(&s_cnt[c * 256 + src[(size_t)p * C + c]], 1u);
            if (b) atomicAdd(&s_cnt[(C + c) * 256 + tmpl[(size_t)p * C + c]], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < nb; i += 256)
        if (s_cnt[i]) atomicAdd(&counts[i], s_cnt[i]);
}

// one workgroup per channel, thread v owns value v
__global__ void __launch_bounds__(256)
k_hist_lut(const uint32_t* __restrict__ counts, int C, double* __restrict__ lut) {
    __shared__ uint32_t s_a[256], s_b[256];
    __shared__ double s_tq[256];
    const int c = blockIdx.x, v = threadIdx.x;
    s_a[v] = counts[c * 256 + v];
    s_b[v] = counts[(C + c) * 256 + v];
    __syncthreads();
    // inclusive prefix sums (Hillis-Steele; 256 entries)
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t xa = v >= o ? s_a[v - o] : 0u, xb = v >= o ? s_b[v - o] : 0u;
        __syncthreads();
        s_a[v] += xa; s_b[v] += xb;
        __syncthreads();
    }
    const double n_src = (double)s_a[255], n_tmpl = (double)s_b[255];
    const double x = (double)s_a[v] / n_src;                       // src_quantiles[v]
    s_tq[v] = (double)s_b[v] / n_tmpl;                             // tmpl_quantiles[v]
    __syncthreads();
    // np.interp(x, xp = tmpl_quantiles, fp = 0..255): j = last index with xp[j] <= x
    double r;
    if (x != x) {
        r = x;
    } else if (x > s_tq[255]) {
        r = 255.0;                                                  // right = fp[-1]
    } else if (x < s_tq[0]) {
        r = 0.0;                                                    // left = fp[0]
    } else {
        int lo = 0, hi = 256;                                       // first index with xp[idx] > x
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if (x >= s_tq[mid]) lo = mid + 1; else hi = mid;
        }
        const int j = lo - 1;
        if (j == 255 || s_tq[j] == x) {
            r = (double)j;
        } else {
            const double slope = ((double)(j + 1) - (double)j) / (s_tq[j + 1] - s_tq[j]);
            r = slope * (x - s_tq[j]) + (double)j;
            if (r != r) {
                r = slope * (x - s_tq[j + 1]) + (double)(j + 1);
            }
        }
    }
    lut[c * 256 + v] = r;
}

__global__ void k_apply_lut(const uint8_t* __restrict__ src, const double* __restrict__ lut, long long n, int C,
                            double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    out[i] = lut[c * 256 + src[i]];
}

extern "C" int gd_hist_match(const uint8_t* src, const uint8_t* tmpl, const uint8_t* m_src, const uint8_t* m_tmpl, int npix, int C,
                             uint32_t* counts, double* lut, double* out, void* stream) {
    GD_REQUIRE(src && tmpl && m_src && m_tmpl && counts && lut && out, GD_EINVAL, "gd_hist_match: null pointer");
    GD_REQUIRE(npix > 0 && C > 0 && C <= 4, GD_EINVAL, "gd_hist_match: bad sizes (npix=%d C=%d, C <= 4)", npix, C);
    hipStream_t st = as_stream(stream);
    gd_zero_async(counts, (size_t)2 * C * 256 * sizeof(uint32_t), st);
    int blocks = (npix + 255) / 256;
    if (blocks > 512) blocks = 512;
    k_masked_hist<<<blocks, 256, (size_t)2 * C * 256 * sizeof(uint32_t), st>>>(src, tmpl, m_src, m_tmpl, npix, C, counts);
    k_hist_lut<<<C, 256, 0, st>>>(counts, C, lut);
    const long long n = (long long)npix * C;
    k_apply_lut<<<(int)((n + 255) / 256), 256, 0, st>>>(src, lut, n, C, out);
    GD_CHECK_LAUNCH("gd_hist_match");
    return GD_OK;
}
