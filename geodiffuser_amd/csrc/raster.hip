// R3a — point rasterizer: the integer warp-index grid (bit-exact against oracle/c/raster_ref.c).
//
// Replaces pytorch3d.rasterize_points as called at GeoDiffuser/utils/warp_utils.py:111.
// The reference re-rasterizes f identical clouds on every hooked attention call; the grid depends
// only on (coords, S, radius, K), so the host rasterizes ONCE per resolution per edit
// and shares idx/w across heads, layers, passes and steps.
//
// Algorithm (HBM-bound scatter/gather, no GEMM shape):
//   count : one thread per point walks its (conservative) pixel box, exact hit test, atomic count
//   scan  : exclusive prefix sum over the S*S counters (single workgroup, wave-shuffle scan)
//   fill  : same walk, slot = atomicAdd(cursor) -> candidate list (CSR)
//   select: one thread per pixel: sort its candidates by point index (visit order of the published
//           algorithm), run the K-slot queue (evict first max-z slot on strictly smaller z), stable
//           bubble sort on z, write idx / zbuf / dist2.
// Arithmetic of the hit test is kept identical to the oracle: binary32, dist2 = fma(dx,dx,dy*dy).
#include "common.hpp"

#pragma clang fp contract(off)

#define GD_MAX_K 32
#define SEL_THREADS 64

__device__ __forceinline__ float pix_to_ndc(int i, int S) {
    const float range = 2.0f;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S;
}

__device__ __forceinline__ void cand_range(float p, float r, int S, int& lo, int& hi) {
    double a = ((1.0 - ((double)p + (double)r)) * S - 1.0) * 0.5;
    double b = ((1.0 - ((double)p - (double)r)) * S - 1.0) * 0.5;
    if (!(a == a) || !(b == b)) { lo = 0; hi = -1; return; }
    if (a < -2.0) a = -2.0;
    if (b > S + 1.0) b = S + 1.0;
    if (b < a) { lo = 0; hi = -1; return; }
    int l = (int)floor(a) - 1, h = (int)ceil(b) + 1;
    if (l < 0) l = 0;
    if (h > S - 1) h = S - 1;
    lo = l; hi = h;
}

// exact hit test shared by count / fill / select
__device__ __forceinline__ bool hit(float px, float py, float pz, float xf, float yf, float r2, float& d2) {
    if (pz < 0) return false;
    const float dx = xf - px, dy = yf - py;
    d2 = __builtin_fmaf(dx, dx, dy * dy);
    return d2 < r2;
}

template <bool FILL>
__global__ void k_count_fill(const float* __restrict__ pts, int P, int S, float radius, float r2,
                             int* __restrict__ count, const int* __restrict__ offset, int* __restrict__ cursor,
                             int* __restrict__ cand) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
    int c0, c1, r0, r1;
    cand_range(px, radius, S, c0, c1);
    cand_range(py, radius, S, r0, r1);
    for (int r = r0; r <= r1; ++r) {
        const float yf = pix_to_ndc(S - 1 - r, S);
        for (int c = c0; c <= c1; ++c) {
            const float xf = pix_to_ndc(S - 1 - c, S);
            float d2;
            if (!hit(px, py, pz, xf, yf, r2, d2)) continue;
            const int pix = r * S + c;
            if (FILL) {
                const int slot = atomicAdd(&cursor[pix], 1);
                cand[offset[pix] + slot] = p;
            } else {
                atomicAdd(&count[pix], 1);
            }
        }
    }
}

// exclusive scan of count[0..n) into offset[0..n], offset[n] = total, in two launches over blocks of SCAN_BLOCK elements (one workgroup
// of 1024 threads looping over the array took 281 us at 512^2: 256 iterations x 3 barriers):
//   k_scan_local: every workgroup scans its own block (16 consecutive elements per thread, wave scan of the thread sums) and leaves the
//                 block's total in bsum[block];
//   k_scan_add:   adds the sum of the preceding blocks' totals (<= a few hundred values, summed by one wave) to the block's offsets.
#define SCAN_BLOCK 4096
__global__ void __launch_bounds__(256) k_scan_local(const int* __restrict__ count, int* __restrict__ offset, int* __restrict__ bsum, int n) {
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i0 = blockIdx.x * SCAN_BLOCK + tid * 16;
    int v[16];
    int tsum = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { v[e] = (i0 + e < n) ? count[i0 + e] : 0; tsum += v[e]; }
    int x = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    int pre = x - tsum;
    for (int w = 0; w < wid; ++w) pre += wave_tot[w];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (i0 + e < n) offset[i0 + e] = pre;
        pre += v[e];
    }
    if (tid == 255) bsum[blockIdx.x] = pre;
}

__global__ void __launch_bounds__(256) k_scan_add(int* __restrict__ offset, const int* __restrict__ bsum, int n, int nblocks) {
    __shared__ int carry_s;
    const int tid = threadIdx.x;
    if (tid < 64) {
        int c = 0;
        for (int b2 = tid; b2 < (int)blockIdx.x; b2 += 64) c += bsum[b2];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (tid == 0) carry_s = c;
    }
    __syncthreads();
    const int carry = carry_s;
    const int i0 = blockIdx.x * SCAN_BLOCK + tid * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (i0 + e < n) offset[i0 + e] += carry;
    if ((int)blockIdx.x == nblocks - 1 && tid == 0) offset[n] = carry + bsum[blockIdx.x];
}

__device__ void sort_indices(int* a, int n) {
    if (n <= 32) {
        for (int i = 1; i < n; ++i) {
            const int v = a[i];
            int j = i - 1;
            while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
            a[j + 1] = v;
        }
        return;
    }
    // in-place heapsort for the (degenerate) long lists
    for (int start = n / 2 - 1; start >= 0; --start) {
        int root = start;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= n) break;
            if (child + 1 < n && a[child] < a[child + 1]) child++;
            if (a[root] < a[child]) { int t = a[root]; a[root] = a[child]; a[child] = t; root = child; } else break;
        }
    }
    for (int end = n - 1; end > 0; --end) {
        int t = a[0]; a[0] = a[end]; a[end] = t;
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && a[child] < a[child + 1]) child++;
            if (a[root] < a[child]) { int t2 = a[root]; a[root] = a[child]; a[child] = t2; root = child; } else break;
        }
    }
}

__global__ void __launch_bounds__(SEL_THREADS)
k_select(const float* __restrict__ pts, int S, int K, float r2, const int* __restrict__ offset,
         int* __restrict__ cand, int32_t* __restrict__ idx, float* __restrict__ zbuf, float* __restrict__ dist2) {
    __shared__ float qz[GD_MAX_K][SEL_THREADS];
    __shared__ float qd[GD_MAX_K][SEL_THREADS];
    __shared__ int qi[GD_MAX_K][SEL_THREADS];
    const int t = threadIdx.x;
    const int pix = blockIdx.x * SEL_THREADS + t;
    if (pix >= S * S) return;
    const int yi = pix / S, xi = pix - yi * S;
    const float yf = pix_to_ndc(S - 1 - yi, S), xf = pix_to_ndc(S - 1 - xi, S);
    const int beg = offset[pix], n = offset[pix + 1] - beg;
    int* lst = cand + beg;
    sort_indices(lst, n);
    int qn = 0, qmax_i = -1;
    float qmax_z = -1000.0f;
    for (int e = 0; e < n; ++e) {
        const int p = lst[e];
        const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
        float d2;
        if (!hit(px, py, pz, xf, yf, r2, d2)) continue;   // always true for listed candidates
        if (qn < K) {
            qz[qn][t] = pz; qi[qn][t] = p; qd[qn][t] = d2;
            if (pz > qmax_z) { qmax_z = pz; qmax_i = qn; }
            qn++;
        } else if (pz < qmax_z) {
            qz[qmax_i][t] = pz; qi[qmax_i][t] = p; qd[qmax_i][t] = d2;
            qmax_z = pz;
            for (int i = 0; i < K; ++i)
                if (qz[i][t] > qmax_z) { qmax_z = qz[i][t]; qmax_i = i; }
        }
    }
    for (int i = 0; i < qn - 1; ++i)
        for (int j = 0; j < qn - i - 1; ++j)
            if (qz[j + 1][t] < qz[j][t]) {
                float a = qz[j][t]; qz[j][t] = qz[j + 1][t]; qz[j + 1][t] = a;
                float b = qd[j][t]; qd[j][t] = qd[j + 1][t]; qd[j + 1][t] = b;
                int c = qi[j][t]; qi[j][t] = qi[j + 1][t]; qi[j + 1][t] = c;
            }
    const size_t o = (size_t)pix * K;
    for (int k = 0; k < K; ++k) {
        const bool v = k < qn;
        idx[o + k] = v ? qi[k][t] : -1;
        if (zbuf) zbuf[o + k] = v ? qz[k][t] : -1.0f;
        dist2[o + k] = v ? qd[k][t] : -1.0f;
    }
}

static size_t hits_per_point_bound(int S, float radius_ndc) {
    // pixel centres strictly within r of a point: their unit squares are disjoint and lie inside
    // a disc of radius r + sqrt(2)/2 (pixel units)
    const double r_px = (double)radius_ndc * S * 0.5;
    const double area = 3.14159265358979 * (r_px + 0.7072) * (r_px + 0.7072);
    size_t b = (size_t)area + 2;
    return b;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t gd_rasterize_workspace_bytes(int P, int S, float radius_ndc) {
    if (P <= 0 || S <= 0) return 0;
    const size_t npix = (size_t)S * S;
    const size_t nblocks = (npix + SCAN_BLOCK - 1) / SCAN_BLOCK;
    return align256((npix + 1) * 4) * 3 + align256(nblocks * 4) + align256((size_t)P * hits_per_point_bound(S, radius_ndc) * 4) + 256;
}

extern "C" int gd_rasterize_points(const float* pts, int P, int S, float radius_ndc, int K,
                                   int32_t* idx, float* zbuf, float* dist2,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    GD_REQUIRE(pts && idx && dist2 && workspace, GD_EINVAL, "gd_rasterize_points: null pointer");
    GD_REQUIRE(P > 0 && S > 0 && K > 0 && K <= GD_MAX_K, GD_EINVAL,
               "gd_rasterize_points: bad sizes P=%d S=%d K=%d (K<=%d)", P, S, K, GD_MAX_K);
    GD_REQUIRE(workspace_bytes >= gd_rasterize_workspace_bytes(P, S, radius_ndc), GD_EWORKSPACE,
               "gd_rasterize_points: workspace %zu < %zu", workspace_bytes, gd_rasterize_workspace_bytes(P, S, radius_ndc));
    hipStream_t st = as_stream(stream);
    const size_t npix = (size_t)S * S;
    char* w = (char*)workspace;
    int* count = (int*)w;            w += align256((npix + 1) * 4);
    int* offset = (int*)w;           w += align256((npix + 1) * 4);
    int* cursor = (int*)w;           w += align256((npix + 1) * 4);
    const int nblocks = (int)((npix + SCAN_BLOCK - 1) / SCAN_BLOCK);
    int* bsum = (int*)w;             w += align256((size_t)nblocks * 4);
    int* cand = (int*)w;
    const float r2 = radius_ndc * radius_ndc;
    gd_zero_async(count, (npix + 1) * 4, st);
    gd_zero_async(cursor, (npix + 1) * 4, st);
    const int tp = 256;
    k_count_fill<false><<<(P + tp - 1) / tp, tp, 0, st>>>(pts, P, S, radius_ndc, r2, count, nullptr, nullptr, nullptr);
    k_scan_local<<<nblocks, 256, 0, st>>>(count, offset, bsum, (int)npix);
    k_scan_add<<<nblocks, 256, 0, st>>>(offset, bsum, (int)npix, nblocks);
    k_count_fill<true><<<(P + tp - 1) / tp, tp, 0, st>>>(pts, P, S, radius_ndc, r2, nullptr, offset, cursor, cand);
    k_select<<<((int)npix + SEL_THREADS - 1) / SEL_THREADS, SEL_THREADS, 0, st>>>(pts, S, K, r2, offset, cand, idx, zbuf, dist2);
    GD_CHECK_LAUNCH("gd_rasterize_points");
    return GD_OK;
}
