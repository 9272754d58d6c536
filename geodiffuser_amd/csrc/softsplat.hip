// N4 — bilinear forward splatting ("softmax splatting", summation form) and its two gradients, f32.
//
// gfx950 equivalents of the three cupy/CUDA kernel strings of GeoDiffuser/utils/softsplat.py (softsplat_out :284-340,
// softsplat_ingrad :364-420, softsplat_flowgrad :430-510).  They are dead on the reference's live path (the point splat replaced
// them) and are provided so that every native component of the reference has a counterpart here.
//
//   target of source pixel (y, x):  (fx, fy) = (x + flow[n,0,y,x], y + flow[n,1,y,x]);  non-finite targets contribute nothing
//   out[n,c,Y,X] += in[n,c,y,x] * w(Y,X)  for the four integer neighbours of the target inside the image, bilinear weights
//
// One thread per source pixel: corner indices and weights are computed once and reused for all C channels (the reference spends a
// thread per (pixel, channel) and recomputes them); reads of `in` are coalesced along x, the scatter uses f32 atomics.  HBM /
// atomic bound: algorithmic bytes 4*N*H*W*(2 + C) read + 4*4*N*C*H*W atomically added.
#include "common.hpp"

struct Corners { int idx[4]; float w[4]; bool ok[4]; };

__device__ __forceinline__ bool splat_corners(const float* __restrict__ flow, int n, int y, int x, int H, int W, Corners& k,
                                              float& fx, float& fy, int& x0, int& y0) {
    const size_t hw = (size_t)H * W;
    fx = (float)x + flow[((size_t)n * 2) * hw + (size_t)y * W + x];
    fy = (float)y + flow[((size_t)n * 2 + 1) * hw + (size_t)y * W + x];
    if (!isfinite(fx) || !isfinite(fy)) return false;
    x0 = (int)floorf(fx); y0 = (int)floorf(fy);
    const float ax = (float)(x0 + 1) - fx, bx = fx - (float)x0, ay = (float)(y0 + 1) - fy, by = fy - (float)y0;
    const int xs[4] = {x0, x0 + 1, x0, x0 + 1}, ys[4] = {y0, y0, y0 + 1, y0 + 1};
    const float ws[4] = {ax * ay, bx * ay, ax * by, bx * by};              // NW, NE, SW, SE
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        k.ok[c] = xs[c] >= 0 && xs[c] < W && ys[c] >= 0 && ys[c] < H;
        k.idx[c] = ys[c] * W + xs[c];
        k.w[c] = ws[c];
    }
    return true;
}

__global__ void k_softsplat_fwd(const float* __restrict__ in, const float* __restrict__ flow, int N, int C, int H, int W,
                                float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * H * W) return;
    const int x = (int)(gid % W), y = (int)((gid / W) % H), n = (int)(gid / ((long long)W * H));
    Corners k; float fx, fy; int x0, y0;
    if (!splat_corners(flow, n, y, x, H, W, k, fx, fy, x0, y0)) return;
    const size_t hw = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float v = in[((size_t)n * C + c) * hw + (size_t)y * W + x];
        float* o = out + ((size_t)n * C + c) * hw;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k.ok[q]) atomicAdd(o + k.idx[q], v * k.w[q]);
    }
}

// ingrad[n,c,y,x] = sum_corners outgrad[n,c,corner] * w;  flowgrad[n,0|1,y,x] = sum_c in * sum_corners outgrad * dw/dfx|dfy
__global__ void k_softsplat_bwd(const float* __restrict__ in, const float* __restrict__ flow, const float* __restrict__ og, int N, int C,
                                int H, int W, float* __restrict__ ingrad, float* __restrict__ flowgrad) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * H * W) return;
    const int x = (int)(gid % W), y = (int)((gid / W) % H), n = (int)(gid / ((long long)W * H));
    const size_t hw = (size_t)H * W, pix = (size_t)y * W + x;
    Corners k; float fx, fy; int x0, y0;
    if (!splat_corners(flow, n, y, x, H, W, k, fx, fy, x0, y0)) {          // the reference leaves its zero-initialised outputs untouched
        if (ingrad) for (int c = 0; c < C; ++c) ingrad[((size_t)n * C + c) * hw + pix] = 0.f;
        if (flowgrad) { flowgrad[((size_t)n * 2) * hw + pix] = 0.f; flowgrad[((size_t)n * 2 + 1) * hw + pix] = 0.f; }
        return;
    }
    const float ax = (float)(x0 + 1) - fx, bx = fx - (float)x0, ay = (float)(y0 + 1) - fy, by = fy - (float)y0;
    const float dwx[4] = {-ay, ay, -by, by}, dwy[4] = {-ax, -bx, ax, bx};   // d w / d fx, d w / d fy
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* g = og + ((size_t)n * C + c) * hw;
        float gi = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k.ok[q]) {
                const float gv = g[k.idx[q]];
                gi = __builtin_fmaf(gv, k.w[q], gi);
                sx = __builtin_fmaf(gv, dwx[q], sx);
                sy = __builtin_fmaf(gv, dwy[q], sy);
            }
        if (ingrad) ingrad[((size_t)n * C + c) * hw + pix] = gi;
        if (flowgrad) {
            const float v = in[((size_t)n * C + c) * hw + pix];
            gx = __builtin_fmaf(v, sx, gx);
            gy = __builtin_fmaf(v, sy, gy);
        }
    }
    if (flowgrad) { flowgrad[((size_t)n * 2) * hw + pix] = gx; flowgrad[((size_t)n * 2 + 1) * hw + pix] = gy; }
}

extern "C" int gd_softsplat_fwd(const float* in, const float* flow, int N, int C, int H, int W, float* out, void* stream) {
    GD_REQUIRE(in && flow && out, GD_EINVAL, "gd_softsplat_fwd: null pointer");
    GD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, GD_EINVAL, "gd_softsplat_fwd: bad sizes");
    hipStream_t st = as_stream(stream);
    gd_zero_async(out, (size_t)N * C * H * W * sizeof(float), st);
    const long long total = (long long)N * H * W;
    k_softsplat_fwd<<<(int)((total + 255) / 256), 256, 0, st>>>(in, flow, N, C, H, W, out);
    GD_CHECK_LAUNCH("gd_softsplat_fwd");
    return GD_OK;
}

extern "C" int gd_softsplat_bwd(const float* in, const float* flow, const float* outgrad, int N, int C, int H, int W, float* ingrad,
                                float* flowgrad, void* stream) {
    GD_REQUIRE(in && flow && outgrad && (ingrad || flowgrad), GD_EINVAL, "gd_softsplat_bwd: null pointer");
    GD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, GD_EINVAL, "gd_softsplat_bwd: bad sizes");
    const long long total = (long long)N * H * W;
    k_softsplat_bwd<<<(int)((total + 255) / 256), 256, 0, as_stream(stream)>>>(in, flow, outgrad, N, C, H, W, ingrad, flowgrad);
    GD_CHECK_LAUNCH("gd_softsplat_bwd");
    return GD_OK;
}
