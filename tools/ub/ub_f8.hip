// Operand-layout check of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) and of v_cvt_pk_fp8_f32 on gfx950, with exact
// integer data (development aid for attn_fwd_f8.hip).   hipcc --offload-arch=gfx950 -O2 tools/ub/ub_f8.hip -o /tmp/ub_f8 && /tmp/ub_f8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// OCP e4m3fn encode of a small non-negative integer (exact up to 16) / decode of any code
static uint8_t enc(int v) {
    if (v == 0) return 0;
    int e = 0; float m = (float)v;
    while (m >= 2.f) { m *= 0.5f; ++e; }
    const int mant = (int)((m - 1.f) * 8.f + 0.5f);
    return (uint8_t)(((e + 7) << 3) | mant);
}
static float dec(uint8_t c) {
    const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    float v = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    if (e == 15 && m == 7) v = NAN;
    return s ? -v : v;
}

__global__ void k_mfma(const uint8_t* A /*[32][64]*/, const uint8_t* B /*[64][32]*/, float* C /*[32][32]*/) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { i32x8 v; uint8_t b[32]; } a, b;
    for (int j = 0; j < 32; ++j) { a.b[j] = A[r * 64 + 32 * h + j]; b.b[j] = B[(32 * h + j) * 32 + r]; }      // hypothesis: byte j <-> k = 32 h + j
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];                        // row = (i&3)+8(i>>2)+4h, col = r
}

__global__ void k_cvt(const float* x, uint8_t* y, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n + 1) return;
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[i + 1], 0, false);
    y[i] = (uint8_t)(w & 0xFF); y[i + 1] = (uint8_t)((w >> 8) & 0xFF);
}

int main() {
    std::vector<uint8_t> A(32 * 64), B(64 * 32);
    std::vector<float> Cref(32 * 32, 0.f), C(32 * 32);
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) A[r * 64 + k] = enc((r * 3 + k * 5 + (k >> 3)) % 7);
    for (int k = 0; k < 64; ++k) for (int c = 0; c < 32; ++c) B[k * 32 + c] = enc((k * 2 + c * 7 + (k >> 2)) % 5);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < 64; ++k) s += dec(A[r * 64 + k]) * dec(B[k * 32 + c]); Cref[r * 32 + c] = s; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32 * 32; ++i) if (C[i] != Cref[i]) { if (bad < 5) printf("mismatch at [%d][%d]: %g vs %g\n", i / 32, i % 32, C[i], Cref[i]); ++bad; }
    printf("mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, scale 2^0), byte j <-> k = 32h + j: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    // cvt: round-to-nearest-even? saturation?
    const float xs[] = {0.f, 1.f, 1.0625f, 1.1875f, 1.3125f, 0.0019531f, 0.001f, 0.00097f, 448.f, 464.f, 480.f, 500.f, 1e6f, -3.3f, 0.0166f, 17.f, 18.f, 19.f, 27.f, 29.f, 0.30f, 7.5e-4f};
    const int n = sizeof(xs) / 4; float* dx; uint8_t* dy; std::vector<uint8_t> y(n + 1);
    hipMalloc(&dx, (n + 2) * 4); hipMalloc(&dy, n + 2); hipMemset(dx, 0, (n + 2) * 4); hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(dx, dy, n + (n & 1));
    hipMemcpy(y.data(), dy, n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", xs[i], y[i], dec(y[i]));
    return bad != 0;
}
