// Issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3) vs v_mfma_f32_32x32x16_bf16 on gfx950: one wave per SIMD, dependent-free chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    i32x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838; }
    bf16x8 ah, bh; for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)1.0f; bh[i] = (__bf16)1.0f; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c[j], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* o; long long* c; hipMalloc(&o, 1 << 20); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) for (int waves = 1; waves <= 2; ++waves) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (mode == 0) k<0><<<256, 256 * waves>>>(o, c, 10); else k<1><<<256, 256 * waves>>>(o, c, 10);
        hipDeviceSynchronize(); hipEventRecord(e0);
        if (mode == 0) k<0><<<256, 256 * waves>>>(o, c, iters); else k<1><<<256, 256 * waves>>>(o, c, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 4;
        const double flop = (mode == 0 ? 2.0 * 32 * 32 * 64 : 2.0 * 32 * 32 * 16) * n * 256 * 4 * waves;
        printf("%s, %d wave(s)/SIMD: %.1f clock64 ticks per instruction (one wave), %.1f TFLOP/s chip-wide\n", mode == 0 ? "f8f6f4 32x32x64 e4m3" : "bf16   32x32x16     ", waves,
               (double)cy / n, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
