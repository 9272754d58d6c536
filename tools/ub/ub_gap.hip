// Micro-benchmark of ONE "gap" of the attention inner loop on gfx950 (development aid, not part of the library):
// cycles per gap (s_memtime) for 1 MFMA + the softmax vector work of 2 probabilities, in several formulations, at 1 and 2 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -o ub_gap tools/ub/ub_gap.hip && ./ub_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2 t; t[0] = (__bf16)a; t[1] = (__bf16)b; uint32_t w = __builtin_bit_cast(uint32_t, t); asm volatile("" : "+v"(w)); return w;
}
__device__ __forceinline__ f32x16 mf(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, long long* cyc, int iters, float c, float mc) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    f32x16 acc0, acc1, X;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; X[i] = 0.001f * (lane + 3 * i); }
    u32x4 P = {0, 0, 0, 0};
    float ps0 = 0.f, ps1 = 0.f;
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const char* lp = lds + lane * 16;
    u32x4 L = {0, 0, 0, 0};
    float x0 = 0.f, x1 = 0.f, p0 = 1.f, p1 = 1.f;                // pipeline registers (MODE 3/5/6)
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 0) { if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0); }
            if (MODE == 1) { acc0 = mf(a, b, acc0); }
            if (MODE == 2 || MODE == 4) {
                if (MODE == 2) { if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0); }
                const float q0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[2 * g], c, -mc));
                const float q1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[2 * g + 1], c, -mc));
                P[g & 3] = pack2(q0, q1); ps0 += q0; ps1 += q1;
            }
            if (MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7) {
                if (MODE != 5) { if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0); }
                if (MODE == 6) L = *(const u32x4*)(lp + (g & 3) * 1024);
                // stage 3 of group g-2: sum + pack; stage 2 of group g-1: exp; stage 1 of group g: fma
                P[g & 3] = pack2(p0, p1); ps0 += p0; ps1 += p1;
                p0 = __builtin_amdgcn_exp2f(x0); p1 = __builtin_amdgcn_exp2f(x1);
                x0 = __builtin_fmaf(X[2 * g], c, -mc); x1 = __builtin_fmaf(X[2 * g + 1], c, -mc);
                if (MODE == 6) { asm volatile("" :: "v"(L)); }
                if (MODE == 7) { ps0 += 0.f; }
            }
            if (MODE == 9) {          // pre-scaled queries: MFMA + 2 x (exp, add) + cvt, each result used at once
                if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0);
                const float q0 = __builtin_amdgcn_exp2f(X[2 * g]);
                const float q1 = __builtin_amdgcn_exp2f(X[2 * g + 1]);
                P[g & 3] = pack2(q0, q1); ps0 += q0; ps1 += q1;
            }
            if (MODE == 10) {         // the same, the pack / adds of a gap consume the PREVIOUS gap's exponentials
                if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0);
                P[g & 3] = pack2(p0, p1); ps0 += p0; ps1 += p1;
                p0 = __builtin_amdgcn_exp2f(X[2 * g]); p1 = __builtin_amdgcn_exp2f(X[2 * g + 1]);
            }
            if (MODE == 11) {         // exponentials first, then the MFMA, then the previous gap's pack / adds
                p0 = __builtin_amdgcn_exp2f(X[2 * g]); p1 = __builtin_amdgcn_exp2f(X[2 * g + 1]);
                if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0);
                P[g & 3] = pack2(x0, x1); ps0 += x0; ps1 += x1;
                x0 = p0; x1 = p1;
            }
            if (MODE == 8) {          // MFMA + 7 independent plain VALU (no transcendental)
                if (g & 1) acc1 = mf(a, b, acc1); else acc0 = mf(a, b, acc0);
                x0 = __builtin_fmaf(X[2 * g], c, x0); x1 = __builtin_fmaf(X[2 * g + 1], c, x1);
                p0 = __builtin_fmaf(X[2 * g], mc, p0); p1 = __builtin_fmaf(X[2 * g + 1], mc, p1);
                ps0 += X[g]; ps1 += X[g + 8]; P[g & 3] = pack2(X[g], X[15 - g]);
            }
            SB();
        }
        if (MODE >= 2) { asm volatile("" : "+v"(X)); }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = ps0 + ps1 + x0 + x1 + p0 + p1;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    r += (float)(P[0] + P[1] + P[2] + P[3] + L[0]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char* name, int blocks) {
    float* out; long long* cyc; const int iters = 2000;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    for (int rep = 0; rep < 3; ++rep) k<MODE><<<blocks, 256>>>(out, cyc, iters, 0.18f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, cyc, iters, 0.18f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4); hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 8.0);
    // s_memtime ticks at 100 MHz on this part?  report both the raw tick count per gap and the wall-clock ns per gap
    printf("%-46s blocks=%4d (%d wave/SIMD): %7.2f memtime-ticks/gap   %7.2f ns/gap (wall)\n", name, blocks, blocks / 256, med, ms * 1e6 / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0>("MFMA only, 2 accumulators", blocks);
        run<1>("MFMA only, 1 accumulator chain", blocks);
        run<8>("MFMA + 7 independent plain VALU", blocks);
        run<2>("MFMA + 2x(fma,exp,add)+cvt dependent", blocks);
        run<3>("MFMA + same work, pipelined across gaps", blocks);
        run<4>("VALU only: 2x(fma,exp,add)+cvt dependent", blocks);
        run<5>("VALU only: pipelined across gaps", blocks);
        run<6>("MFMA + pipelined + ds_read_b128", blocks);
        run<9>("pre-scaled: MFMA + 2x(exp,add)+cvt dependent", blocks);
        run<10>("pre-scaled: pack/add one gap behind the exps", blocks);
        run<11>("pre-scaled: exps, MFMA, previous pack/adds", blocks);
    }
    return 0;
}
