// Shared helpers for the gfx950 kernels behind include/geodiff_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/geodiff_hip.h"

#define GD_WAVE 64

void gd_set_error(const char* fmt, ...);

#define GD_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            gd_set_error(__VA_ARGS__);       \
            return (code);                   \
        }                                    \
    } while (0)

#define GD_CHECK_LAUNCH(name)                                                          \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            gd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
            return GD_ELAUNCH;                                                         \
        }                                                                              \
    } while (0)

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

template <typename T> struct elem_traits;
template <> struct elem_traits<f16_t> {
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }
    static __device__ __forceinline__ f16_t from_f32(float x) { return (f16_t)x; }
};
template <> struct elem_traits<bf16_t> {
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
    static __device__ __forceinline__ bf16_t from_f32(float x) { return (bf16_t)x; }
};

// wave-level sum over 64 lanes (result valid in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// Zero-fill as a KERNEL node: memset nodes inside a captured hipGraph were observed to race with neighbouring kernel nodes
// on this stack (memory faults on replay that disappear under AMD_SERIALIZE_KERNEL=3), kernels are ordered correctly.
void gd_zero_async(void* ptr, size_t bytes, hipStream_t st);
