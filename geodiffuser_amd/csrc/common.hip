// Error plumbing + version for libgeodiff_hip.so (include/geodiff_hip.h).
#include "common.hpp"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void gd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gd_version(void) { return GD_ABI_VERSION; }
extern "C" const char* gd_last_error(void) { return g_err; }
extern "C" const char* gd_error_string(int code) {
    switch (code) {
        case GD_OK: return "ok";
        case GD_EINVAL: return "invalid argument";
        case GD_EWORKSPACE: return "workspace too small";
        case GD_ELAUNCH: return "kernel launch failed";
        case GD_EUNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

__global__ void k_zero_u32(uint32_t* __restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

void gd_zero_async(void* ptr, size_t bytes, hipStream_t st) {
    const size_t n = (bytes + 3) / 4;                      // all scratch buffers are multiples of 4 bytes
    if (n == 0) return;
    k_zero_u32<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((uint32_t*)ptr, n);
}

// gd_copy_rows: blockIdx.y = entry, blockIdx.x strides over its 16-byte words
__global__ void k_copy_rows(const gd_copy_rows_t* __restrict__ entries, int row) {
    const gd_copy_rows_t e = entries[blockIdx.y];
    const long long n16 = e.bytes >> 4;
    const u32x4* __restrict__ src = (const u32x4*)((const char*)e.src + (long long)row * e.bytes);
    u32x4* __restrict__ dst = (u32x4*)e.dst;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int gd_copy_rows(const gd_copy_rows_t* entries, int n, int row, int64_t max_bytes, void* stream) {
    GD_REQUIRE(entries && n > 0 && n <= 65535 && row >= 0 && max_bytes > 0 && (max_bytes & 15) == 0, GD_EINVAL,
               "gd_copy_rows: bad arguments (n=%d, row=%d, max_bytes=%lld)", n, row, (long long)max_bytes);
    // enough blocks per entry for the largest one at 4 x 16 bytes per thread; smaller entries leave blocks idle (they exit at once)
    long long bx = (max_bytes / 16 + 256 * 4 - 1) / (256 * 4);
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    k_copy_rows<<<dim3((unsigned)bx, (unsigned)n), 256, 0, as_stream(stream)>>>(entries, row);
    GD_CHECK_LAUNCH("gd_copy_rows");
    return GD_OK;
}

// Id of the capture sequence `stream` is recording (0: not capturing).  Host-side helper for ops.zeros_f32: a pre-zeroed chunk must not
// be shared between two hipGraph captures (the fill belongs to one graph only).  Lives here so that it asks the HIP runtime this
// library is bound to — dlopen("libamdhip64.so") from Python can map a SECOND runtime next to the one PyTorch bundles.
extern "C" int gd_stream_capture_id(void* stream, unsigned long long* id) {
    GD_REQUIRE(id, GD_EINVAL, "gd_stream_capture_id: null pointer");
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    const hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &status, &cid);
    GD_REQUIRE(e == hipSuccess, GD_ELAUNCH, "gd_stream_capture_id: %s", hipGetErrorString(e));
    *id = status == hipStreamCaptureStatusActive ? cid + 1 : 0;
    return GD_OK;
}
