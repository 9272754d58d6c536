// UNet harness: 3x3 convolution (padding 1; stride 1 or 2; optionally over the nearest-neighbour 2x upsampling of its input) on NHWC
// 16-bit tensors as an implicit GEMM on the gfx950 matrix cores.  Not a SURVEY §8 row of its own: it is the library call
// (torch.nn.functional.conv2d -> MIOpen) that the reference's UNet makes ~60 times per pass around the attention hooks, and after round 2
// the largest single share of an edit (31 % of the kernel time; MIOpen's kernels for these shapes issue the 32x32x8 matrix instruction
// and leave 100-160 of 256 CUs idle at batch 1-3).
//
//   out[p, k] = bias[k] + residual[p, k] + sum_{ky, kx, c} in[img, y*stride + ky - 1, x*stride + kx - 1, c] * w[k, ky, kx, c]
//
// GEMM view: the reduction index (tap, c) is contiguous in BOTH operands — w is [K, 3, 3, C] ("KYXC", what a channels_last
// torch weight holds) and an input pixel's C channels are contiguous — so a reduction step of 64 channels of one tap is a 64-row x 128-B
// tile of each operand, the shape the attention kernels already stage (attn_common.hpp: swizzled 8 KB LDS image, conflict-free
// ds_read_b128 fragments).  A = weights (row = output channel), B = pixels (column = MFMA lane), v_mfma_f32_32x32x16: a lane ends up
// with 16 output channels of ITS pixel, stored as 8-byte pieces.
//
// Workgroup = 4 waves = (64 PI) pixels x (64 KI) output channels; wave (wp, wk) owns PI x KI accumulator blocks.  Halo / padding /
// tails are out-of-range buffer-load offsets (the buffer returns zeros).  Two stagings of the operand tiles (template flag DMA), both
// two reduction steps ahead of the matrix work and one barrier per step, bit-identical results:
//   registers : global -> registers -> ds_write_b128 into a double-buffered LDS image (the 128 x 128 tile);
//   DMA       : buffer_load_dwordx4 ... lds straight into a THREE-stage image (the smaller tiles: no staging registers, no ds_write;
//               the image's swizzle is applied to the source address).
// Launches that cannot fill the chip split the REDUCTION over workgroups (f32 partials, folded in a fixed order by k_conv_fold:
// deterministic).
#include <math.h>
#include <stdlib.h>
#include "attn_common.hpp"

// one direct-to-LDS piece: 64 lanes x 16 B land at lds_dst + 16 * lane (wave-uniform destination, per-lane source offset).  The builtin
// only exists in the device pass; the host pass needs a body to instantiate the kernel's launch stub.
__device__ __forceinline__ void conv_dma16(const __amdgpu_buffer_rsrc_t rsrc, char* lds_dst, uint32_t voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
#endif
}

struct ConvArgs {
    const void* in; const void* w; const void* bias; const void* res; void* out; float* ws;
    int n, Hi, Wi, C, K, Ho, Wo, stride, up;
    int P;                       // n * Ho * Wo output pixels
    int Hv, Wv;                  // extent of the (virtual) input grid the taps index: 2 Hi x 2 Wi when upsampling
    int tiles_p, tiles_k, ksplit, steps, spc, nwg;     // steps = 9 C / 64 reduction steps, spc of them per split
    int cpt;                     // steps per tap = C / 64
    int order;                   // workgroup order inside an XCD's chunk: 0 = pixel tile slowest, 1 = pixel tile fastest (see gd_conv3x3)
};

template <typename T, int PI, int KI, bool DMA>
__global__ void __launch_bounds__(256, 2)
k_conv3x3(const ConvArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int NT = PI + KI;                  // 64-row tiles per stage: PI pixel tiles then KI weight tiles
    constexpr int NS = (DMA && NT <= 3) ? 3 : 2;   // LDS stages: direct-to-LDS loads need no registers, so the small tiles run three
    __shared__ __attribute__((aligned(16))) char lds[NS][NT][ATT_TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wk = wave >> 1;
    int wg = xcd_remap(blockIdx.x, a.nwg);
    int sp, tk, tp;
    if (a.order == 0) {
        sp = wg % a.ksplit; wg /= a.ksplit;
        tk = wg % a.tiles_k; tp = wg / a.tiles_k;
    } else {
        tp = wg % a.tiles_p; wg /= a.tiles_p;
        sp = wg % a.ksplit; tk = wg / a.ksplit;
    }
    const int p0 = tp * (64 * PI), k0 = tk * (64 * KI);
    const int s_lo = sp * a.spc;
    const int s_hi = (s_lo + a.spc) < a.steps ? (s_lo + a.spc) : a.steps;
    const int ns = s_hi - s_lo;

    const int C = a.C;
    const __amdgpu_buffer_rsrc_t ib = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.n * a.Hi * a.Wi * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wb = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.K * 9 * C * 2), 0x00020000);

    // this thread's chunks: rows (tid >> 3) + 32 i of every tile, 16-B chunk tid & 7
    // register staging: chunk tid & 7 of the row, stored to its swizzled slot.  Direct-to-LDS (DMA): a wave-instruction fills 1 KB = 8 rows
    // x 8 slots in lane order, so lane i must FETCH the chunk that belongs in slot i & 7 of its row (the swizzle moves to the source
    // address; rows crow and crow + 32 share it)
    const int crow = tid >> 3;
    int cch = tid & 7;
    if (DMA) {
        const int x_ = (crow >> 1) & 7;
        cch ^= (x_ & 2) | ((x_ & 1) << 2) | ((x_ >> 2) & 1);
    }
    int pbase[2 * PI], vy[2 * PI], vx[2 * PI];
#pragma unroll
    for (int r = 0; r < 2 * PI; ++r) {
        const int p = p0 + (r >> 1) * 64 + (r & 1) * 32 + crow;
        const int hw = a.Ho * a.Wo;
        const int img = p / hw, rem = p - img * hw;
        const int y = rem / a.Wo, x = rem - y * a.Wo;
        pbase[r] = img * a.Hi * a.Wi;
        vy[r] = p < a.P ? y * a.stride - 1 : -(1 << 20);
        vx[r] = x * a.stride - 1;
    }
    uint32_t woff[2 * KI];
#pragma unroll
    for (int r = 0; r < 2 * KI; ++r)
        woff[r] = (uint32_t)(((size_t)(k0 + (r >> 1) * 64 + (r & 1) * 32 + crow) * 9 * C + cch * 8) * 2);
    int loff[2];
    loff[0] = img_off(crow, cch);
    loff[1] = img_off(crow + 32, cch);

    // Load stream.  Steps are loaded in ascending order, so the position inside the reduction is running SCALAR state: the byte offset of
    // the 64-channel chunk inside the tap (c0b, the buffer loads' scalar offset), the tap (ky, kx) and the weight tiles' step offset.
    // The per-thread part — which input pixel a tile row reads for this tap, or out of range for the halo — changes only when the tap
    // does (every C / 64 steps) and is recomputed there; a step itself costs no vector address arithmetic.  (The first version
    // re-derived everything per step: ~100 vector / scalar instructions with two integer divisions, 0.5 us per step and CU — more than
    // the loads, LDS traffic and MFMAs together; tools/price_conv.py.)
    uint32_t roff[2 * PI];
    int ld_ky, ld_kx, ld_c0b, ld_wso;
    {
        const int tap0 = s_lo / a.cpt;
        ld_ky = tap0 / 3; ld_kx = tap0 - ld_ky * 3;
        ld_c0b = (s_lo - tap0 * a.cpt) * 128;
        ld_wso = s_lo * 128;
    }
#define GD_CONV_SET_TAP()                                                                                                \
    _Pragma("unroll") for (int r = 0; r < 2 * PI; ++r) {                                                                 \
        int y_ = vy[r] + ld_ky, x_ = vx[r] + ld_kx;                                                                      \
        const bool ok_ = (unsigned)y_ < (unsigned)a.Hv && (unsigned)x_ < (unsigned)a.Wv;                                 \
        if (a.up) { y_ >>= 1; x_ >>= 1; }                                                                                \
        roff[r] = ok_ ? (uint32_t)(((pbase[r] + y_ * a.Wi + x_) * C + cch * 8) * 2) : 0x80000000u;                       \
    }
    GD_CONV_SET_TAP();
#define GD_CONV_LOAD(R)                                                                                                  \
    {                                                                                                                    \
        _Pragma("unroll") for (int r = 0; r < 2 * PI; ++r) R[r] = __builtin_amdgcn_raw_buffer_load_b128(ib, roff[r], ld_c0b, 0); \
        _Pragma("unroll") for (int r = 0; r < 2 * KI; ++r)                                                               \
            R[2 * PI + r] = __builtin_amdgcn_raw_buffer_load_b128(wb, woff[r], ld_wso, 0);                               \
        ld_wso += 128;                                                                                                   \
        ld_c0b += 128;                                                                                                   \
        if (ld_c0b == 2 * C) {                                                                                           \
            ld_c0b = 0;                                                                                                  \
            if (++ld_kx == 3) { ld_kx = 0; ++ld_ky; }                                                                    \
            GD_CONV_SET_TAP();                                                                                           \
        }                                                                                                                \
    }
// the same load stream straight into the LDS image of the stage at DST (buffer_load ... lds): no staging registers, no ds_write
#define GD_CONV_DMA(DST)                                                                                                 \
    {                                                                                                                    \
        char* const d_ = (DST) + wave * 1024;                                                                            \
        _Pragma("unroll") for (int r = 0; r < 2 * PI; ++r)                                                               \
            conv_dma16(ib, d_ + (r >> 1) * ATT_TILE_BYTES + (r & 1) * 4096, roff[r], ld_c0b);                            \
        _Pragma("unroll") for (int r = 0; r < 2 * KI; ++r)                                                               \
            conv_dma16(wb, d_ + (PI + (r >> 1)) * ATT_TILE_BYTES + (r & 1) * 4096, woff[r], ld_wso);                     \
        ld_wso += 128;                                                                                                   \
        ld_c0b += 128;                                                                                                   \
        if (ld_c0b == 2 * C) {                                                                                           \
            ld_c0b = 0;                                                                                                  \
            if (++ld_kx == 3) { ld_kx = 0; ++ld_ky; }                                                                    \
            GD_CONV_SET_TAP();                                                                                           \
        }                                                                                                                \
    }
// retire this wave's pieces of every step but the newest KEEP ones, finish its own fragment reads, then meet the other waves: after
// the barrier the retired stage may be read and the stage read last may be overwritten
#define GD_CONV_DMA_SYNC(KEEP)                                                                                           \
    {                                                                                                                    \
        if ((KEEP) == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                     \
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NT) : "memory");                                    \
        __builtin_amdgcn_s_barrier();                                                                                    \
    }
#define GD_CONV_STORE(R, BUF)                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 2 * NT; ++r) *(u32x4*)(lds[BUF][r >> 1] + loff[r & 1]) = R[r]

    f32x16 acc[KI][PI];
#pragma unroll
    for (int j = 0; j < KI; ++j)
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    const FragOffs fo = make_frag_offs(lane);

// all fragment reads of the step first (the MFMAs then wait on them with counted lgkmcnt, not one LDS latency each)
#define GD_CONV_COMPUTE(BUF)                                                                                             \
    {                                                                                                                    \
        V8 wf[4][KI], pf[4][PI];                                                                                         \
        _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                               \
            _Pragma("unroll") for (int j = 0; j < KI; ++j) {                                                             \
                const int b_ = wk * KI + j;                                                                              \
                wf[s4][j] = rd_row<T>(lds[BUF][PI + (b_ >> 1)], fo, b_ & 1, s4);                                         \
            }                                                                                                            \
            _Pragma("unroll") for (int i = 0; i < PI; ++i) {                                                             \
                const int b_ = wp * PI + i;                                                                              \
                pf[s4][i] = rd_row<T>(lds[BUF][b_ >> 1], fo, b_ & 1, s4);                                                \
            }                                                                                                            \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                                 \
            _Pragma("unroll") for (int j = 0; j < KI; ++j)                                                               \
                _Pragma("unroll") for (int i = 0; i < PI; ++i) acc[j][i] = TR::mfma32(wf[s4][j], pf[s4][i], acc[j][i]);  \
    }

#define GD_CONV_COMPUTE_AT(BASE)                                                                                         \
    {                                                                                                                    \
        V8 wf[4][KI], pf[4][PI];                                                                                         \
        _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                               \
            _Pragma("unroll") for (int j = 0; j < KI; ++j) {                                                             \
                const int b_ = wk * KI + j;                                                                              \
                wf[s4][j] = rd_row<T>((BASE) + (PI + (b_ >> 1)) * ATT_TILE_BYTES, fo, b_ & 1, s4);                       \
            }                                                                                                            \
            _Pragma("unroll") for (int i = 0; i < PI; ++i) {                                                             \
                const int b_ = wp * PI + i;                                                                              \
                pf[s4][i] = rd_row<T>((BASE) + (b_ >> 1) * ATT_TILE_BYTES, fo, b_ & 1, s4);                              \
            }                                                                                                            \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                                 \
            _Pragma("unroll") for (int j = 0; j < KI; ++j)                                                               \
                _Pragma("unroll") for (int i = 0; i < PI; ++i) acc[j][i] = TR::mfma32(wf[s4][j], pf[s4][i], acc[j][i]);  \
    }

    if (DMA) {
        // stages rotate: cur is read this step, the step after next is loaded into the stage that was read last step
        char* cur = lds[0][0];
        char* nxt = lds[1][0];
        char* far_ = lds[NS - 1][0];
        GD_CONV_DMA(cur);
        if (NS == 3) {
            if (ns > 1) GD_CONV_DMA(nxt);
            if (ns > 1) { GD_CONV_DMA_SYNC(1); } else { GD_CONV_DMA_SYNC(0); }
#pragma unroll 1
            for (int it = 0; it < ns; ++it) {
                const bool more = it + 2 < ns;
                if (more) GD_CONV_DMA(far_);
                GD_CONV_COMPUTE_AT(cur);
                if (more) { GD_CONV_DMA_SYNC(1); } else { GD_CONV_DMA_SYNC(0); }
                char* const t_ = cur; cur = nxt; nxt = far_; far_ = t_;
            }
        } else {
            GD_CONV_DMA_SYNC(0);
#pragma unroll 1
            for (int it = 0; it < ns; ++it) {
                if (it + 1 < ns) GD_CONV_DMA(nxt);
                GD_CONV_COMPUTE_AT(cur);
                GD_CONV_DMA_SYNC(0);
                char* const t_ = cur; cur = nxt; nxt = t_;
            }
        }
    } else {
    u32x4 R0[2 * NT], R1[2 * NT];
    GD_CONV_LOAD(R0);
    if (ns > 1) GD_CONV_LOAD(R1);
    GD_CONV_STORE(R0, 0);
    if (ns > 2) GD_CONV_LOAD(R0);
    __syncthreads();
    int it = 0;
    // steady state: both loads unconditional (a conditional load makes the compiler wait vmcnt(0) before the LDS stores, i.e. for the
    // loads it has just issued); the last <= 4 steps run in the guarded copy below
#pragma unroll 1
    for (; it + 4 < ns; it += 2) {
        GD_CONV_COMPUTE(0);
        GD_CONV_STORE(R1, 1);
        __syncthreads();
        GD_CONV_LOAD(R1);
        GD_CONV_COMPUTE(1);
        GD_CONV_STORE(R0, 0);
        __syncthreads();
        GD_CONV_LOAD(R0);
    }
#pragma unroll 1
    for (; it + 2 <= ns; it += 2) {
        GD_CONV_COMPUTE(0);
        GD_CONV_STORE(R1, 1);
        __syncthreads();
        if (it + 3 < ns) GD_CONV_LOAD(R1);
        GD_CONV_COMPUTE(1);
        if (it + 2 < ns) GD_CONV_STORE(R0, 0);
        __syncthreads();
        if (it + 4 < ns) GD_CONV_LOAD(R0);
    }
    if (it < ns) { GD_CONV_COMPUTE(0); }
    }
#undef GD_CONV_LOAD
#undef GD_CONV_DMA
#undef GD_CONV_DMA_SYNC
#undef GD_CONV_COMPUTE_AT
#undef GD_CONV_SET_TAP
#undef GD_CONV_STORE
#undef GD_CONV_COMPUTE

    // epilogue: lane = pixel, 16 output channels per block in 4 groups of 4 consecutive ones
#pragma unroll
    for (int ii = 0; ii < PI; ++ii) {
        const int p = p0 + (wp * PI + ii) * 32 + (lane & 31);
        if (p >= a.P) continue;
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int kb = k0 + (wk * KI + j) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = kb + 8 * g + 4 * h;
                if (ch >= a.K) continue;
                if (a.ksplit > 1) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][ii][4 * g + e];
                    *(f32x4*)(a.ws + ((size_t)sp * a.P + p) * a.K + ch) = v;
                } else {
                    f32x4 f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = acc[j][ii][4 * g + e];
                    if (a.bias) {
                        const typename TR::vec4 b = *(const typename TR::vec4*)((const T*)a.bias + ch);
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] += TR::to_f32(b[e]);
                    }
                    if (a.res) {
                        const typename TR::vec4 r = *(const typename TR::vec4*)((const T*)a.res + (size_t)p * a.K + ch);
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] += TR::to_f32(r[e]);
                    }
                    typename TR::vec4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = TR::from_f32(f[e]);
                    *(typename TR::vec4*)((T*)a.out + (size_t)p * a.K + ch) = v;
                }
            }
        }
    }
}

// fold the reduction splits in a fixed order, add the bias, round once: 4 channels per thread
template <typename T>
__global__ void k_conv_fold(const ConvArgs a) {
    using TR = elem_traits<T>;
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tot = (size_t)a.P * a.K / 4;
    if (i4 >= tot) return;
    f32x4 s = *(const f32x4*)(a.ws + i4 * 4);
    for (int sp = 1; sp < a.ksplit; ++sp) {
        const f32x4 t = *(const f32x4*)(a.ws + (size_t)sp * a.P * a.K + i4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += t[e];
    }
    const int ch = (int)((i4 * 4) % a.K);
    if (a.bias) {
        const typename TR::vec4 b = *(const typename TR::vec4*)((const T*)a.bias + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += TR::to_f32(b[e]);
    }
    if (a.res) {
        const typename TR::vec4 r = *(const typename TR::vec4*)((const T*)a.res + i4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += TR::to_f32(r[e]);
    }
    typename TR::vec4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = TR::from_f32(s[e]);
    *(typename TR::vec4*)((T*)a.out + i4 * 4) = v;
}

// configuration of a launch: tile shape (PI, KI) and reduction split; cfg (may be NULL) forces them (tuning: gd_conv3x3_cfg_t)
static void conv_plan(int P, int K, int steps, int* pi, int* ki, int* ksplit, const gd_conv3x3_cfg_t* cfg) {
    if (cfg && cfg->pi > 0) {
        *pi = cfg->pi; *ki = cfg->ki; *ksplit = cfg->ksplit;
        if (*ksplit > steps) *ksplit = steps;
        return;
    }
    // Measured on MI355X over the UNet's shapes at batch 1 and 3 (tools/bench_conv.py --sweep): the best configuration is, almost
    // everywhere, the LARGEST tile for which tiles x splits reaches ~480 workgroups (two per CU) with at least 20 reduction steps
    // left per split; 128-channel tiles need K % 128 == 0 (K = 320 would waste a fifth of them).
    static const int cand[4][2] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}};
    int sp = 1;
    *pi = 1; *ki = 1;
    bool found = false;
    // short reductions (C = 320: 45 steps) whose 64 x 64 tiles already make one to two and a half workgroups per CU: no split, no fold
    // (sweep: 64^2 x 320 at batch 1 22 vs 24 us, 32^2 320 -> 640 at batch 3 23 vs 26 us)
    {
        const long long t11 = (long long)((P + 63) / 64) * ((K + 63) / 64);
        if (steps <= 45 && t11 >= 240 && t11 <= 640) { *ksplit = 1; return; }
    }
    for (int c = 0; c < 4 && !found; ++c) {
        const int cp = cand[c][0], ck = cand[c][1];
        if ((ck == 2 && K % 128 != 0) || (cp == 2 && P < 128)) continue;
        const long long tiles = (long long)((P + 64 * cp - 1) / (64 * cp)) * ((K + 64 * ck - 1) / (64 * ck));
        int s2 = tiles >= 384 ? 1 : (int)(512 / tiles);
        if (s2 > 1 && steps / s2 < 20) continue;
        *pi = cp; *ki = ck; sp = s2; found = true;
    }
    if (!found) {                       // tiny launches: 64 x 64 tiles, ~15 steps per split
        const long long tiles = (long long)((P + 63) / 64) * ((K + 63) / 64);
        sp = steps / 15;
        if ((long long)sp * tiles > 512) sp = (int)(512 / tiles);
    }
    if (sp < 1) sp = 1;
    if (sp > 64) sp = 64;
    *ksplit = sp;
}

static bool conv_cfg_ok(const gd_conv3x3_cfg_t* cfg) {
    return !cfg || cfg->pi <= 0 || ((cfg->pi == 1 || cfg->pi == 2) && (cfg->ki == 1 || cfg->ki == 2) && cfg->ksplit >= 1 && cfg->ksplit <= 64);
}

extern "C" size_t gd_conv3x3_workspace_bytes(int n, int Ho, int Wo, int C, int K, const gd_conv3x3_cfg_t* cfg) {
    int pi, ki, sp;
    const int P = n * Ho * Wo;
    if (P <= 0 || C <= 0 || K <= 0 || C % 64 || !conv_cfg_ok(cfg)) return 0;
    conv_plan(P, K, 9 * C / 64, &pi, &ki, &sp, cfg);
    return sp > 1 ? (size_t)sp * P * K * sizeof(float) : 0;
}

extern "C" int gd_conv3x3(const void* in, const void* w, const void* bias, const void* residual, void* out, int n, int Hi, int Wi, int C, int K, int stride,
                          int upsample, const gd_conv3x3_cfg_t* cfg, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    GD_REQUIRE(in && w && out, GD_EINVAL, "gd_conv3x3: null pointer");
    GD_REQUIRE(conv_cfg_ok(cfg), GD_EINVAL, "gd_conv3x3: cfg: PI, KI in {1, 2}, ksplit in 1..64");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_conv3x3: dtype must be f16/bf16");
    GD_REQUIRE(n > 0 && Hi > 0 && Wi > 0 && C > 0 && K > 0, GD_EINVAL, "gd_conv3x3: bad sizes");
    GD_REQUIRE(C % 64 == 0 && K % 8 == 0, GD_EUNSUPPORTED, "gd_conv3x3: C=%d must be a multiple of 64 and K=%d of 8", C, K);
    GD_REQUIRE((stride == 1 || stride == 2) && (upsample == 0 || (upsample == 1 && stride == 1)), GD_EUNSUPPORTED,
               "gd_conv3x3: stride %d / upsample %d unsupported", stride, upsample);
    GD_REQUIRE((size_t)n * Hi * Wi * C * 2 < 0x7FFFFFFFull && (size_t)K * 9 * C * 2 < 0x7FFFFFFFull, GD_EUNSUPPORTED,
               "gd_conv3x3: operand larger than 2 GiB");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = w; a.bias = bias; a.res = residual; a.out = out;
    a.n = n; a.Hi = Hi; a.Wi = Wi; a.C = C; a.K = K; a.stride = stride; a.up = upsample;
    a.Hv = upsample ? 2 * Hi : Hi; a.Wv = upsample ? 2 * Wi : Wi;
    a.Ho = upsample ? 2 * Hi : (Hi - 1) / stride + 1;          // padding 1, kernel 3
    a.Wo = upsample ? 2 * Wi : (Wi - 1) / stride + 1;
    a.P = n * a.Ho * a.Wo;
    a.cpt = C / 64;
    a.steps = 9 * a.cpt;
    int pi, ki, sp;
    conv_plan(a.P, K, a.steps, &pi, &ki, &sp, cfg);
    a.tiles_p = (a.P + 64 * pi - 1) / (64 * pi);
    a.tiles_k = (K + 64 * ki - 1) / (64 * ki);
    a.spc = (a.steps + sp - 1) / sp;
    a.ksplit = (a.steps + a.spc - 1) / a.spc;                  // no empty split
    a.nwg = a.tiles_p * a.tiles_k * a.ksplit;
    {
        // Which operand an XCD's L2 (4 MB, not shared between the 8 XCDs) should see only a slice of.  An XCD runs a contiguous chunk of
        // the 1-D grid (xcd_remap); with the pixel tile as the slowest index it covers a few pixel tiles x ALL (channel tile, split)
        // pairs, i.e. every XCD streams the whole weight tensor (22-59 MB at 32^2 / 16^2: measured 130 MB of L2 misses for a 34 MB
        // problem); with the pixel tile fastest it covers all pixels x a slice of the weights.  Take the order that fetches less.
        const double I = (double)n * Hi * Wi * C * 2, W = (double)K * 9 * C * 2;
        const double chunk = a.nwg / 8.0, cols = (double)a.tiles_k * a.ksplit, rows = a.tiles_p;
        const double r0 = fmin(rows, ceil(chunk / cols) + 1), w0 = fmin(1.0, chunk / cols);
        const double c1 = fmin(cols, ceil(chunk / rows) + 1), i1 = fmin(1.0, chunk / rows);
        const double t0 = I * r0 / rows + W * w0, t1 = I * i1 + W * c1 / cols;
        a.order = t1 < t0 ? 1 : 0;
    }
    if (a.ksplit > 1) {
        const size_t need = (size_t)a.ksplit * a.P * K * sizeof(float);
        GD_REQUIRE(workspace && workspace_bytes >= need, GD_EWORKSPACE, "gd_conv3x3: workspace %zu B < %zu B", workspace_bytes, need);
        a.ws = (float*)workspace;
    }
    hipStream_t st = as_stream(stream);
    const bool dma = !(cfg && cfg->dma == 0);           // direct-to-LDS staging for the tiles that get three LDS stages with it
#define GD_CONV(PI_, KI_)                                                                      \
    {                                                                                          \
        if (dma && (PI_ + KI_) <= 3) {                                                        \
            if (dtype == GD_F16) k_conv3x3<f16_t, PI_, KI_, true><<<a.nwg, 256, 0, st>>>(a);   \
            else k_conv3x3<bf16_t, PI_, KI_, true><<<a.nwg, 256, 0, st>>>(a);                  \
        } else {                                                                               \
            if (dtype == GD_F16) k_conv3x3<f16_t, PI_, KI_, false><<<a.nwg, 256, 0, st>>>(a);  \
            else k_conv3x3<bf16_t, PI_, KI_, false><<<a.nwg, 256, 0, st>>>(a);                 \
        }                                                                                      \
    }
    if (pi == 2 && ki == 2) GD_CONV(2, 2)
    else if (pi == 2) GD_CONV(2, 1)
    else if (ki == 2) GD_CONV(1, 2)
    else GD_CONV(1, 1)
#undef GD_CONV
    if (a.ksplit > 1) {
        const size_t tot = (size_t)a.P * K / 4;
        const unsigned blocks = (unsigned)((tot + 255) / 256);
        if (dtype == GD_F16) k_conv_fold<f16_t><<<blocks, 256, 0, st>>>(a);
        else k_conv_fold<bf16_t><<<blocks, 256, 0, st>>>(a);
    }
    GD_CHECK_LAUNCH("gd_conv3x3");
    return GD_OK;
}
