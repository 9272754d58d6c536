// R8 — removal loss (the largest single contraction of the path) and its sparse backward.
//
// Replaces removal_loss_geodiff (GeoDiffuser/utils/attention_processors.py:248-280):
//     corr = bmm(replace_att[:, inpaint rows], base_att^T);  masked row-max x 2;  distance-weighted log ratio.
// The reference materialises corr [f, n_inp, N] and two masked copies; here corr lives only in MFMA accumulators:
//   k_corr_max : C^T[j, r] = sum_m Pb[j, m] Pe[r, m] with the inpaint row r on the LANE, so the masked running
//                max/arg-max over j is lane-local; tiles are combined with one 64-bit atomicMax per (row, mask) on
//                (value bits << 32 | ~j)  => larger value wins, first index wins ties (torch.max on CPU).
//   k_reduce   : unpack, exp(-dist) weight from the analytic pixel-centre distance, scalar loss.
//   k_bwd      : only the two arg-max rows of Pb carry gradient (U/attention_processors.py:259-268); softmax
//                backward of the n_inp rows and the rank-1 updates of dq / dk.
// MFMA-bound (k_corr_max): algorithmic FLOPs = 2 * H * R * N * M.
#include <stdlib.h>
#include "attn_common.hpp"
#include "edit_layer.hpp"

#define CM_T 128   // rows of each operand tile
#define CM_K 64    // k per chunk

struct CorrArgs {
    const void* Pe; const void* Pb; const float* m_inp; const float* m_wo; unsigned long long* best;
    const int32_t* n_valid;       // device scalar: rows [n_valid, R) of Pe are padding slots (not computed, `best` stays 0); NULL: all R
    int H, R, N, Mpad, rtiles, jtiles;
};

template <typename T>
__device__ __forceinline__ void cm_load(const T* __restrict__ base, int row0, int nrows, int k0, int Mpad, int tid, u32x4 (&r)[4]) {
    const int c = tid & 7;
    const bool kin = (k0 + c * 8) < Mpad;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = row0 + (tid >> 3) + 32 * i;
        row = row < nrows ? row : nrows - 1;
        u32x4 z = {0u, 0u, 0u, 0u};
        r[i] = kin ? *(const u32x4*)(base + (size_t)row * Mpad + k0 + c * 8) : z;
    }
}
__device__ __forceinline__ void cm_store(char* lds, int tid, const u32x4 (&r)[4]) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(lds + img_off((tid >> 3) + 32 * i, c)) = r[i];
}

template <typename T>
__global__ void __launch_bounds__(256, 2)
k_corr_max(const CorrArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) char lds[2][2][CM_T * 128];      // [buf][A = Pb rows (j) | B = Pe rows (r)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wj = wave >> 1, wr = wave & 1;
    const int hd = blockIdx.y;
    // The r-tiles of one j-tile read the same 128 rows of Pb (1 MB at M = 4096).  Consecutive block ids land on DIFFERENT XCDs (round
    // robin), each with its own L2, so without a remap every r-tile re-fetched its Pb rows from HBM (PMC: 539 MB fetched per launch against
    // 194 MB of operands).  xcd_remap hands each XCD a contiguous range of logical ids, i.e. all r-tiles of a j-tile, dispatched together.
    const int lid = xcd_remap(blockIdx.x, a.rtiles * a.jtiles);
    const int jt = lid / a.rtiles, rt = lid - jt * a.rtiles;
    const int j0 = jt * CM_T, r0 = rt * CM_T;
    if (a.n_valid && r0 >= a.n_valid[0]) return;
    const T* __restrict__ pb = (const T*)a.Pb + (size_t)hd * a.N * a.Mpad;
    const T* __restrict__ pe = (const T*)a.Pe + (size_t)hd * a.R * a.Mpad;

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;

    const int KC = (a.Mpad + CM_K - 1) / CM_K;
    u32x4 ar[4], br[4];
    cm_load<T>(pb, j0, a.N, 0, a.Mpad, tid, ar);
    cm_load<T>(pe, r0, a.R, 0, a.Mpad, tid, br);
    cm_store(lds[0][0], tid, ar);
    cm_store(lds[0][1], tid, br);
    __syncthreads();
    for (int kc = 0; kc < KC; ++kc) {
        const int cur = kc & 1;
        const bool more = (kc + 1) < KC;
        if (more) {
            cm_load<T>(pb, j0, a.N, (kc + 1) * CM_K, a.Mpad, tid, ar);
            cm_load<T>(pe, r0, a.R, (kc + 1) * CM_K, a.Mpad, tid, br);
        }
        const char* la = lds[cur][0];
        const char* lb = lds[cur][1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            V8 af[2], bf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = read_row_frag<T>(la, wj * 2 + x, s, lane);
                bf[x] = read_row_frag<T>(lb, wr * 2 + x, s, lane);
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = TR::mfma32(af[x], bf[y], acc[x][y]);
        }
        if (more) {
            cm_store(lds[cur ^ 1][0], tid, ar);
            cm_store(lds[cur ^ 1][1], tid, br);
        }
        __syncthreads();
    }
    // masked running max over j (lane-local), r on the lane
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        float b_in = -1.f, b_wo = -1.f;
        int ji = 0, jw = 0;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = j0 + wj * 64 + x * 32 + acc_key(i, h);
                if (j < a.N) {
                    const float v = acc[x][y][i];
                    const float vi = v * a.m_inp[j], vw = v * a.m_wo[j];
                    if (vi > b_in) { b_in = vi; ji = j; }
                    if (vw > b_wo) { b_wo = vw; jw = j; }
                }
            }
        unsigned long long pi = b_in >= 0.f ? (((unsigned long long)__float_as_uint(b_in) << 32) | (0xFFFFFFFFu - (unsigned)ji)) : 0ull;
        unsigned long long pw = b_wo >= 0.f ? (((unsigned long long)__float_as_uint(b_wo) << 32) | (0xFFFFFFFFu - (unsigned)jw)) : 0ull;
        const unsigned long long oi = __shfl_xor(pi, 32, 64), ow = __shfl_xor(pw, 32, 64);
        pi = pi > oi ? pi : oi;
        pw = pw > ow ? pw : ow;
        const int r = r0 + wr * 64 + y * 32 + (lane & 31);
        if (h == 0 && r < a.R) {
            unsigned long long* dst = a.best + ((size_t)hd * a.R + r) * 2;
            atomicMax(dst, pi);
            atomicMax(dst + 1, pw);
        }
    }
}


// ---- k_corr_max2: the same contraction on direct-to-LDS staging -----------------------------------------------------------------
// Round 4.  k_corr_max stages both operands through registers (8 global loads + 8 ds_write_b128 per wave and 64-deep chunk, 13 cycles of
// the LDS store path each) and synchronises once per 16 MFMAs: 0.24-0.34 of the MFMA peak (profiles/r03_bwd_pmc.md: MFMA busy 25.6 %).
// What the 64-query attention kernel learnt (DESIGN 4a') applied to a plain NT GEMM with a lane-local epilogue:
//   * operands arrive by buffer_load ... lds (no staging registers, no ds_write): per 64-deep chunk AJ WJ / 2 + 2 swizzled 64-row tile
//     images (Pb rows, then 2 of Pe), every wave fetching its 8-row pieces of every tile (w64_dma's addressing: the image's XOR moves
//     to the source address; issued through dma_asm, see attn_common.hpp);
//   * three stages in SEPARATE __shared__ arrays, chunk k + 2 issued right after the barrier that retires chunk k - 1; the barrier is
//     the raw s_barrier behind a COUNTED s_waitcnt vmcnt (a __syncthreads() fence drains every direct-to-LDS load in flight,
//     cdna_hip_programming.md "Pipelining across barriers"): one barrier per chunk, two chunks in flight across it;
//   * wave tile (32 AJ) x 64 with 2 WJ waves: AJ + 2 fragment reads per 2 AJ MFMAs.
// Same MFMA operand order and the same k order as k_corr_max: bit-identical sums, hence identical (value, index) pairs (tested).
// Needs Mpad % 64 == 0 and N % (32 AJ WJ) == 0 (the self-attention layers: 4096 / 1024 keys); everything else stays on k_corr_max.
// Ordering of a stage (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"): a wave's pieces of
// chunk k are retired by ITS vmcnt wait at the top of step k, the barrier that follows makes every wave's pieces visible to every
// reader, the reads of step k come after that barrier; a stage is re-issued (chunk k + 2 into chunk k - 1's stage) after the SAME
// barrier, which every wave reaches only once its fragment reads of chunk k - 1 have returned (lgkmcnt(0) in front of the barrier).
template <typename T, int AJ, int WJ>
__global__ void __launch_bounds__(128 * WJ, 1)
k_corr_max2(const CorrArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    constexpr int NTA = AJ * WJ / 2;           // 64-row tile images of Pb per stage
    constexpr int NT = NTA + 2;                // ... and two of Pe
    constexpr int PPT = 4 / WJ;                // 8-row pieces per wave and tile (8 pieces per tile over 2 WJ waves)
    constexpr int PPW = PPT * NT;              // direct-to-LDS pieces per wave and chunk
    __shared__ __attribute__((aligned(16))) char st0[NT * ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) char st1[NT * ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) char st2[NT * ATT_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wj = wave >> 1, wr = wave & 1;
    // Dispatch order: the r-tile is the SLOWEST index — all (head, j-tile) pairs of r-tile 0, then of r-tile 1, ...  The list is padded
    // (n_valid < R) and the r-tiles past it exit at once, but a workgroup that exits still took its turn in the dispatcher's queue:
    // with the dead r-tiles interleaved (grid.x = j-tile x r-tile) 240 live workgroups out of 320 ran as TWO rounds of a one-per-CU
    // kernel (100 us against 60 us for the same 240 launched alone, tools/bench_corr2.py); at the end of the grid they cost nothing.
    // The per-r-tile count U is a multiple of 8, so the workgroups of one (head, j-tile) land on the same XCD for every r-tile (round
    // robin over 8 XCDs) within one round: their reads of the same 256 rows of Pb meet in that XCD's L2.
    const int U = a.jtiles;                                              // = 8 * ceil(j-tiles x heads / 8), set by the launcher
    const int rt = blockIdx.x / U, u = blockIdx.x - rt * U;
    const int jt_n = a.N / (32 * AJ * WJ);
    if (u >= jt_n * a.H) return;
    const int hd = u / jt_n, jt = u - hd * jt_n;
    const int j0 = jt * (32 * AJ * WJ), r0 = rt * CM_T;
    const int nv = a.n_valid ? (a.n_valid[0] < a.R ? a.n_valid[0] : a.R) : a.R;
    if (r0 >= nv) return;
    const bool live = (r0 + wr * 64) < nv;                              // this wave's 64 rows hold at least one live slot
    const int Mpad = a.Mpad;
    const T* __restrict__ pb = (const T*)a.Pb + ((size_t)hd * a.N + j0) * Mpad;
    const T* __restrict__ pe = (const T*)a.Pe + ((size_t)hd * a.R + r0) * Mpad;
    // raw buffers: rows of Pe past the padded list (a last tile of 64 rows) read as zeros
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, 0x7FFFFFFF, 0x00020000);
    const long long be = (long long)(a.R - r0) * Mpad * (long long)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)pe, 0, (int)(be < 0x7FFFFFFF ? be : 0x7FFFFFFF), 0x00020000);
    uint32_t voff[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int row = (PPT * wave + j) * 8 + (lane >> 3);
        const int x = (row >> 1) & 7, g = (x & 2) | ((x & 1) << 2) | ((x >> 2) & 1);
        voff[j] = (uint32_t)(((unsigned)row * (unsigned)Mpad + (unsigned)(((lane & 7) ^ g) * 8)) * (unsigned)sizeof(T));
    }
    const int tB = 64 * Mpad * (int)sizeof(T);                          // bytes from one 64-row tile to the next
    const FragOffs fo = make_frag_offs(lane);

    f32x16 acc[AJ][2];
#pragma unroll
    for (int x = 0; x < AJ; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;

    const int KC = Mpad / CM_K;
    // pieces [P0, P1) of chunk KCH (piece index = tile * PPT + p)
#define CM2_ISSUE(ST, KCH, P0, P1)                                                                           \
    {                                                                                                        \
        const int kb_ = (KCH) * CM_K * (int)sizeof(T);                                                       \
        _Pragma("unroll") for (int q_ = (P0); q_ < (P1); ++q_) {                                             \
            const int t_ = q_ / PPT, p_ = q_ % PPT;                                                          \
            dma_asm(t_ < NTA ? rsA : rsB, (ST) + t_ * ATT_TILE_BYTES + (PPT * wave + p_) * 1024, voff[p_],   \
                    (t_ < NTA ? t_ : t_ - NTA) * tB + kb_);                                                  \
        }                                                                                                    \
    }
#define CM2_FRAGS(ST, S_, B_)                                                                               \
    {                                                                                                        \
        _Pragma("unroll") for (int x_ = 0; x_ < AJ; ++x_) {                                                  \
            const int gb_ = wj * AJ + x_;                                                                    \
            af_[B_][x_] = rd_row<T>((ST) + (gb_ >> 1) * ATT_TILE_BYTES, fo, gb_ & 1, S_);                    \
        }                                                                                                    \
        _Pragma("unroll") for (int y_ = 0; y_ < 2; ++y_) bf_[B_][y_] = rd_row<T>((ST) + (NTA + wr) * ATT_TILE_BYTES, fo, y_, S_); \
    }
    // One chunk.  Top: this wave's pieces of chunk K have landed (chunk K + 1's may still be in flight) and its fragment reads of chunk
    // K - 1 have returned; behind the barrier chunk K + 2 goes into chunk K - 1's stage, a quarter of the wave's pieces in front of each
    // k-step's MFMAs (a piece costs ~60 cycles of issue: spread, not a 12-piece burst in front of the first MFMA).  Fragments of k-step
    // s + 1 are requested before the MFMAs of k-step s are issued (two fragment sets; the sched_barriers pin that order: the scheduler
    // otherwise sinks the reads back to their first use and the LDS latency of every k-step sits in front of its MFMAs).
#define CM2_STEP(CUR, NXT, K)                                                                                \
    if ((K) < KC) {                                                                                          \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");                                \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        /* (past the end: a harmless repeat of the last chunk into a dead stage keeps the piece count of every step the same) */ \
        const int kn_ = (K) + 2 < KC ? (K) + 2 : KC - 1;                                                     \
        if (live) {                                                                                          \
            V8 af_[2][AJ], bf_[2][2];                                                                        \
            CM2_FRAGS(CUR, 0, 0)                                                                             \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                               \
                CM2_ISSUE(NXT, kn_, s_ * PPW / 4, (s_ + 1) * PPW / 4)                                        \
                if (s_ < 3) CM2_FRAGS(CUR, s_ + 1, (s_ + 1) & 1)                                             \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                _Pragma("unroll") for (int x_ = 0; x_ < AJ; ++x_)                                            \
                    _Pragma("unroll") for (int y_ = 0; y_ < 2; ++y_)                                         \
                        acc[x_][y_] = TR::mfma32(af_[s_ & 1][x_], bf_[s_ & 1][y_], acc[x_][y_]);             \
                __builtin_amdgcn_sched_barrier(0);                                                           \
            }                                                                                                \
        } else {                                                                                             \
            CM2_ISSUE(NXT, kn_, 0, PPW)                                                                      \
        }                                                                                                    \
    }
    CM2_ISSUE(st0, 0, 0, PPW)
    CM2_ISSUE(st1, KC > 1 ? 1 : 0, 0, PPW)
#pragma unroll 1
    for (int kc = 0; kc < KC; kc += 3) {
        CM2_STEP(st0, st2, kc)
        CM2_STEP(st1, st0, kc + 1)
        CM2_STEP(st2, st1, kc + 2)
    }
#undef CM2_STEP
#undef CM2_FRAGS
#undef CM2_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the repeats of the last chunk: nothing may land after the wave ends
    if (!live) return;
    // masked running max over j (lane-local), r on the lane — as k_corr_max
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        float b_in = -1.f, b_wo = -1.f;
        int ji = 0, jw = 0;
#pragma unroll
        for (int x = 0; x < AJ; ++x)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = j0 + (wj * AJ + x) * 32 + acc_key(i, h);
                const float v = acc[x][y][i];
                const float vi = v * a.m_inp[j], vw = v * a.m_wo[j];
                if (vi > b_in) { b_in = vi; ji = j; }
                if (vw > b_wo) { b_wo = vw; jw = j; }
            }
        unsigned long long pi = b_in >= 0.f ? (((unsigned long long)__float_as_uint(b_in) << 32) | (0xFFFFFFFFu - (unsigned)ji)) : 0ull;
        unsigned long long pw = b_wo >= 0.f ? (((unsigned long long)__float_as_uint(b_wo) << 32) | (0xFFFFFFFFu - (unsigned)jw)) : 0ull;
        const unsigned long long oi = __shfl_xor(pi, 32, 64), ow = __shfl_xor(pw, 32, 64);
        pi = pi > oi ? pi : oi;
        pw = pw > ow ? pw : ow;
        const int r = r0 + wr * 64 + y * 32 + (lane & 31);
        if (h == 0 && r < a.R) {
            unsigned long long* dst = a.best + ((size_t)hd * a.R + r) * 2;
            atomicMax(dst, pi);
            atomicMax(dst + 1, pw);
        }
    }
}

// which kernel: 0 = k_corr_max (any shape), 24 / 22 = k_corr_max2 with AJ x WJ = 2 x 4 (eight waves of 64 x 64) / 2 x 2
static int corr_variant(int H, int R, int N, int Mpad, int want) {
    if (Mpad % CM_K != 0 || Mpad < 4 * CM_K || want == 1) return 0;
    if (want == 24 && N % 256 == 0) return 24;
    if (want == 22 && N % 128 == 0) return 22;
    // measured (64^2: 5 heads x 4096 x 4096): eight waves of 64 x 64 beat four of 128 x 64 at every list length (44 against 50 us per
    // round); the 32^2 layers (1024 keys: 16 chunks) are launch-bound and fastest on 128 x 128 tiles
    if (N % 256 == 0 && N >= 2048) return 24;
    return (N % 128 == 0) ? 22 : 0;
}

static int corr_max_launch(const void* Pe, const void* Pb, const float* m_inp, const float* m_wo, const int32_t* n_valid_dev,
                           int H, int R, int N, int Mpad, unsigned long long* best, int dtype, void* stream, bool clear, int variant) {
    GD_REQUIRE(Pe && Pb && m_inp && m_wo && best, GD_EINVAL, "gd_removal_corr_max: null pointer");
    GD_REQUIRE(H > 0 && R > 0 && N > 0 && Mpad > 0 && (Mpad & 7) == 0, GD_EINVAL, "gd_removal_corr_max: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_removal_corr_max: dtype must be f16/bf16");
    CorrArgs a;
    a.Pe = Pe; a.Pb = Pb; a.m_inp = m_inp; a.m_wo = m_wo; a.best = best; a.n_valid = n_valid_dev;
    a.H = H; a.R = R; a.N = N; a.Mpad = Mpad;
    a.rtiles = (R + CM_T - 1) / CM_T;
    a.jtiles = (N + CM_T - 1) / CM_T;
    hipStream_t st = as_stream(stream);
    if (clear) gd_zero_async(best, (size_t)H * R * 2 * sizeof(unsigned long long), st);
    const int var = corr_variant(H, R, N, Mpad, variant);
    if (var) {
        const int aj = var / 10, wjn = var % 10;
        a.jtiles = ((N / (32 * aj * wjn)) * H + 7) / 8 * 8;             // (head, j-tile) pairs per r-tile, padded to the 8 XCDs
        dim3 grid2(a.rtiles * a.jtiles, 1);
#define GD_CM2(AJ_, WJ_)                                                                            \
        if (dtype == GD_F16) k_corr_max2<f16_t, AJ_, WJ_><<<grid2, 128 * WJ_, 0, st>>>(a);          \
        else k_corr_max2<bf16_t, AJ_, WJ_><<<grid2, 128 * WJ_, 0, st>>>(a)
        if (var == 24) { GD_CM2(2, 4); } else { GD_CM2(2, 2); }
#undef GD_CM2
        GD_CHECK_LAUNCH("gd_removal_corr_max");
        return GD_OK;
    }
    dim3 grid(a.rtiles * a.jtiles, H);
    if (dtype == GD_F16) k_corr_max<f16_t><<<grid, 256, 0, st>>>(a);
    else k_corr_max<bf16_t><<<grid, 256, 0, st>>>(a);
    GD_CHECK_LAUNCH("gd_removal_corr_max");
    return GD_OK;
}

extern "C" int gd_removal_corr_max(const void* Pe, const void* Pb, const float* m_inp, const float* m_wo, const int32_t* n_valid_dev,
                                   int H, int R, int N, int Mpad, unsigned long long* best, int clear, int variant, int dtype, void* stream) {
    GD_REQUIRE(variant == 0 || variant == 1 || variant == 22 || variant == 24, GD_EINVAL, "gd_removal_corr_max: variant %d (0, 1, 22, 24)", variant);
    return corr_max_launch(Pe, Pb, m_inp, m_wo, n_valid_dev, H, R, N, Mpad, best, dtype, stream, clear != 0, variant);
}

__global__ void k_removal_reduce(const unsigned long long* __restrict__ best, const int32_t* __restrict__ rows,
                                 const int32_t* __restrict__ n_valid, int H, int R, int S,
                                 float* __restrict__ p_in, int32_t* __restrict__ j_in, float* __restrict__ p_wo,
                                 int32_t* __restrict__ j_wo, float* __restrict__ wgt, float* __restrict__ loss_acc) {
    // ONE workgroup walks all H*R rows (a few thousand) in a fixed thread-strided order and folds the loss terms through the wave tree
    // and the four wave sums in index order: bit-reproducible, no floating-point atomics (body shared with gd_edit_losses_fused)
    __shared__ float part[4];
    const float sum = removal_reduce_body(best, rows, n_valid, H, R, S, p_in, j_in, p_wo, j_wo, wgt, part);
    if (threadIdx.x == 0) loss_acc[0] += sum;
}

extern "C" int gd_removal_loss_reduce(const unsigned long long* best, const int32_t* rows, const int32_t* n_valid_dev, int H, int R, int S,
                                      float* p_in, int32_t* j_in, float* p_wo, int32_t* j_wo, float* wgt,
                                      float* loss_acc, void* stream) {
    GD_REQUIRE(best && rows && p_in && j_in && p_wo && j_wo && wgt && loss_acc, GD_EINVAL, "gd_removal_loss_reduce: null pointer");
    GD_REQUIRE(H > 0 && R > 0 && S > 0, GD_EINVAL, "gd_removal_loss_reduce: bad sizes");
    k_removal_reduce<<<1, 256, 0, as_stream(stream)>>>(best, rows, n_valid_dev, H, R, S, p_in, j_in, p_wo, j_wo, wgt, loss_acc);
    GD_CHECK_LAUNCH("gd_removal_loss_reduce");
    return GD_OK;
}

// ---- backward ------------------------------------------------------------------------------------------

struct RmBwdArgs {
    const void* Pe; const void* Pb; const void* q; const void* k; const int32_t* rows;
    const float* p_in; const int32_t* j_in; const float* p_wo; const int32_t* j_wo; const float* wgt;
    const float* m_inp; const float* m_wo; float coef; const float* gscale; const float* gscale2;
    int H, R, N, M, Mpad, D; float scale; float* dq; float* dk; float* ds_ws; const float* rowdot; float* dq_part;
    const int32_t* n_valid;       // device scalar: slots [n_valid, R) of the row list are padding (weight 0) and are skipped; NULL: all R
};

// rowdot[h, r] = sum_m A[h,r,m] * dA[h,r,m]   (one wave per inpaint row; feeds the softmax backward below; body: edit_layer.hpp)
template <typename T>
__global__ void __launch_bounds__(256)
k_removal_rowdot(const RmBwdArgs a, float* __restrict__ rowdot) {
    removal_rowdot_body<T>(a, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), rowdot);
}

#define RM_RB 8      // inpaint rows per wave: each K row fetched from L2 serves 8 rows
#define RM_MCH 512   // keys per wave: the key range is split over waves (more parallelism; per-chunk dq partials folded in order)

template <typename T>
__global__ void __launch_bounds__(256)
k_removal_bwd(const RmBwdArgs a) {
    using TR = elem_traits<T>;
    const int lane = threadIdx.x & 63;
    const int blocks_per_head = (a.R + RM_RB - 1) / RM_RB;
    const int msplit = (a.M + RM_MCH - 1) / RM_MCH;
    const int nch = a.D / ATT_D;                                  // 64-wide slices of the head dim (1 for SD2.1)
    int wb = blockIdx.x * 4 + (threadIdx.x >> 6);                 // one wave per (head, block of RM_RB inpaint rows, key chunk, d slice)
    if (wb >= a.H * blocks_per_head * msplit * nch) return;
    const int dch = wb % nch; wb /= nch;
    const int mc = wb % msplit; wb /= msplit;
    const int m_lo = mc * RM_MCH, m_hi = (m_lo + RM_MCH) < a.M ? (m_lo + RM_MCH) : a.M;
    const int hd = wb / blocks_per_head, r0 = (wb - hd * blocks_per_head) * RM_RB;
    const int nv = a.n_valid ? (a.n_valid[0] < a.R ? a.n_valid[0] : a.R) : a.R;      // live slots of the row list
    if (r0 >= nv) return;
    const float cf = (a.gscale ? a.coef * a.gscale[0] : a.coef) * (a.gscale2 ? a.gscale2[0] : 1.0f);
    const T* __restrict__ kp = (const T*)a.k + (size_t)hd * a.M * a.D + dch * ATT_D;
    const T* pe[RM_RB]; const T* pbw[RM_RB]; const T* pbi[RM_RB];
    float cw[RM_RB], ci[RM_RB], dot[RM_RB], acc[RM_RB];
    int qrow[RM_RB];
#pragma unroll
    for (int i = 0; i < RM_RB; ++i) {
        const int r = (r0 + i) < nv ? (r0 + i) : (nv - 1);
        const bool live = (r0 + i) < nv;
        const int row = hd * a.R + r;
        const int ji = a.j_in[row], jw = a.j_wo[row];
        qrow[i] = a.rows[r];
        cw[i] = live ? -cf * a.wgt[row] * a.m_wo[jw] / (a.p_wo[row] + 1e-4f) : 0.f;
        ci[i] = live ? cf * a.wgt[row] * a.m_inp[ji] / (a.p_in[row] + 1e-4f) : 0.f;
        pe[i] = (const T*)a.Pe + ((size_t)hd * a.R + r) * a.Mpad;
        pbw[i] = (const T*)a.Pb + ((size_t)hd * a.N + jw) * a.Mpad;
        pbi[i] = (const T*)a.Pb + ((size_t)hd * a.N + ji) * a.Mpad;
        dot[i] = a.rowdot[row]; acc[i] = 0.f;
    }
    // dq_i[d = lane] = scale * sum_m dS_i[m] K[m][d];   dS (scaled) optionally stored for the dk reduction
    for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
        const int m = m0 + lane;
        float ds[RM_RB];
#pragma unroll
        for (int i = 0; i < RM_RB; ++i) {
            ds[i] = 0.f;
            if (m < m_hi) {
                const float A = TR::to_f32(pe[i][m]);
                ds[i] = A * (cw[i] * TR::to_f32(pbw[i][m]) + ci[i] * TR::to_f32(pbi[i][m]) - dot[i]) * a.scale;
            }
            if (a.ds_ws && dch == 0 && (r0 + i) < nv && m < m_hi) a.ds_ws[((size_t)hd * a.R + r0 + i) * a.Mpad + m] = ds[i];
        }
        // 16 key rows of K in flight at a time (a load-use chain per key made this loop latency-bound); keys past the chunk
        // end have dS = 0, their (clamped) K rows contribute nothing
#pragma unroll 1
        for (int mb = 0; mb < 64; mb += 16) {
            if (m0 + mb >= m_hi) break;
            float kv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int mk = (m0 + mb + u) < a.M ? (m0 + mb + u) : (a.M - 1);
                kv[u] = TR::to_f32(kp[(size_t)mk * a.D + lane]);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < RM_RB; ++i)
                    acc[i] = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ds[i]), mb + u)), kv[u], acc[i]);
        }
    }
    // per-(key chunk) partial of dq; the chunks of a row are summed in index order by k_removal_dq_fold (no f32 atomics: bit-reproducible)
#pragma unroll
    for (int i = 0; i < RM_RB; ++i)
        if ((r0 + i) < nv) a.dq_part[(((size_t)mc * a.H + hd) * a.R + r0 + i) * a.D + dch * ATT_D + lane] = acc[i];
}

// ---- k_removal_bwd2: the dS K product on the matrix pipe (self-attention layers: head dim 64, full key tiles) -------------------
// k_removal_bwd walks every (row, key) pair on the vector pipe (one v_readlane + one v_fma per key, row and 64 channels: 78 us at 64^2
// with 307 rows, VALU-bound at 47 % busy).  Here the inpaint row sits on the MFMA lane like the query in the attention backward:
//   dS^T[m, r] = A[r, m] (cw_r Pb[j_wo(r), m] + ci_r Pb[j_in(r), m] - dot_r) scale        16-bit, built per lane from three row reads
//   dq^T[d, r] += K^T[d, m] dS^T[m, r]                                                   K^T by the hardware-transposed LDS read
// One workgroup = (head, 128 list slots, one run of RM_MCH keys): four waves of 32 rows share the K tiles, which arrive by
// direct-to-LDS loads (two stages).  The partials keep k_removal_bwd's layout [key run][head][slot][64] f32, so both folds
// (k_removal_dq_fold, gd_edit_dq_fold) stay as they are.  dS is rounded to 16 bits before the product (the vector-pipe kernel keeps it in
// f32): the same rounding the attention backward applies to its dS.
template <typename T>
__global__ void __launch_bounds__(256, 2)
k_removal_bwd2(const RmBwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    using V4 = typename TR::vec4;
    __shared__ __attribute__((aligned(16))) char st0[ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) char st1[ATT_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int msplit = a.M / RM_MCH;                                  // launcher: M % RM_MCH == 0
    const int rblocks = (a.R + 127) / 128;
    int wg = blockIdx.x;
    const int mc = wg % msplit; wg /= msplit;
    const int rb = wg % rblocks; const int hd = wg / rblocks;
    const int nv = a.n_valid ? (a.n_valid[0] < a.R ? a.n_valid[0] : a.R) : a.R;
    if (rb * 128 >= nv) return;
    const int r = rb * 128 + wave * 32 + (lane & 31);
    const bool live = r < nv;
    const int rc = live ? r : nv - 1;
    const int row = hd * a.R + rc;
    const float cf = (a.gscale ? a.coef * a.gscale[0] : a.coef) * (a.gscale2 ? a.gscale2[0] : 1.0f);
    const int ji = a.j_in[row], jw = a.j_wo[row];
    const float cw = live ? -cf * a.wgt[row] * a.m_wo[jw] / (a.p_wo[row] + 1e-4f) : 0.f;
    const float ci = live ? cf * a.wgt[row] * a.m_inp[ji] / (a.p_in[row] + 1e-4f) : 0.f;
    const float dot = a.rowdot[row];
    const int m_lo = mc * RM_MCH;
    const T* __restrict__ pe = (const T*)a.Pe + ((size_t)hd * a.R + rc) * a.Mpad + m_lo + 4 * h;
    const T* __restrict__ pbw = (const T*)a.Pb + ((size_t)hd * a.N + jw) * a.Mpad + m_lo + 4 * h;
    const T* __restrict__ pbi = (const T*)a.Pb + ((size_t)hd * a.N + ji) * a.Mpad + m_lo + 4 * h;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.k + ((size_t)hd * a.M + m_lo) * ATT_D), 0, 0x7FFFFFFF, 0x00020000);
    uint32_t voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rw = (2 * wave + j) * 8 + (lane >> 3);
        const int x = (rw >> 1) & 7, g = (x & 2) | ((x & 1) << 2) | ((x >> 2) & 1);
        voff[j] = (uint32_t)((rw * ATT_D + ((lane & 7) ^ g) * 8) * (int)sizeof(T));
    }
#define RM2_ISSUE(ST, TI)                                                                  \
    {                                                                                      \
        dma_asm(rsK, (ST) + wave * 2048, voff[0], (TI) * ATT_TILE_BYTES);                  \
        dma_asm(rsK, (ST) + wave * 2048 + 1024, voff[1], (TI) * ATT_TILE_BYTES);           \
    }
    constexpr int NTL = RM_MCH / ATT_BN;                              // key tiles per run
    RM2_ISSUE(st0, 0)
    const FragOffs fo = make_frag_offs(lane);
    f32x16 dq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[j][i] = 0.f;
    // this lane's probabilities of one tile: per 16-key step the keys 4 h .. 4 h + 3 and 8 + 4 h .. 8 + 4 h + 3 (the k order of rd_tr)
    V4 ae[8], bw[8], bi[8];
#define RM2_LOAD(TI)                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                           \
        _Pragma("unroll") for (int q2 = 0; q2 < 2; ++q2) {                                                     \
            const int o_ = (TI) * ATT_BN + 16 * ks + 8 * q2;                                                   \
            ae[2 * ks + q2] = *(const V4*)(pe + o_); bw[2 * ks + q2] = *(const V4*)(pbw + o_); bi[2 * ks + q2] = *(const V4*)(pbi + o_); \
        }
    RM2_LOAD(0)
#define RM2_STEP(CUR, NXT, TT)                                                                                \
    {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("" ::: "memory");                                                                        \
        V8 dsf[4];                                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                    \
                const float A_ = TR::to_f32(ae[2 * ks + (j >> 2)][j & 3]);                                     \
                const float g_ = cw * TR::to_f32(bw[2 * ks + (j >> 2)][j & 3]) + ci * TR::to_f32(bi[2 * ks + (j >> 2)][j & 3]); \
                dsf[ks][j] = TR::from_f32(A_ * (g_ - dot) * a.scale);                                          \
            }                                                                                                 \
        if ((TT) + 1 < NTL) { RM2_ISSUE(NXT, (TT) + 1) RM2_LOAD((TT) + 1) }                                   \
        _Pragma("unroll") for (int dblk = 0; dblk < 2; ++dblk)                                                 \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                   \
                dq[dblk] = TR::mfma32(rd_tr<T>((CUR), fo, dblk, ks), dsf[ks], dq[dblk]);                       \
    }
#pragma unroll 1
    for (int t = 0; t < NTL; t += 2) {
        RM2_STEP(st0, st1, t)
        RM2_STEP(st1, st0, t + 1)
    }
#undef RM2_STEP
#undef RM2_LOAD
#undef RM2_ISSUE
    if (!live) return;
    float* __restrict__ pp = a.dq_part + (((size_t)mc * a.H + hd) * a.R + r) * ATT_D;
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = dq[dblk][4 * g + j];
            *(f32x4*)(pp + dblk * 32 + 8 * g + 4 * h) = w;
        }
}

// k_removal_bwd2 where it applies (D = 64, M a multiple of RM_MCH, no dk); k_removal_bwd serves every other shape
static bool rm_bwd2_applies(int M, int D, bool need_dk) { return D == ATT_D && M % RM_MCH == 0 && !need_dk; }
template <typename T>
static void rm_bwd2_launch(const RmBwdArgs& a, hipStream_t st) {
    const int grid = a.H * ((a.R + 127) / 128) * (a.M / RM_MCH);
    k_removal_bwd2<T><<<grid, 256, 0, st>>>(a);
}

// dq[h, rows[r], :] += sum_c dq_part[c, h, r, :]  (c ascending).  Padding slots of the row list (weight 0, contribution exactly 0) are
// skipped, so every live row has exactly one writer.
// T16 != void: the sum is added IN PLACE to a 16-bit gradient (dq16 = T16(float(dq16) + s): the rounding of adding an f32 tensor and
// casting, without the f32 tensor, its zero fill and three element-wise launches).
template <typename T16>
__global__ void k_removal_dq_fold(const float* __restrict__ dq_part, const int32_t* __restrict__ rows, const float* __restrict__ wgt,
                                  int msplit, int H, int R, int N, int D, float* __restrict__ dq, T16* __restrict__ dq16) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= H * R * D) return;
    const int d = gid % D, hr = gid / D;
    if (wgt[hr] == 0.f) return;
    const int hd = hr / R, r = hr - hd * R;
    float s = 0.f;
    for (int c = 0; c < msplit; ++c) s += dq_part[((size_t)c * H * R + hr) * D + d];
    const size_t o = ((size_t)hd * N + rows[r]) * D + d;
    if (dq) dq[o] += s;
    if (dq16) dq16[o] = (T16)((float)dq16[o] + s);
}

// dk[h, m, d] += sum_r dS[h, r, m] * q[h, rows[r], d]   (cross-attention: few keys; one thread per (m, d) of a head)
template <typename T>
__global__ void k_removal_dk(const float* __restrict__ ds_ws, const T* __restrict__ q, const int32_t* __restrict__ rows,
                             const int32_t* __restrict__ n_valid, int R_all, int N, int M, int Mpad, int D, float* __restrict__ dk) {
    const int R = n_valid ? (n_valid[0] < R_all ? n_valid[0] : R_all) : R_all;      // padding slots hold no dS
    const int hd = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= M * D) return;
    const int m = o / D, d = o - m * D;
    const float* dsh = ds_ws + (size_t)hd * R_all * Mpad + m;
    const T* qh = q + (size_t)hd * N * D + d;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int r = 0;
    for (; r + 8 <= R; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[u] = __builtin_fmaf(dsh[(size_t)(r + u) * Mpad], elem_traits<T>::to_f32(qh[(size_t)rows[r + u] * D]), acc[u]);
    }
    for (; r < R; ++r) acc[0] = __builtin_fmaf(dsh[(size_t)r * Mpad], elem_traits<T>::to_f32(qh[(size_t)rows[r] * D]), acc[0]);
    dk[((size_t)hd * M + m) * D + d] += ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

extern "C" size_t gd_removal_bwd_workspace_bytes(int H, int R, int M, int Mpad, int D, int need_dk) {
    const size_t msplit = (size_t)(M + RM_MCH - 1) / RM_MCH;
    return ((size_t)H * R * (1 + msplit * D + (need_dk ? (size_t)Mpad : 0))) * sizeof(float);
}

// dq_f32 / dq16_inout given: the complete backward (row dots, dS K products, fold).  Both NULL: the products only — gd_edit_losses_bwd
// computed the row dots into the workspace and gd_edit_dq_fold folds the partials.
extern "C" int gd_removal_bwd(const gd_removal_bwd_t* rm, float* dq_f32, void* dq16_inout, int dtype, void* stream) {
    GD_REQUIRE(rm && rm->Pe && rm->Pb && rm->q && rm->k && rm->rows && rm->p_in && rm->j_in && rm->p_wo && rm->j_wo && rm->wgt && rm->m_inp && rm->m_wo
               && rm->workspace, GD_EINVAL, "gd_removal_bwd: null pointer (the workspace of gd_removal_bwd_workspace_bytes() is required)");
    GD_REQUIRE(rm->D == 64 || rm->D == 128 || rm->D == 192, GD_EUNSUPPORTED, "gd_removal_bwd: head dim %d unsupported (64, 128, 192)", rm->D);
    GD_REQUIRE(rm->H > 0 && rm->R > 0 && rm->N > 0 && rm->M > 0 && rm->Mpad >= rm->M, GD_EINVAL, "gd_removal_bwd: bad sizes");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_removal_bwd: dtype must be f16/bf16");
    const bool complete = dq_f32 || dq16_inout;
    RmBwdArgs a;
    a.Pe = rm->Pe; a.Pb = rm->Pb; a.q = rm->q; a.k = rm->k; a.rows = rm->rows; a.p_in = rm->p_in; a.j_in = rm->j_in; a.p_wo = rm->p_wo; a.j_wo = rm->j_wo;
    a.wgt = rm->wgt; a.m_inp = rm->m_inp; a.m_wo = rm->m_wo; a.coef = rm->coef; a.gscale = rm->gscale; a.gscale2 = rm->gscale2;
    a.H = rm->H; a.R = rm->R; a.N = rm->N; a.M = rm->M; a.Mpad = rm->Mpad; a.D = rm->D; a.n_valid = rm->n_valid;
    const int H = rm->H, R = rm->R, M = rm->M, D = rm->D;
    // workspace: rowdot [H*R] | dq partials [msplit, H, R, D] | dS [H, R, Mpad] (only with dk_f32)
    const int msplit = (M + RM_MCH - 1) / RM_MCH;
    a.rowdot = rm->workspace;
    a.dq_part = rm->workspace + (size_t)H * R;
    a.scale = rm->scale; a.dq = dq_f32; a.dk = rm->dk_f32;
    a.ds_ws = rm->dk_f32 ? a.dq_part + (size_t)msplit * H * R * D : nullptr;
    const int waves = H * ((R + RM_RB - 1) / RM_RB) * msplit * (D / ATT_D);
    const int blocks = (waves + 3) / 4;
    hipStream_t st = as_stream(stream);
    if (complete) {
        if (dtype == GD_F16) k_removal_rowdot<f16_t><<<(H * R + 3) / 4, 256, 0, st>>>(a, rm->workspace);
        else k_removal_rowdot<bf16_t><<<(H * R + 3) / 4, 256, 0, st>>>(a, rm->workspace);
    }
    if (rm->variant != 1 && rm_bwd2_applies(M, D, rm->dk_f32 != nullptr)) { if (dtype == GD_F16) rm_bwd2_launch<f16_t>(a, st); else rm_bwd2_launch<bf16_t>(a, st); }
    else if (dtype == GD_F16) k_removal_bwd<f16_t><<<blocks, 256, 0, st>>>(a);
    else k_removal_bwd<bf16_t><<<blocks, 256, 0, st>>>(a);
    if (complete) {
        if (dtype == GD_F16) k_removal_dq_fold<f16_t><<<(H * R * D + 255) / 256, 256, 0, st>>>(a.dq_part, rm->rows, rm->wgt, msplit, H, R, rm->N, D, dq_f32, (f16_t*)dq16_inout);
        else k_removal_dq_fold<bf16_t><<<(H * R * D + 255) / 256, 256, 0, st>>>(a.dq_part, rm->rows, rm->wgt, msplit, H, R, rm->N, D, dq_f32, (bf16_t*)dq16_inout);
    }
    if (rm->dk_f32) {
        dim3 grid((M * D + 255) / 256, H);
        if (dtype == GD_F16) k_removal_dk<f16_t><<<grid, 256, 0, st>>>(a.ds_ws, (const f16_t*)rm->q, rm->rows, rm->n_valid, R, rm->N, M, rm->Mpad, D, rm->dk_f32);
        else k_removal_dk<bf16_t><<<grid, 256, 0, st>>>(a.ds_ws, (const bf16_t*)rm->q, rm->rows, rm->n_valid, R, rm->N, M, rm->Mpad, D, rm->dk_f32);
    }
    GD_CHECK_LAUNCH("gd_removal_bwd");
    return GD_OK;
}
