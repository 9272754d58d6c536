// R5/R6/R7 — flash-style attention forward on the matrix cores (gfx950 MFMA), D = 64.
//
// Replaces compute_attention (torch.baddbmm + F.softmax, GeoDiffuser/utils/attention_sharing.py:30-47)
// + torch.bmm(attn, v) (GeoDiffuser/utils/attention_processors.py:428,433,549,557,644,647).  The reference
// materialises sim (fp16) and attn (fp32) [f,N,N] in HBM — 168 + 335 MB per 64^2 layer per map; here the
// map never leaves registers.  Up to 4 independent (q,k,v,out) segments run in one launch (vanilla rows,
// edit_out with the warped queries, replace_out) so that small resolutions still fill 256 CUs.
//
// MFMA-bound: algorithmic FLOPs = 4 * BH * N * M * D per launch (QK^T and PV).
#include "attn_common.hpp"

struct FwdArgs {
    gd_attn_seg_t seg[GD_ATTN_MAX_SEGS];
    int bh_end[GD_ATTN_MAX_SEGS];   // exclusive prefix of bh
    int nseg;
    int N, M;
    int tiles;                      // query tiles per (bh)
    int nwg;
    float c;                        // scale * log2(e)
    float scale;
};

template <typename T>
__global__ void __launch_bounds__(256)
k_attn_fwd(const FwdArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) char lds[2][2][ATT_TILE_BYTES];   // [buf][K|V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const int gbh = wg / a.tiles, tile = wg - gbh * a.tiles;
    int sidx = 0;
#pragma unroll
    for (int i = 0; i < GD_ATTN_MAX_SEGS - 1; ++i)
        if (i < a.nseg - 1 && gbh >= a.bh_end[i]) sidx = i + 1;
    const int bh = gbh - (sidx ? a.bh_end[sidx - 1] : 0);
    const gd_attn_seg_t sg = a.seg[sidx];
    const int N = a.N, M = a.M;
    const T* __restrict__ qp = (const T*)sg.q + (size_t)bh * N * ATT_D;
    const T* __restrict__ kp = (const T*)sg.k + (size_t)bh * M * ATT_D;
    const T* __restrict__ vp = (const T*)sg.v + (size_t)bh * M * ATT_D;

    // this lane's query (B operand column); lanes l and l^32 share the query, split d / keys
    const int qrow = tile * ATT_BM + wave * 32 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    V8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const V8*)(qp + (size_t)qld * ATT_D + 16 * s + 8 * h);

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int T_tiles = (M + ATT_BN - 1) / ATT_BN;
    u32x4 kr[2], vr[2];
    tile_load<T>(kp, 0, M, tid, kr);
    tile_load<T>(vp, 0, M, tid, vr);
    tile_store(lds[0][0], tid, kr);
    tile_store(lds[0][1], tid, vr);
    __syncthreads();

    for (int t = 0; t < T_tiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < T_tiles;
        if (more) {
            tile_load<T>(kp, (t + 1) * ATT_BN, M, tid, kr);
            tile_load<T>(vp, (t + 1) * ATT_BN, M, tid, vr);
        }
        const char* lk = lds[cur][0];
        const char* lv = lds[cur][1];

        // S^T tile: 64 keys x 32 queries per wave
        f32x16 s_acc[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s_acc[blk][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) s_acc[blk] = TR::mfma32(read_row_frag<T>(lk, blk, s, lane), qf[s], s_acc[blk]);
        }
        const int kv0 = t * ATT_BN;
        if (kv0 + ATT_BN > M) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kv0 + blk * 32 + acc_key(i, h) >= M) s_acc[blk][i] = -INFINITY;
        }
        float mx = s_acc[0][0];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s_acc[blk][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * a.c);
        const float mc = m_new * a.c;
        float ps = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[blk][i], a.c, -mc));
                s_acc[blk][i] = p;
                ps += p;
            }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }

        // O^T += V^T P^T
        V8 pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) pf[ks] = acc_to_frag<T>(s_acc[ks >> 1], ks & 1);
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) o[dblk] = TR::mfma32(read_tr_frag<T>(lv, dblk, ks, lane), pf[ks], o[dblk]);

        if (more) {
            tile_store(lds[cur ^ 1][0], tid, kr);
            tile_store(lds[cur ^ 1][1], tid, vr);
        }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < N) {
        T* __restrict__ op = (T*)sg.out + ((size_t)bh * N + qrow) * ATT_D;
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(o[dblk][4 * g + j] * inv);
                *(typename TR::vec4*)(op + dblk * 32 + 8 * g + 4 * h) = w;
            }
        if (sg.lse && h == 0) sg.lse[(size_t)bh * N + qrow] = m_run * a.scale + __logf(l_tot);
    }
}

extern "C" int gd_attn_fwd(const gd_attn_seg_t* segs, int nseg, int N, int M, int D, float scale, int dtype, void* stream) {
    GD_REQUIRE(segs && nseg >= 1 && nseg <= GD_ATTN_MAX_SEGS, GD_EINVAL, "gd_attn_fwd: nseg=%d (1..%d)", nseg, GD_ATTN_MAX_SEGS);
    GD_REQUIRE(D == ATT_D, GD_EUNSUPPORTED, "gd_attn_fwd: head dim %d unsupported (only 64)", D);
    GD_REQUIRE(N > 0 && M > 0, GD_EINVAL, "gd_attn_fwd: bad sizes N=%d M=%d", N, M);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_fwd: dtype must be f16/bf16");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    int tot = 0;
    for (int i = 0; i < nseg; ++i) {
        GD_REQUIRE(segs[i].q && segs[i].k && segs[i].v && segs[i].out && segs[i].bh > 0, GD_EINVAL,
                   "gd_attn_fwd: segment %d has a null pointer or bh<=0", i);
        a.seg[i] = segs[i];
        tot += segs[i].bh;
        a.bh_end[i] = tot;
    }
    a.nseg = nseg; a.N = N; a.M = M;
    a.tiles = (N + ATT_BM - 1) / ATT_BM;
    a.nwg = a.tiles * tot;
    a.scale = scale;
    a.c = scale * 1.4426950408889634f;
    if (dtype == GD_F16) k_attn_fwd<f16_t><<<a.nwg, 256, 0, as_stream(stream)>>>(a);
    else k_attn_fwd<bf16_t><<<a.nwg, 256, 0, as_stream(stream)>>>(a);
    GD_CHECK_LAUNCH("gd_attn_fwd");
    return GD_OK;
}

// ---------------------------------------------------------------------------------------------------
// gd_attn_probs: P[bh, r, m] = exp(scale * q[rows[r]] . k[m] - lse[rows[r]])   (16-bit, row stride Mpad)
// Same swapped QK^T tile; the 32 x 64 P tile of each wave is transposed through LDS so that every store
// is a full 128-B row segment.
// ---------------------------------------------------------------------------------------------------
struct ProbsArgs {
    const void* q; const void* k; const float* lse; const int32_t* rows; void* P;
    int N, R, M, Mpad, tiles, nwg;
    float c;      // scale * log2(e)
    float l2e;
};

template <typename T>
__global__ void __launch_bounds__(256)
k_attn_probs(const ProbsArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) char ldsk[2][ATT_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) T stage[4][32][ATT_BN + 8];     // per-wave P tile, padded rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const int bh = wg / a.tiles, tile = wg - bh * a.tiles;
    const int N = a.N, M = a.M, R = a.R;
    const T* __restrict__ qp = (const T*)a.q + (size_t)bh * N * ATT_D;
    const T* __restrict__ kp = (const T*)a.k + (size_t)bh * M * ATT_D;
    T* __restrict__ Pp = (T*)a.P + (size_t)bh * R * a.Mpad;

    const int r_out = tile * ATT_BM + wave * 32 + (lane & 31);
    const int r_c = r_out < R ? r_out : R - 1;
    const int qrow = a.rows ? a.rows[r_c] : r_c;
    V8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const V8*)(qp + (size_t)qrow * ATT_D + 16 * s + 8 * h);
    const float lse2 = a.lse[(size_t)bh * N + qrow] * a.l2e;

    const int T_tiles = (a.Mpad + ATT_BN - 1) / ATT_BN;
    u32x4 kr[2];
    tile_load<T>(kp, 0, M, tid, kr);
    tile_store(ldsk[0], tid, kr);
    __syncthreads();
    for (int t = 0; t < T_tiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < T_tiles;
        if (more) tile_load<T>(kp, (t + 1) * ATT_BN, M, tid, kr);
        const int kv0 = t * ATT_BN;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 s_acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) s_acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) s_acc = TR::mfma32(read_row_frag<T>(ldsk[cur], blk, s, lane), qf[s], s_acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = kv0 + blk * 32 + 8 * g + 4 * h + j;
                    const float p = key < M ? __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[4 * g + j], a.c, -lse2)) : 0.f;
                    w[j] = TR::from_f32(p);
                }
                *(typename TR::vec4*)(&stage[wave][lane & 31][blk * 32 + 8 * g + 4 * h]) = w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // each wave writes its own 32 rows x 64 keys: 8 lanes x 16 B per row, 8 rows per pass
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rr = pass * 8 + (lane >> 3), cc = (lane & 7) * 8;
            const int rg = tile * ATT_BM + wave * 32 + rr;
            if (rg < R && kv0 + cc < a.Mpad)
                *(u32x4*)(Pp + (size_t)rg * a.Mpad + kv0 + cc) = *(const u32x4*)(&stage[wave][rr][cc]);
        }
        if (more) tile_store(ldsk[cur ^ 1], tid, kr);
        __syncthreads();
    }
}

extern "C" int gd_attn_probs(const void* q, const void* k, const float* lse, const int32_t* rows,
                             int BH, int N, int R, int M, int Mpad, int D, float scale, void* P, int dtype, void* stream) {
    GD_REQUIRE(q && k && lse && P, GD_EINVAL, "gd_attn_probs: null pointer");
    GD_REQUIRE(D == ATT_D, GD_EUNSUPPORTED, "gd_attn_probs: head dim %d unsupported (only 64)", D);
    GD_REQUIRE(BH > 0 && N > 0 && R > 0 && M > 0 && Mpad >= M && (Mpad & 7) == 0, GD_EINVAL,
               "gd_attn_probs: bad sizes (Mpad must be a multiple of 8 and >= M)");
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_attn_probs: dtype must be f16/bf16");
    ProbsArgs a;
    a.q = q; a.k = k; a.lse = lse; a.rows = rows; a.P = P;
    a.N = N; a.R = R; a.M = M; a.Mpad = Mpad;
    a.tiles = (R + ATT_BM - 1) / ATT_BM;
    a.nwg = a.tiles * BH;
    a.c = scale * 1.4426950408889634f;
    a.l2e = 1.4426950408889634f;
    if (dtype == GD_F16) k_attn_probs<f16_t><<<a.nwg, 256, 0, as_stream(stream)>>>(a);
    else k_attn_probs<bf16_t><<<a.nwg, 256, 0, as_stream(stream)>>>(a);
    GD_CHECK_LAUNCH("gd_attn_probs");
    return GD_OK;
}
