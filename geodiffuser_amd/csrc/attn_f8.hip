// fp8 (OCP e4m3) attention forward on the block-scaled matrix instruction of gfx950 — BASELINE configs[4] ("fp8 MFMA attention path").
//
// The reference has no such path (its SDXL line is commented out, GeoDiffuser/utils/diffusion.py:106; attention is
// compute_attention + torch.bmm in fp16, GeoDiffuser/utils/attention_sharing.py:30-47).  This is an OPT-IN mode for no-grad passes: q, k, v
// are quantised per head to e4m3 (x * 448 / absmax(head)), both GEMMs run on v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, twice
// the bf16 matrix rate) with unit block scales, the probabilities are re-quantised to e4m3 in registers.  PARITY: oracle-defined
// (oracle/ref_cpu.py: attention_fp8_oracle — the same quantisation emulated with torch.float8_e4m3fn, fp32 everywhere else); the
// quantisation kernels are bit-exact against that emulation, lse is exact to 1e-3, outputs carry the e4m3 rounding of P (3 mantissa bits).
//
// Operand layout of the K = 64 instruction, verified with exact integer data on the device (tools/ub/ub_f8.hip): lane l (r = l & 31,
// h = l >> 5) holds byte j of its 32-byte fragment <-> A[row r][k = 32 h + j] / B[k = 32 h + j][col r]; C / D as every 32x32 MFMA:
// col = r, row = (reg & 3) + 8 (reg >> 2) + 4 h.  v_cvt_pk_fp8_f32 rounds to nearest even and turns |x| >= 480 into NaN (tools/ub/ub_f8.hip),
// like torch's cast; inputs are clamped to +-448 first.
//
//   c = scale log2(e) (aq / 448) (ak / 448) = mant 2^ex (frexp, mant in [0.5, 1)): the mantissa goes into the quantisation of q
//   (q8 = e4m3(q mant 448 / aq)), the power of two into the instruction's block scale, so that
//   E^T block b (32 keys x 32 queries) = 2^ex K8[32 b .. , 0..63] . Q8^T + (3 - m)     one instruction: exponents of 2, ready for v_exp_f32
//   p8 = exp2(E) = 8 exp2(e - m);  P8 = e4m3(p8)   (m: a per-row reference value, see f8_tile)
//   O^T block db (32 d x 32 queries) += VT8[32 db .., 64 keys] . P8^T            one instruction per 64-key tile and d block
// The probability registers of a lane ARE its B fragment of the second product: lane half h owns keys {32 b + (i & 3) + 8 (i >> 2) + 4 h},
// slot j = 16 b + i.  The matching A operand needs V transposed with the keys of a tile in that slot order: gd_quant_fp8_vt writes it so
// (VT8 [BH, M/64, 64 d, 64 slots]).  Both tiles are staged through LDS in 64-byte rows with a 16-byte-chunk XOR swizzle
// (chunk ^ ((row >> 2) & 3)): a quarter wave reading 16 B per lane covers all 64 banks.
#include "attn_common.hpp"

typedef __attribute__((ext_vector_type(8))) int i32x8;

#define F8_MAX 448.0f
#define F8_TILE 4096       // 64 rows x 64 bytes

__device__ __forceinline__ int f8_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ---- per-head absolute maximum: grid (chunks, BH); non-negative floats order like their bit patterns, so the chunk maxima are combined
// with an INTEGER atomicMax (order-independent: bit-reproducible).  amax must be zeroed by the caller of the kernel (the launcher does).
#define F8_AMAX_ROWS 256
struct AmaxArgs { const void* x[3]; int n[3]; float* amax[3]; int heads; };      // up to three tensors per launch (blockIdx.z)

template <typename T>
__global__ void __launch_bounds__(256)
k_absmax_heads(const AmaxArgs a) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    __shared__ float s_part[4];
    const T* __restrict__ x = (const T*)a.x[blockIdx.z];
    const int N = a.n[blockIdx.z], heads = a.heads;
    float* __restrict__ amax = a.amax[blockIdx.z];
    const int bh = blockIdx.y, n0 = blockIdx.x * F8_AMAX_ROWS;
    if (n0 >= N) return;
    const int n1 = (n0 + F8_AMAX_ROWS) < N ? (n0 + F8_AMAX_ROWS) : N;
    const int rs = heads > 0 ? heads * ATT_D : ATT_D;
    const T* base = heads > 0 ? x + (size_t)(bh / heads) * N * rs + (size_t)(bh % heads) * ATT_D : x + (size_t)bh * N * ATT_D;
    float m = 0.f;
    for (int i = n0 * 8 + threadIdx.x; i < n1 * 8; i += 256) {
        const V8 v = *(const V8*)(base + (size_t)(i >> 3) * rs + (i & 7) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(TR::to_f32(v[j])));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = fmaxf(fmaxf(s_part[0], s_part[1]), fmaxf(s_part[2], s_part[3]));
        atomicMax((unsigned int*)amax + bh, __float_as_uint(t));
    }
}

__device__ __forceinline__ float f8_mul(float amax) { return amax > 0.f ? F8_MAX / amax : 1.0f; }
__device__ __forceinline__ float f8_deq(float amax) { return amax > 0.f ? amax / F8_MAX : 1.0f; }
__device__ __forceinline__ float f8_clamp(float x) { return fminf(fmaxf(x, -F8_MAX), F8_MAX); }
// c = scale log2(e) deq(aq) deq(ak) = mant 2^ex with mant in [0.5, 1)   (fp32, this operation order: the oracle repeats it)
__device__ __forceinline__ float f8_c_split(float scale, float aq, float ak, int* ex) {
    const float c = scale * 1.4426950408889634f * f8_deq(aq) * f8_deq(ak);
    return frexpf(c, ex);
}

// ---- rows: out8[bh, n, 0..63] = e4m3(clamp(x * 448 / amax[bh])) ; one thread per 8 channels ---------------------------------------------
// ak != NULL (the query tensor): the multiplier also carries the mantissa of c (see the header comment)
template <typename T>
__device__ __forceinline__ void quant_rows_body(long long g, const T* __restrict__ x, int heads, int N, const float* __restrict__ amax,
                                                const float* __restrict__ ak, float scale, uint8_t* __restrict__ out, long long total8) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    if (g >= total8) return;
    const int c = (int)(g & 7);
    const long long row = g >> 3;
    const int n = (int)(row % N), bh = (int)(row / N);
    const int rs = heads > 0 ? heads * ATT_D : ATT_D;
    const T* src = heads > 0 ? x + ((size_t)(bh / heads) * N + n) * rs + (size_t)(bh % heads) * ATT_D : x + (size_t)row * ATT_D;
    const V8 v = *(const V8*)(src + c * 8);
    float mul = f8_mul(amax[bh]);
    if (ak) {
        int ex;
        mul = mul * f8_c_split(scale, amax[bh], ak[bh], &ex);
    }
    u32x2 w;
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(TR::to_f32(v[0]) * mul), f8_clamp(TR::to_f32(v[1]) * mul), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(TR::to_f32(v[2]) * mul), f8_clamp(TR::to_f32(v[3]) * mul), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(TR::to_f32(v[4]) * mul), f8_clamp(TR::to_f32(v[5]) * mul), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(TR::to_f32(v[6]) * mul), f8_clamp(TR::to_f32(v[7]) * mul), hi, true);
    w[0] = (uint32_t)lo; w[1] = (uint32_t)hi;
    *(u32x2*)(out + (size_t)row * ATT_D + c * 8) = w;
}

template <typename T>
__global__ void k_quant_fp8_rows(const T* __restrict__ x, int heads, int N, const float* __restrict__ amax, const float* __restrict__ ak, float scale,
                                 uint8_t* __restrict__ out, long long total8) {
    quant_rows_body<T>((long long)blockIdx.x * blockDim.x + threadIdx.x, x, heads, N, amax, ak, scale, out, total8);
}

// slot of key kk (0..63) of a tile inside a VT8 row: half h = bit 2 of (kk & 31), i = (kk & 3) + 4 ((kk & 31) >> 3), j = 16 (kk >> 5) + i
__device__ __forceinline__ int f8_slot(int kk) {
    const int b = kk >> 5, k5 = kk & 31;
    return 32 * ((k5 >> 2) & 1) + 16 * b + (k5 & 3) + 4 * (k5 >> 3);
}

// ---- V transposed per 64-key tile: vt8[bh, tile, d, slot(key)] = e4m3(clamp(v[bh, 64 tile + key, d] * 448 / amax[bh])) ---------------------
// One workgroup per (tile, bh): 64 keys x 64 d through LDS.  Keys past M are written as 0.
template <typename T>
__device__ __forceinline__ void quant_vt_body(uint8_t (&s_t)[64][64 + 4], int tile, int bh, const T* __restrict__ v, int heads, int M,
                                              const float* __restrict__ amax, uint8_t* __restrict__ vt8, int tiles) {
    using TR = elem_traits<T>;
    using V8 = typename TR::vec8;
    const int tid = threadIdx.x;
    const int rs = heads > 0 ? heads * ATT_D : ATT_D;
    const T* base = heads > 0 ? v + (size_t)(bh / heads) * M * rs + (size_t)(bh % heads) * ATT_D : v + (size_t)bh * M * ATT_D;
    const float mul = f8_mul(amax[bh]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kk = (tid >> 3) + 32 * i, c = tid & 7;
        const int key = tile * 64 + kk;
        V8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = TR::from_f32(0.f);
        if (key < M) x = *(const V8*)(base + (size_t)key * rs + c * 8);
        const int slot = f8_slot(kk);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const int w = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(TR::to_f32(x[j]) * mul), f8_clamp(TR::to_f32(x[j + 1]) * mul), 0, false);
            s_t[c * 8 + j][slot] = (uint8_t)(w & 0xFF);
            s_t[c * 8 + j + 1][slot] = (uint8_t)((w >> 8) & 0xFF);
        }
    }
    __syncthreads();
    // 64 rows x 64 B out: thread t writes 16 B: row t >> 2, chunk t & 3
    uint8_t* dst = vt8 + ((size_t)bh * tiles + tile) * F8_TILE;
    const int row = tid >> 2, ch = tid & 3;
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        w[j] = (uint32_t)s_t[row][ch * 16 + 4 * j] | ((uint32_t)s_t[row][ch * 16 + 4 * j + 1] << 8) | ((uint32_t)s_t[row][ch * 16 + 4 * j + 2] << 16) |
               ((uint32_t)s_t[row][ch * 16 + 4 * j + 3] << 24);
    *(u32x4*)(dst + row * 64 + ch * 16) = w;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_quant_fp8_vt(const T* __restrict__ v, int heads, int M, const float* __restrict__ amax, uint8_t* __restrict__ vt8, int tiles) {
    __shared__ uint8_t s_t[64][64 + 4];                   // [d][slot], padded rows
    quant_vt_body<T>(s_t, blockIdx.x, blockIdx.y, v, heads, M, amax, vt8, tiles);
}

// q rows | k rows | v tiles in ONE launch (workgroups [0, nq) | [nq, nq + nk) | the rest)
struct QuantArgs {
    const void* q; const void* k; const void* v; const float* aq; const float* ak; const float* av;
    uint8_t* q8; uint8_t* k8; uint8_t* vt8;
    int BH, heads, N, M, tiles, nq, nk; float scale;
};

template <typename T>
__global__ void __launch_bounds__(256)
k_quant_fp8_qkv(const QuantArgs a) {
    __shared__ uint8_t s_t[64][64 + 4];
    const int b = blockIdx.x;
    if (b < a.nq) {
        quant_rows_body<T>((long long)b * 256 + threadIdx.x, (const T*)a.q, a.heads, a.N, a.aq, a.ak, a.scale, a.q8, (long long)a.BH * a.N * 8);
    } else if (b < a.nq + a.nk) {
        quant_rows_body<T>((long long)(b - a.nq) * 256 + threadIdx.x, (const T*)a.k, a.heads, a.M, a.ak, nullptr, 0.f, a.k8, (long long)a.BH * a.M * 8);
    } else {
        const int i = b - a.nq - a.nk;
        quant_vt_body<T>(s_t, i % a.tiles, i / a.tiles, (const T*)a.v, a.heads, a.M, a.av, a.vt8, a.tiles);
    }
}

// ---- the attention forward ----------------------------------------------------------------------------------------------------------------
struct F8Args {
    const uint8_t* q8; const uint8_t* k8; const uint8_t* vt8;       // [BH,N,64], [BH,M,64], [BH,M/64,64,64]
    const float* aq; const float* ak; const float* av;              // per-head absmax [BH]
    void* out; float* lse;                                          // out: head-major [BH,N,64] or token-major [B,N,heads*64] 16-bit; lse [BH,N] or NULL
    int BH, N, M, tiles, nwg, heads;
    float scale;
};

__device__ __forceinline__ i32x8 f8_frag(const char* lds, int row, int h) {        // 32 bytes: chunks 2h, 2h+1 of `row`
    union { i32x8 v; u32x4 q[2]; } u;
    u.q[0] = *(const u32x4*)(lds + f8_off(row, 2 * h));
    u.q[1] = *(const u32x4*)(lds + f8_off(row, 2 * h + 1));
    return u.v;
}

// e4m3 x e4m3; sb: the E8M0 block scale of the B operand in every byte (127 = 2^0), A's scale is 2^0
__device__ __forceinline__ f32x16 f8_mfma(i32x8 a, i32x8 b, f32x16 c, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, sb);
}

// Reference value instead of a running maximum (as in attn_fwd_mp.hip): probabilities are taken against a per-row reference m that starts as
// the exact maximum of the first key tile; P8 = e4m3(8 * 2^(e - m)).  No maximum is computed per tile: a lane adds up its 32 values of the
// tile anyway (the row sum), and as long as that partial sum stays <= 448 every single value is inside the e4m3 range (32 values AT the
// reference add up to 256).  When some lane of the wave exceeds it (a row outgrew its reference by more than a factor ~1.75-56, depending on
// how many keys are that large) all 32 rows of the wave re-reference to their own running maximum — cold path, a handful of times per
// row block.  l, O and lse are exact for any reference.
#define F8_PSHIFT 3.0f          // P8 = e4m3(2^3 p)
#define F8_PSUM_LIMIT 448.0f

__device__ __forceinline__ float f8_probs(const f32x16& s0, const f32x16& s1, i32x8& pv) {
    union { i32x8 v; int w[8]; } pf;
    pf.v = pv;                                         // (v_cvt_pk_fp8_f32 keeps the other half of its destination: reuse the old registers)
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        const float p0 = __builtin_amdgcn_exp2f(s0[i]), p1 = __builtin_amdgcn_exp2f(s0[i + 1]);
        const float p2 = __builtin_amdgcn_exp2f(s0[i + 2]), p3 = __builtin_amdgcn_exp2f(s0[i + 3]);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, pf.w[i >> 2], false);
        pf.w[i >> 2] = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, w, true);
        ps0 += p0 + p2; ps1 += p1 + p3;
    }
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        const float p0 = __builtin_amdgcn_exp2f(s1[i]), p1 = __builtin_amdgcn_exp2f(s1[i + 1]);
        const float p2 = __builtin_amdgcn_exp2f(s1[i + 2]), p3 = __builtin_amdgcn_exp2f(s1[i + 3]);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, pf.w[4 + (i >> 2)], false);
        pf.w[4 + (i >> 2)] = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, w, true);
        ps0 += p0 + p2; ps1 += p1 + p3;
    }
    pv = pf.v;
    return ps0 + ps1;
}

// one 64-key tile of one wave (32 query rows): scores, e4m3 probabilities (+ range check), O update
__device__ __forceinline__ void f8_tile(const char* lk, const char* lv, const i32x8& qv, int sb, int r, int h, bool first,
                                        f32x16 (&o)[2], f32x16& cinit, i32x8& pv, float& m_run, float& l_run) {
    // E = e + (3 - m): the reference enters as the initial accumulator, exp2(E) is 8 p
    f32x16 s0 = f8_mfma(f8_frag(lk, r, h), qv, cinit, sb);
    f32x16 s1 = f8_mfma(f8_frag(lk, 32 + r, h), qv, cinit, sb);
    float ps = first ? INFINITY : f8_probs(s0, s1, pv);
    if (__builtin_amdgcn_ballot_w64(!(ps <= F8_PSUM_LIMIT)) != 0) {
        // cold path (first tile; afterwards only when a lane's partial row sum leaves the range): every row of the wave re-references to
        // its own running maximum, O and l are rescaled, the exponents of this tile re-biased and the probabilities taken again
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[i], s1[i]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float e_mx = first ? mx : mx - F8_PSHIFT + m_run;               // back to the e domain
        const float m_new = fmaxf(m_run, e_mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);           // 0 on the first tile (m_run = -1e30)
        l_run *= alpha;
        const float shift = first ? F8_PSHIFT - m_new : m_run - m_new;       // E_new = E + shift
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; s0[i] += shift; s1[i] += shift; cinit[i] = F8_PSHIFT - m_new; }
        ps = f8_probs(s0, s1, pv);
    }
    l_run += ps;                                                              // row sum of 8 p (the un-rounded probabilities)
    o[0] = f8_mfma(f8_frag(lv, r, h), pv, o[0], 0x7F7F7F7F);
    o[1] = f8_mfma(f8_frag(lv, 32 + r, h), pv, o[1], 0x7F7F7F7F);
}

// NT key tiles per barrier (2 when the tile count is even: twice the prefetch distance and half the barriers — a tile is only four matrix
// instructions here, a single tile per barrier left the global loads of the next one exposed)
template <typename TO, int NT>
__global__ void __launch_bounds__(256, 3)
k_attn_fwd_f8(const F8Args a) {
    using TR = elem_traits<TO>;
    __shared__ __attribute__((aligned(16))) char lds[2][NT][2][F8_TILE];      // [buf][tile][K8 | VT8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
    const int wg = xcd_remap(blockIdx.x, a.nwg);
    const int bh = wg / a.tiles, tile = wg - bh * a.tiles;
    const int N = a.N, M = a.M;
    const uint8_t* __restrict__ qp = a.q8 + (size_t)bh * N * ATT_D;
    const uint8_t* __restrict__ kp = a.k8 + (size_t)bh * M * ATT_D;
    const uint8_t* __restrict__ vp = a.vt8 + (size_t)bh * (M / 64) * F8_TILE;

    const int qrow = tile * ATT_BM + wave * 32 + r;
    const int qld = qrow < N ? qrow : N - 1;
    union { i32x8 v; u32x4 q[2]; } qf;
    qf.q[0] = *(const u32x4*)(qp + (size_t)qld * ATT_D + 32 * h);
    qf.q[1] = *(const u32x4*)(qp + (size_t)qld * ATT_D + 32 * h + 16);
    // q8 carries the mantissa of c (gd_fp8_quant_rows with amax_k); its power of two is the block scale of the query operand
    int ex;
    (void)f8_c_split(a.scale, a.aq[bh], a.ak[bh], &ex);
    int sb = 127 + ex;
    sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
    sb = sb | (sb << 8) | (sb << 16) | (sb << 24);

    f32x16 o[2], cinit;                             // cinit: 3 - m in every register, the initial accumulator of the score products
    i32x8 pv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; cinit[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) pv[i] = 0;
    float m_run = -1.0e30f, l_run = 0.f;            // m_run: the row's reference (exponent-of-2 domain); finite sentinel, it enters cinit

    const int T = M / 64;
    const int lrow = tid >> 2, lch = tid & 3;      // this thread's 16-B chunk of a 64 x 64 B tile
    const size_t loff = (size_t)lrow * 64 + lch * 16;
    u32x4 kr[NT], vr[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        kr[u] = *(const u32x4*)(kp + (size_t)u * F8_TILE + loff);
        vr[u] = *(const u32x4*)(vp + (size_t)u * F8_TILE + loff);
        *(u32x4*)(lds[0][u][0] + f8_off(lrow, lch)) = kr[u];
        *(u32x4*)(lds[0][u][1] + f8_off(lrow, lch)) = vr[u];
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < T; t += NT) {
        const int cur = (t / NT) & 1;
        const bool more = (t + NT) < T;
        if (more) {
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                kr[u] = *(const u32x4*)(kp + (size_t)(t + NT + u) * F8_TILE + loff);
                vr[u] = *(const u32x4*)(vp + (size_t)(t + NT + u) * F8_TILE + loff);
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
            f8_tile(lds[cur][u][0], lds[cur][u][1], qf.v, sb, r, h, t + u == 0, o, cinit, pv, m_run, l_run);
        if (more) {
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                *(u32x4*)(lds[cur ^ 1][u][0] + f8_off(lrow, lch)) = kr[u];
                *(u32x4*)(lds[cur ^ 1][u][1] + f8_off(lrow, lch)) = vr[u];
            }
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);                    // = 8 * sum_k p
    const float inv = f8_deq(a.av[bh]) / l_tot;                               // the 8 of P8 cancels against the 8 in l_tot
    if (qrow < N) {
        size_t off;
        if (a.heads > 0) off = ((size_t)(bh / a.heads) * N + qrow) * (a.heads * ATT_D) + (size_t)(bh % a.heads) * ATT_D;
        else off = ((size_t)bh * N + qrow) * ATT_D;
        TO* __restrict__ op = (TO*)a.out + off;
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename TR::vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = TR::from_f32(o[dblk][4 * g + j] * inv);
                *(typename TR::vec4*)(op + dblk * 32 + 8 * g + 4 * h) = w;
            }
        if (a.lse && h == 0) a.lse[(size_t)bh * N + qrow] = (m_run * 0.6931471805599453f) + __logf(l_tot * 0.125f);
    }
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------------------
extern "C" int gd_fp8_absmax_heads(const void* x, int BH, int heads, int N, float* amax, int dtype, void* stream) {
    GD_REQUIRE(x && amax && BH > 0 && N > 0 && heads >= 0, GD_EINVAL, "gd_fp8_absmax_heads: bad argument");
    GD_REQUIRE(heads == 0 || BH % heads == 0, GD_EINVAL, "gd_fp8_absmax_heads: BH=%d is not a multiple of heads=%d", BH, heads);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_fp8_absmax_heads: dtype must be f16/bf16");
    hipStream_t st = as_stream(stream);
    gd_zero_async(amax, (size_t)BH * sizeof(float), st);
    AmaxArgs aa;
    aa.x[0] = aa.x[1] = aa.x[2] = x; aa.n[0] = aa.n[1] = aa.n[2] = N; aa.amax[0] = aa.amax[1] = aa.amax[2] = amax; aa.heads = heads;
    dim3 grid((N + F8_AMAX_ROWS - 1) / F8_AMAX_ROWS, BH, 1);
    if (dtype == GD_F16) k_absmax_heads<f16_t><<<grid, 256, 0, st>>>(aa);
    else k_absmax_heads<bf16_t><<<grid, 256, 0, st>>>(aa);
    GD_CHECK_LAUNCH("gd_fp8_absmax_heads");
    return GD_OK;
}

extern "C" int gd_fp8_quant_rows(const void* x, int BH, int heads, int N, const float* amax, const float* amax_k, float scale, void* out8,
                                 int dtype, void* stream) {
    GD_REQUIRE(x && amax && out8 && BH > 0 && N > 0 && heads >= 0, GD_EINVAL, "gd_fp8_quant_rows: bad argument");
    GD_REQUIRE(heads == 0 || BH % heads == 0, GD_EINVAL, "gd_fp8_quant_rows: BH=%d is not a multiple of heads=%d", BH, heads);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_fp8_quant_rows: dtype must be f16/bf16");
    const long long total8 = (long long)BH * N * 8;
    const int blocks = (int)((total8 + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) k_quant_fp8_rows<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, heads, N, amax, amax_k, scale, (uint8_t*)out8, total8);
    else k_quant_fp8_rows<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, heads, N, amax, amax_k, scale, (uint8_t*)out8, total8);
    GD_CHECK_LAUNCH("gd_fp8_quant_rows");
    return GD_OK;
}

extern "C" int gd_fp8_quant_vt(const void* v, int BH, int heads, int M, const float* amax, void* vt8, int dtype, void* stream) {
    GD_REQUIRE(v && amax && vt8 && BH > 0 && M > 0 && heads >= 0, GD_EINVAL, "gd_fp8_quant_vt: bad argument");
    GD_REQUIRE(heads == 0 || BH % heads == 0, GD_EINVAL, "gd_fp8_quant_vt: BH=%d is not a multiple of heads=%d", BH, heads);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_fp8_quant_vt: dtype must be f16/bf16");
    const int tiles = (M + 63) / 64;
    dim3 grid(tiles, BH);
    hipStream_t st = as_stream(stream);
    if (dtype == GD_F16) k_quant_fp8_vt<f16_t><<<grid, 256, 0, st>>>((const f16_t*)v, heads, M, amax, (uint8_t*)vt8, tiles);
    else k_quant_fp8_vt<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)v, heads, M, amax, (uint8_t*)vt8, tiles);
    GD_CHECK_LAUNCH("gd_fp8_quant_vt");
    return GD_OK;
}

extern "C" int gd_fp8_quantize_qkv(const void* q, const void* k, const void* v, int BH, int heads, int N, int M, float scale, float* amax3,
                                   void* q8, void* k8, void* vt8, int dtype, void* stream) {
    GD_REQUIRE(q && k && v && amax3 && q8 && k8 && vt8 && BH > 0 && N > 0 && M > 0 && heads >= 0, GD_EINVAL, "gd_fp8_quantize_qkv: bad argument");
    GD_REQUIRE(heads == 0 || BH % heads == 0, GD_EINVAL, "gd_fp8_quantize_qkv: BH=%d is not a multiple of heads=%d", BH, heads);
    GD_REQUIRE(dtype == GD_F16 || dtype == GD_BF16, GD_EINVAL, "gd_fp8_quantize_qkv: dtype must be f16/bf16");
    hipStream_t st = as_stream(stream);
    gd_zero_async(amax3, (size_t)3 * BH * sizeof(float), st);
    AmaxArgs aa;
    aa.x[0] = q; aa.x[1] = k; aa.x[2] = v; aa.n[0] = N; aa.n[1] = M; aa.n[2] = M;
    aa.amax[0] = amax3; aa.amax[1] = amax3 + BH; aa.amax[2] = amax3 + 2 * BH; aa.heads = heads;
    const int nmax = N > M ? N : M;
    dim3 agrid((nmax + F8_AMAX_ROWS - 1) / F8_AMAX_ROWS, BH, 3);
    if (dtype == GD_F16) k_absmax_heads<f16_t><<<agrid, 256, 0, st>>>(aa);
    else k_absmax_heads<bf16_t><<<agrid, 256, 0, st>>>(aa);
    QuantArgs qa;
    qa.q = q; qa.k = k; qa.v = v; qa.aq = amax3; qa.ak = amax3 + BH; qa.av = amax3 + 2 * BH;
    qa.q8 = (uint8_t*)q8; qa.k8 = (uint8_t*)k8; qa.vt8 = (uint8_t*)vt8;
    qa.BH = BH; qa.heads = heads; qa.N = N; qa.M = M; qa.tiles = (M + 63) / 64; qa.scale = scale;
    qa.nq = (int)(((long long)BH * N * 8 + 255) / 256);
    qa.nk = (int)(((long long)BH * M * 8 + 255) / 256);
    const int nblocks = qa.nq + qa.nk + qa.tiles * BH;
    if (dtype == GD_F16) k_quant_fp8_qkv<f16_t><<<nblocks, 256, 0, st>>>(qa);
    else k_quant_fp8_qkv<bf16_t><<<nblocks, 256, 0, st>>>(qa);
    GD_CHECK_LAUNCH("gd_fp8_quantize_qkv");
    return GD_OK;
}

extern "C" int gd_attn_fwd_fp8(const void* q8, const void* k8, const void* vt8, const float* amax_q, const float* amax_k, const float* amax_v,
                               int BH, int heads, int N, int M, int D, float scale, void* out, float* lse, int out_dtype, void* stream) {
    GD_REQUIRE(q8 && k8 && vt8 && amax_q && amax_k && amax_v && out, GD_EINVAL, "gd_attn_fwd_fp8: null pointer");
    GD_REQUIRE(D == ATT_D, GD_EUNSUPPORTED, "gd_attn_fwd_fp8: head dim %d unsupported (only 64)", D);
    GD_REQUIRE(BH > 0 && N > 0 && M > 0 && M % 64 == 0, GD_EUNSUPPORTED, "gd_attn_fwd_fp8: the key count must be a multiple of 64 (M=%d)", M);
    GD_REQUIRE(heads >= 0 && (heads == 0 || BH % heads == 0), GD_EINVAL, "gd_attn_fwd_fp8: BH=%d is not a multiple of heads=%d", BH, heads);
    GD_REQUIRE(out_dtype == GD_F16 || out_dtype == GD_BF16, GD_EINVAL, "gd_attn_fwd_fp8: out dtype must be f16/bf16");
    F8Args a;
    a.q8 = (const uint8_t*)q8; a.k8 = (const uint8_t*)k8; a.vt8 = (const uint8_t*)vt8; a.aq = amax_q; a.ak = amax_k; a.av = amax_v;
    a.out = out; a.lse = lse; a.BH = BH; a.N = N; a.M = M; a.heads = heads; a.scale = scale;
    a.tiles = (N + ATT_BM - 1) / ATT_BM;
    a.nwg = a.tiles * BH;
    hipStream_t st = as_stream(stream);
    const bool two = (M / 64) % 2 == 0;
    if (out_dtype == GD_F16) {
        if (two) k_attn_fwd_f8<f16_t, 2><<<a.nwg, 256, 0, st>>>(a);
        else k_attn_fwd_f8<f16_t, 1><<<a.nwg, 256, 0, st>>>(a);
    } else {
        if (two) k_attn_fwd_f8<bf16_t, 2><<<a.nwg, 256, 0, st>>>(a);
        else k_attn_fwd_f8<bf16_t, 1><<<a.nwg, 256, 0, st>>>(a);
    }
    GD_CHECK_LAUNCH("gd_attn_fwd_fp8");
    return GD_OK;
}
