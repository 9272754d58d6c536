// Tile helpers shared by the attention forward / backward / probs kernels (D = 64, 16-bit storage).
//
// Geometry (gfx950, wave = 64 lanes, v_mfma_f32_32x32x16_{f16,bf16}):
//   workgroup = 4 waves = 128 query rows (32 per wave); key/value tile = 64 keys.
//   S^T = K Q^T is computed "swapped": A = K tile (key rows), B = Q^T, so the accumulator has the QUERY
//   on the lane (col = lane & 31) and the 32 keys of a block in its 16 registers x 2 lane halves:
//       key(reg, h) = (reg & 3) + 8 * (reg >> 2) + 4 * h,   h = lane >> 5.
//   Row max / sum are then lane-local plus one exchange with lane ^ 32, and the accumulator converted
//   to 16-bit IS the B operand of the next product (O^T += V^T P^T, dQ^T += K^T dS^T) with no lane
//   movement; the matching A operand (V^T / K^T) is read from the row-major LDS tile with
//   ds_read_b64_tr_b16 (hardware transpose).
//
// LDS image (bytes; one tile = 64 rows x 128 B): the 16-B chunk c of row r is stored at
//       r*128 + ((c ^ g((r >> 1) & 7)) << 4),     g = swap bit0 <-> bit2 of a 3-bit value.
//   * read as an A operand with key rows (ds_read_b128, 16-lane groups cover all 16 combinations of
//     (r & 1, (r >> 1) & 7)): g is a bijection, so the 16 lanes hit 16 distinct 16-B slots of the 256-B bank row;
//   * read transposed (ds_read_b64_tr_b16, 4 key rows x 16 columns per 16 lanes): bit2 of g(.) = (r >> 1) & 1
//     sends rows q and q+2 of a block to different 64-B windows, rows q and q+1 differ in r & 1 (128 B apart),
//     so the four rows of a block land in four distinct 64-B bank windows.
//   One image therefore serves both kinds of read (K in the backward is read both ways).
#pragma once
#include "common.hpp"
#include "splat_common.hpp"

#define ATT_D 64
#define ATT_BM 128   // query rows per workgroup
#define ATT_BN 64    // keys per tile
#define ATT_TILE_BYTES (ATT_BN * ATT_D * 2)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int img_off(int r, int c) {
    const int x = (r >> 1) & 7;
    const int g = (x & 2) | ((x & 1) << 2) | ((x >> 2) & 1);
    return r * 128 + ((c ^ g) << 4);
}

// global -> registers: this thread's two 16-B chunks of a 64 x 64 tile (rows tid/8 and tid/8 + 32)
template <typename T>
__device__ __forceinline__ void tile_load(const T* __restrict__ base, int row0, int nrows, int tid, u32x4 (&r)[2],
                                          int row_stride = ATT_D) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = row0 + (tid >> 3) + 32 * i;
        row = row < nrows ? row : nrows - 1;                 // clamp; out-of-range keys are masked later
        r[i] = *(const u32x4*)(base + (size_t)row * row_stride + c * 8);
    }
}

__device__ __forceinline__ void tile_store(char* lds, int tid, const u32x4 (&r)[2]) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 32 * i;
        *(u32x4*)(lds + img_off(row, c)) = r[i];
    }
}

// A operand (32 key rows x 16 d) of k-step s; key row = blk*32 + (lane & 31)
template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec8 read_row_frag(const char* lds, int blk, int s, int lane) {
    const int r = blk * 32 + (lane & 31);
    const int c = 2 * s + (lane >> 5);
    return *(const typename elem_traits<T>::vec8*)(lds + img_off(r, c));
}

// A operand (32 d rows x 16 keys) = transposed tile (hardware transpose read).
//   dblk: which 32 of the 64 feature columns;  ks: 16-key step (0..3)
//   element j of lane (r = lane & 31, h = lane >> 5) = tile[key = 16*ks + 8*(j>>2) + 4*h + (j&3)][d = 32*dblk + r]
template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec8 read_tr_frag(const char* lds, int dblk, int ks, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
    union { s16x4 v[2]; typename elem_traits<T>::vec8 f; } u;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int key = 16 * ks + 8 * hh + 4 * h + q;
        const int c = dblk * 4 + g1 * 2 + (p >> 1);
        const int off = img_off(key, c) + (p & 1) * 8;
        u.v[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
    }
    return u.f;
}

// accumulator registers 8*s .. 8*s+7 -> 16-bit B fragment of k-step s (k order matches read_tr_frag)
template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec8 acc_to_frag(const f32x16& a, int s) {
    typename elem_traits<T>::vec8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = elem_traits<T>::from_f32(a[8 * s + j]);
    return f;
}

__device__ __forceinline__ int acc_key(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// bijective XCD-aware remap of a 1-D grid: blocks that land on one XCD (id % 8) get a contiguous chunk
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// ---- forward kernels: launch arguments and fragment helpers shared by attn_fwd.hip and attn_fwd_mp.hip ----
#define GD_ATTN_MAX_ORDER 256
struct FwdArgs {
    gd_attn_seg_t seg[GD_ATTN_MAX_SEGS];
    int bh_end[GD_ATTN_MAX_SEGS];   // exclusive prefix of bh
    int nseg;
    int N, M;
    int tiles;                      // query tiles per (bh)
    int nwg;
    float c;                        // scale * log2(e)
    float scale;
    int q_prescaled;                // the queries already carry c (gd_attn_seg_t::q_scaled): scores arrive in the log2 domain
    int lsum;                       // gd_attn_seg_t::q_scaled == 2: row sums over the ROUNDED probabilities (k_attn_fwd_w64 LSUM)
    // pipelined kernels: launch order of the heads.  order[g] = (segment << 12) | head-in-segment for the g-th head of the grid; heads
    // of different segments are interleaved (n_order == 0: segment after segment, via bh_end)
    int n_order;
    unsigned short order[GD_ATTN_MAX_ORDER];
    // segments with a query row list (gd_attn_seg_t::q_rows; cseg = the first one, -1: none): their units (tiles_cs[i] per head of the i-th
    // such segment, cu_end[i] = exclusive prefix of their unit counts) follow the units of all other segments
    int cseg, ncseg, units_full;
    int cu_end[GD_ATTN_MAX_ROWLIST_SEGS], tiles_cs[GD_ATTN_MAX_ROWLIST_SEGS];
    unsigned char cseg_of[GD_ATTN_MAX_ROWLIST_SEGS];
    // split-KV (launches that would leave most CUs idle): split sp handles key tiles [sp*tps, (sp+1)*tps) and leaves an
    // un-normalised partial (O, m, l) in the workspace; k_attn_combine merges them
    int nsplit, tps, tot_bh;
    float* ws_o;                    // [nsplit, tot_bh, N, 64] f32
    float* ws_ml;                   // [nsplit, tot_bh, N, 2]  f32 (reference max, row sum)
    // even split of the key tiles over the workgroups (attn_fwd_mp.hip, SK): the launch's units (head x 128-query tile) x key tiles are
    // one linear range; workgroup w handles tiles [w*sk_tpw, (w+1)*sk_tpw) whatever unit borders fall inside, so every workgroup has the
    // same work (20 heads at 64^2: 640 units on 512 resident workgroups ran 1.57 rounds instead of 1.25).  A unit cut over several
    // workgroups is merged by the workgroup that finishes last: un-normalised (O, reference, row sum) through sk_ws, arrival counter per
    // unit in sk_cnt (zero before the first launch; the merging workgroup leaves it zero).
    int sk_tpw, sk_total;
    int sk_parts;                   // k_attn_fwd_w64: > 0 = the units from sk_unit0 on are cut into sk_parts runs of key tiles, one per workgroup
    int sk_unit0;                   //   (0: every unit — few units; units_full: only the segment with a query row list), the units below it stay whole
    int sk_force;                   // tuning hook: split every launch that can be split (default: where the last round is badly filled)
    int sk_mode;                    // hand-off of the parts: 0 = agent-scope release / acquire fences around the ticket, 1 = device-scope stores and loads only
    f32x4* sk_ws;                   // [2 * nwg] slots of GD_SK_SLOT_F4 float4
    int* sk_cnt;                    // [units]
};
#define GD_SK_SLOT_F4 (4 * 9 * 64)  // 4 waves x (8 chunks of O + 1 of (reference, row sum)) x 64 lanes
#define GD_SK_SLOTS 512             // resident workgroups of 4 waves at two waves per SIMD on 256 CUs
// workspace layout: [2 * GD_SK_SLOTS part slots][arrival counter per unit] — the counters sit at the same offset for every launch shape,
// so a launch can only ever find the zeros its predecessors left there
#define GD_SK_SLOT_BYTES ((size_t)2 * GD_SK_SLOTS * GD_SK_SLOT_F4 * sizeof(f32x4))
size_t gd_attn_sk_workspace_bytes(int tot_bh, int N, int M);

// Per-lane LDS byte offsets of the fragment reads, computed once (the swizzle term does not depend on the k-step /
// key block, which therefore become immediates): krd[s] for the row fragments, vrd[dblk][hh] for the transposed ones.
struct FragOffs { int krd[4]; int vrd[2][2]; };

__device__ __forceinline__ FragOffs make_frag_offs(int lane) {
    FragOffs f;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.krd[s] = img_off(r, 2 * s + h);
    const int g1 = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            f.vrd[dblk][hh] = img_off(8 * hh + 4 * h + q, dblk * 4 + g1 * 2 + (p >> 1)) + (p & 1) * 8;
    return f;
}

template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec8 rd_row(const char* lds, const FragOffs& f, int blk, int s) {
    return *(const typename elem_traits<T>::vec8*)(lds + f.krd[s] + blk * 4096);
}
template <typename T>
__device__ __forceinline__ typename elem_traits<T>::vec8 rd_tr(const char* lds, const FragOffs& f, int dblk, int ks) {
    union { s16x4 v[2]; typename elem_traits<T>::vec8 x; } u;
    u.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + f.vrd[dblk][0] + ks * 2048));
    u.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + f.vrd[dblk][1] + ks * 2048));
    return u.x;
}

// One direct-to-LDS piece: 64 lanes x 16 B = 1 KB, written to lds_dst + 16 * lane (wave-uniform destination; the per-lane SOURCE offset
// carries the image's swizzle: the lane that fills 16-byte slot (row, c') of a tile image fetches global chunk c' ^ g(row) of that row)
__device__ __forceinline__ void w64_dma(const __amdgpu_buffer_rsrc_t rsrc, char* lds_dst, uint32_t voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
#endif
}

// The same piece issued from inline asm: the compiler's wait-count pass then does not know that a load writes LDS at all, so it cannot
// insert its conservative `s_waitcnt vmcnt(0)` in front of fragment reads while pieces of LATER stages are in flight (seen in
// k_corr_max2: one drain per three chunks with the builtin, although the stages are separate __shared__ arrays).  The caller owns the
// ordering: counted `s_waitcnt vmcnt(N)` + raw s_barrier (asm volatile with a memory clobber: never reordered against each other or
// against memory accesses).  M0 = LDS byte address of the destination (what the builtin's expansion sets); M0 is a reserved register
// the compiler only ever sets immediately in front of an instruction that reads it.
typedef __attribute__((address_space(3))) char gd_lds_char;
__device__ __forceinline__ void dma_asm(const __amdgpu_buffer_rsrc_t rsrc, char* lds_dst, uint32_t voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t m0v = (uint32_t)(uintptr_t)(gd_lds_char*)lds_dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
#endif
}

// This lane's query row as the four B fragments of S^T = K Q^T (8 channels at d = 16 s + 8 h).  With warp tables on the segment the row
// is the warped, blended query of U/attention_processors.py:424,544 built here from the <= K splat slots of q_base (same code as
// k_composite_tok: bit-identical to reading a q_warp tensor that gd_splat_composite wrote).
template <typename T>
__device__ __forceinline__ void load_q_frags(const gd_attn_seg_t& sg, const T* __restrict__ qp, int rs, int qld, int h,
                                             typename elem_traits<T>::vec8 (&qf)[4]) {
    using V8 = typename elem_traits<T>::vec8;
    if (sg.warp_idx) {
        const int coff[4] = {8 * h, 16 + 8 * h, 32 + 8 * h, 48 + 8 * h};
        composite_chunks<T, 4>(qp, (size_t)rs, coff, sg.warp_idx, sg.warp_w, sg.warp_m, qld, sg.warp_K, qf);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const V8*)(qp + (size_t)qld * rs + 16 * s + 8 * h);
    }
}


// unit (>= units_full) of a launch -> (segment with a query row list, head in it, query tile of its list)
__device__ __forceinline__ void compact_unit(const FwdArgs& a, int unit, int& sidx, int& bh, int& tile) {
    int uc = unit - a.units_full, ci = 0;
#pragma unroll
    for (int i = 0; i < GD_ATTN_MAX_ROWLIST_SEGS - 1; ++i)
        if (i < a.ncseg - 1 && uc >= a.cu_end[i]) ci = i + 1;
    uc -= ci ? a.cu_end[ci - 1] : 0;
    const int tc = a.tiles_cs[ci];
    bh = uc / tc;
    tile = uc - bh * tc;
    sidx = a.cseg_of[ci];
}

// software-pipelined forward (attn_fwd_mp.hip): QB query blocks x KS key ranges per workgroup; a.seg / bh_end / N / M / c / scale filled in
int gd_attn_fwd_mp_launch(FwdArgs a, int qb, int ks, int dtype, hipStream_t st);
