// Device pieces shared by the launches of one hooked optimisation-pass layer (~12 launches per layer):
//   gd_edit_losses_fwd (wv != NULL) = k_losses_fwd's reductions + [last workgroup: removal reduce + fold + assemble]
//   gd_edit_losses_bwd (rm != NULL) = the loss backward  U  k_removal_rowdot (independent work, one grid)
//   gd_edit_dq_fold                 = attention dq partials + removal dq partials -> the 16-bit gradient
// The bodies below are the ones the stand-alone stages run (same arithmetic, same summation order: bit for bit the same results).
#pragma once
#include "common.hpp"

// pixel-centre distance of CoordinateDistances (U/generic_torch.py:126-140): centres (2i+1)/S - 1
__device__ __forceinline__ float pix_dist(int a, int b, int S) {
    const int ya = a / S, xa = a - ya * S, yb = b / S, xb = b - yb * S;
    const float dx = (float)(2 * (xa - xb)) / (float)S, dy = (float)(2 * (ya - yb)) / (float)S;
    return sqrtf(dx * dx + dy * dy + 1e-12f);
}

// gd_removal_loss_reduce's body for ONE 256-thread workgroup: unpack `best`, distance weight, the loss sum (returned in thread 0, folded
// through the wave tree and the four wave sums in index order: bit-reproducible).  `part`: 4 floats of LDS.
__device__ __forceinline__ float removal_reduce_body(const unsigned long long* __restrict__ best, const int32_t* __restrict__ rows,
                                                     const int32_t* __restrict__ n_valid, int H, int R, int S,
                                                     float* __restrict__ p_in, int32_t* __restrict__ j_in, float* __restrict__ p_wo,
                                                     int32_t* __restrict__ j_wo, float* __restrict__ wgt, float* part) {
    float term = 0.f;
    for (int i = threadIdx.x; i < H * R; i += blockDim.x) {
        const unsigned long long bi = best[(size_t)i * 2], bw = best[(size_t)i * 2 + 1];
        // best == 0: no correlation value of this row compared greater than the initial -1, i.e. every one of them was NaN (diverged
        // latents).  torch.max would return NaN there; do the same for the value and keep the INDEX valid — the backward addresses
        // rows of Pb with it (an index of -1 here was an out-of-bounds read).
        const float qnan = __uint_as_float(0x7FC00000u);
        const float pi = bi ? __uint_as_float((unsigned)(bi >> 32)) : qnan, pw = bw ? __uint_as_float((unsigned)(bw >> 32)) : qnan;
        int ji = bi ? (int)(0xFFFFFFFFu - (unsigned)(bi & 0xFFFFFFFFu)) : 0, jw = bw ? (int)(0xFFFFFFFFu - (unsigned)(bw & 0xFFFFFFFFu)) : 0;
        ji = ji < 0 ? 0 : (ji >= S * S ? S * S - 1 : ji);
        jw = jw < 0 ? 0 : (jw >= S * S ? S * S - 1 : jw);
        const int r = i % R;
        // slots r >= n_valid are padding (row list rounded up so that launch dimensions repeat across edits): weight 0 removes
        // them from the loss and, through wgt, from every term of the backward
        const bool live = !n_valid || r < n_valid[0];
        const float w = live ? __expf(-pix_dist(rows[r], jw, S)) : 0.f;
        p_in[i] = pi; j_in[i] = ji; p_wo[i] = pw; j_wo[i] = jw; wgt[i] = w;
        term += live ? w * (-__logf(pw + 1e-4f) + __logf(pi + 1e-4f)) : 0.f;
    }
    term = wave_sum(term);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = term;
    __syncthreads();
    return (part[0] + part[1]) + (part[2] + part[3]);
}

// gd_loss_assemble's arithmetic (one thread)
__device__ __forceinline__ void loss_assemble_body(const float* sums, float rm, const float* __restrict__ inv5, const float* __restrict__ inv_rm,
                                                   const float* __restrict__ wv, const float* __restrict__ inv5_bwd, int use_amodal,
                                                   float* __restrict__ out) {
    float t[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = sums[i] * inv5[i];
    const float l_rm = rm * inv_rm[0];
    float terms[5] = {t[0], t[1], l_rm, t[3] + t[4], use_amodal ? t[2] : t[1] * 0.0f};
    float loss = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { out[i] = terms[i]; loss += terms[i] * wv[i]; }
    out[5] = loss;
    const int perm[5] = {0, 1, 4, 3, 3};
#pragma unroll
    for (int i = 0; i < 5; ++i) out[6 + i] = wv[perm[i]] * inv5_bwd[i];
    out[11] = wv[2] * inv_rm[0];
}

// k_removal_rowdot's body for one wave: rowdot[h, r] = sum_m A[h,r,m] * dA[h,r,m]
template <typename T, typename A>
__device__ __forceinline__ void removal_rowdot_body(const A& a, int row, int lane, float* __restrict__ rowdot) {
    using TR = elem_traits<T>;
    if (row >= a.H * a.R) return;
    const int hd = row / a.R, r = row - hd * a.R;
    if (a.n_valid && r >= a.n_valid[0]) return;
    const float cf = (a.gscale ? a.coef * a.gscale[0] : a.coef) * (a.gscale2 ? a.gscale2[0] : 1.0f);
    const int ji = a.j_in[row], jw = a.j_wo[row];
    const float cw = -cf * a.wgt[row] * a.m_wo[jw] / (a.p_wo[row] + 1e-4f);
    const float ci = cf * a.wgt[row] * a.m_inp[ji] / (a.p_in[row] + 1e-4f);
    const T* __restrict__ pe = (const T*)a.Pe + ((size_t)hd * a.R + r) * a.Mpad;
    const T* __restrict__ pbw = (const T*)a.Pb + ((size_t)hd * a.N + jw) * a.Mpad;
    const T* __restrict__ pbi = (const T*)a.Pb + ((size_t)hd * a.N + ji) * a.Mpad;
    // 8 keys per lane and step (16-byte loads; the padding columns M .. Mpad of a probability row are zeros, gd_attn_probs)
    using V8 = typename TR::vec8;
    float dot = 0.f;
    for (int m = lane * 8; m < a.Mpad; m += 64 * 8) {
        const V8 e8 = *(const V8*)(pe + m), w8 = *(const V8*)(pbw + m), i8 = *(const V8*)(pbi + m);
#pragma unroll
        for (int j = 0; j < 8; ++j) dot = __builtin_fmaf(TR::to_f32(e8[j]), cw * TR::to_f32(w8[j]) + ci * TR::to_f32(i8[j]), dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) rowdot[row] = dot;
}
