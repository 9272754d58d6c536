/*
 * ORACLE (test infrastructure, never shipped / never on the product path).
 *
 * Plain-C restatement of the point rasterizer + alpha compositor that the
 * reference's warp calls through pytorch3d:
 *   reference call sites: GeoDiffuser/utils/warp_utils.py:109-113 (rasterize_points),
 *                         GeoDiffuser/utils/warp_utils.py:131-140 (alpha from dist),
 *                         GeoDiffuser/utils/warp_utils.py:156-160 (compositing.alpha_composite)
 *   third-party dependency (absent from /root/reference, not installable here):
 *     pytorch3d @ 89653419d0973396f3eff1a381ba09a07fffc2ed
 *     (GeoDiffuser/envs/requirements.txt:103)
 *
 * PARITY UNPINNED: the algorithm below is restated from pytorch3d's published
 * naive point rasterizer (one queue per pixel) and compositor; no golden vector
 * of pytorch3d itself exists in the reference, so this file is anchored only on
 * the reference's call sites.  Assumptions (DESIGN.md "Splat semantics"):
 *   A1 pixel (row yi, col xi) has its centre at NDC
 *      (1 - (2*xi+1)/S, 1 - (2*yi+1)/S)      (+x left, +y up)
 *   A2 a point hits a pixel iff dist2 < radius^2 (strict) and z >= 0
 *   A3 per pixel the K smallest-z hits are kept and reported in ascending z
 *   A4 hits are visited in ascending packed point index; a full queue evicts
 *      its FIRST slot holding the maximum z, and only for a strictly smaller z;
 *      the final sort is a stable bubble sort on z (ties keep queue-slot order)
 *   A5 empty slots: idx = -1, zbuf = -1, dist2 = -1
 *   A6 idx indexes the packed cloud (n*P + p)
 *   A7 dist2 = fma(dx, dx, dy*dy) in binary32
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float pix_to_ndc(int i, int S) {
    /* -1 + (2 i + 1) / S, evaluated in binary32 in this order */
    const float range = 2.0f;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S;
}

typedef struct { float z; int32_t idx; float d2; } pixq_t;

/* candidate column/row interval (conservative) for a point coordinate */
static inline void cand_range(float p, float r, int S, int* lo, int* hi) {
    /* centre(c) = 1 - (2c+1)/S  =>  c = ((1 - centre) * S - 1) / 2 */
    double a = ((1.0 - ((double)p + (double)r)) * S - 1.0) * 0.5;
    double b = ((1.0 - ((double)p - (double)r)) * S - 1.0) * 0.5;
    if (!(a == a) || !(b == b)) { *lo = 0; *hi = -1; return; }       /* NaN: no pixel */
    if (a < -2.0) a = -2.0;
    if (b > S + 1.0) b = S + 1.0;
    if (b < a) { *lo = 0; *hi = -1; return; }
    int l = (int)floor(a) - 1, h = (int)ceil(b) + 1;
    if (l < 0) l = 0;
    if (h > S - 1) h = S - 1;
    *lo = l; *hi = h;
}

/*
 * points : [B, P, 3] (x, y, z) already in the rasterizer's NDC convention
 * idx    : [B, S, S, K] int32 (packed index b*P + p, or -1)
 * zbuf   : [B, S, S, K], dist2 : [B, S, S, K]
 * returns 0, or -1 on allocation failure
 */
int gd_ref_rasterize_points(const float* points, int B, int P, int S, float radius, int K,
                            int32_t* idx, float* zbuf, float* dist2) {
    const float r2 = radius * radius;
    const size_t npix = (size_t)S * S;
    pixq_t* q = (pixq_t*)malloc(sizeof(pixq_t) * (size_t)(K > 0 ? K : 1));
    int32_t* count = (int32_t*)malloc(sizeof(int32_t) * (npix + 1));
    if (!q || !count) { free(q); free(count); return -1; }

    for (int b = 0; b < B; ++b) {
        const float* pts = points + (size_t)b * P * 3;
        /* pass 1: count candidates per pixel (conservative boxes) */
        memset(count, 0, sizeof(int32_t) * (npix + 1));
        for (int p = 0; p < P; ++p) {
            int c0, c1, r0, r1;
            cand_range(pts[p * 3 + 0], radius, S, &c0, &c1);
            cand_range(pts[p * 3 + 1], radius, S, &r0, &r1);
            for (int r = r0; r <= r1; ++r)
                for (int c = c0; c <= c1; ++c) count[(size_t)r * S + c + 1]++;
        }
        for (size_t i = 0; i < npix; ++i) count[i + 1] += count[i];
        const size_t total = (size_t)count[npix];
        int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (total ? total : 1));
        int32_t* cursor = (int32_t*)malloc(sizeof(int32_t) * npix);
        if (!cand || !cursor) { free(cand); free(cursor); free(q); free(count); return -1; }
        memcpy(cursor, count, sizeof(int32_t) * npix);
        /* pass 2: fill; points visited in ascending index => lists ascending */
        for (int p = 0; p < P; ++p) {
            int c0, c1, r0, r1;
            cand_range(pts[p * 3 + 0], radius, S, &c0, &c1);
            cand_range(pts[p * 3 + 1], radius, S, &r0, &r1);
            for (int r = r0; r <= r1; ++r)
                for (int c = c0; c <= c1; ++c) cand[cursor[(size_t)r * S + c]++] = p;
        }
        /* pass 3: per-pixel queue (A2-A5) */
        for (int yi = 0; yi < S; ++yi) {
            const float yf = pix_to_ndc(S - 1 - yi, S);
            for (int xi = 0; xi < S; ++xi) {
                const float xf = pix_to_ndc(S - 1 - xi, S);
                const size_t pix = (size_t)yi * S + xi;
                int qn = 0, qmax_i = -1;
                float qmax_z = -1000.0f;
                for (int32_t t = count[pix]; t < count[pix + 1]; ++t) {
                    const int p = cand[t];
                    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
                    if (pz < 0) continue;
                    const float dx = xf - px, dy = yf - py;
                    const float d2 = fmaf(dx, dx, dy * dy);
                    if (!(d2 < r2)) continue;
                    if (qn < K) {
                        q[qn].z = pz; q[qn].idx = b * P + p; q[qn].d2 = d2;
                        if (pz > qmax_z) { qmax_z = pz; qmax_i = qn; }
                        qn++;
                    } else if (pz < qmax_z) {
                        q[qmax_i].z = pz; q[qmax_i].idx = b * P + p; q[qmax_i].d2 = d2;
                        qmax_z = pz;
                        for (int i = 0; i < K; ++i)
                            if (q[i].z > qmax_z) { qmax_z = q[i].z; qmax_i = i; }
                    }
                }
                for (int i = 0; i < qn - 1; ++i)          /* stable bubble sort on z */
                    for (int j = 0; j < qn - i - 1; ++j)
                        if (q[j + 1].z < q[j].z) { pixq_t tmp = q[j]; q[j] = q[j + 1]; q[j + 1] = tmp; }
                const size_t o = ((size_t)b * npix + pix) * K;
                for (int k = 0; k < K; ++k) {
                    if (k < qn) { idx[o + k] = q[k].idx; zbuf[o + k] = q[k].z; dist2[o + k] = q[k].d2; }
                    else { idx[o + k] = -1; zbuf[o + k] = -1.0f; dist2[o + k] = -1.0f; }
                }
            }
        }
        free(cand); free(cursor);
    }
    free(q); free(count);
    return 0;
}

/* brute-force variant (every pixel visits every point) used to validate the
 * candidate boxes of the function above on small inputs */
int gd_ref_rasterize_points_bruteforce(const float* points, int B, int P, int S, float radius, int K,
                                       int32_t* idx, float* zbuf, float* dist2) {
    const float r2 = radius * radius;
    pixq_t* q = (pixq_t*)malloc(sizeof(pixq_t) * (size_t)(K > 0 ? K : 1));
    if (!q) return -1;
    for (int b = 0; b < B; ++b) {
        const float* pts = points + (size_t)b * P * 3;
        for (int yi = 0; yi < S; ++yi) {
            const float yf = pix_to_ndc(S - 1 - yi, S);
            for (int xi = 0; xi < S; ++xi) {
                const float xf = pix_to_ndc(S - 1 - xi, S);
                int qn = 0, qmax_i = -1;
                float qmax_z = -1000.0f;
                for (int p = 0; p < P; ++p) {
                    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
                    if (pz < 0) continue;
                    const float dx = xf - px, dy = yf - py;
                    const float d2 = fmaf(dx, dx, dy * dy);
                    if (!(d2 < r2)) continue;
                    if (qn < K) {
                        q[qn].z = pz; q[qn].idx = b * P + p; q[qn].d2 = d2;
                        if (pz > qmax_z) { qmax_z = pz; qmax_i = qn; }
                        qn++;
                    } else if (pz < qmax_z) {
                        q[qmax_i].z = pz; q[qmax_i].idx = b * P + p; q[qmax_i].d2 = d2;
                        qmax_z = pz;
                        for (int i = 0; i < K; ++i)
                            if (q[i].z > qmax_z) { qmax_z = q[i].z; qmax_i = i; }
                    }
                }
                for (int i = 0; i < qn - 1; ++i)
                    for (int j = 0; j < qn - i - 1; ++j)
                        if (q[j + 1].z < q[j].z) { pixq_t tmp = q[j]; q[j] = q[j + 1]; q[j + 1] = tmp; }
                const size_t o = (((size_t)b * S + yi) * S + xi) * K;
                for (int k = 0; k < K; ++k) {
                    if (k < qn) { idx[o + k] = q[k].idx; zbuf[o + k] = q[k].z; dist2[o + k] = q[k].d2; }
                    else { idx[o + k] = -1; zbuf[o + k] = -1.0f; dist2[o + k] = -1.0f; }
                }
            }
        }
    }
    free(q);
    return 0;
}

/*
 * compositing.alpha_composite restated:
 *   out[b, c, y, x] = sum_k feat[c, idx_k] * alpha_k * prod_{j<k} (1 - alpha_j), slots with idx < 0 skipped
 * idx   : [B, K, S, S] int64-valued int32 (packed)   alphas: [B, K, S, S]
 * feat  : [C, Ptot]                                  out   : [B, C, S, S]
 */
int gd_ref_alpha_composite(const int32_t* idx, const float* alphas, const float* feat,
                           int B, int K, int S, int C, int Ptot, float* out) {
    const size_t npix = (size_t)S * S;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (size_t pix = 0; pix < npix; ++pix) {
                float cum = 1.0f, res = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const int32_t n = idx[((size_t)b * K + k) * npix + pix];
                    if (n < 0) continue;
                    const float a = alphas[((size_t)b * K + k) * npix + pix];
                    res = fmaf(cum * a, feat[(size_t)c * Ptot + n], res);
                    cum = cum * (1.0f - a);
                }
                out[((size_t)b * C + c) * npix + pix] = res;
            }
    return 0;
}
