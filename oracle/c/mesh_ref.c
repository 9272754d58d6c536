/*
 * ORACLE (test infrastructure, never shipped / never on the product path).
 *
 * Plain-C restatement of the mesh coverage that the reference's geometry pre-pass gets from pytorch3d:
 *   reference call sites: GeoDiffuser/utils/warp_utils.py:247 (rasterize_meshes(mesh, S, blur_radius = 1e-6 / (2 S),
 *                         faces_per_pixel, perspective_correct=True)), :252-266 (Fragments -> sample_textures, slot 0 only),
 *                         :364-399 (get_mesh: vertices (-x, -y, Z) of the masked pixels, two triangles per 2x2 quad)
 *   third-party dependency (absent from /root/reference, not installable here):
 *     pytorch3d @ 89653419d0973396f3eff1a381ba09a07fffc2ed (GeoDiffuser/envs/requirements.txt:103)
 *
 * PARITY UNPINNED: restated from pytorch3d's published naive mesh rasterizer (rasterize_meshes_cpu / RasterizeMeshesNaive); the
 * reference holds no golden vector of pytorch3d.  The vertex texture is identically 1 (warp_utils.py:385-388) and only face slot 0 is
 * read (:266), so the output is 1 wherever ANY face is accepted for the pixel, 0 elsewhere.  Per pixel (row yi, col xi) and face:
 *   M1 pixel centre (xf, yf) = (pix_to_ndc(S-1-xi), pix_to_ndc(S-1-yi)),  pix_to_ndc(i) = -1 + (2 i + 1) / S     (+x left, +y up)
 *   M2 faces with |area| <= 1e-8 are skipped, area = edge(v2; v0, v1), edge(p; a, b) = (p.x-a.x)(b.y-a.y) - (p.y-a.y)(b.x-a.x)
 *   M3 barycentrics w_i = edge(p; v_{i+1}, v_{i+2}) / (area + 1e-8), then the perspective correction
 *        w0' = w0 z1 z2, w1' = z0 w1 z2, w2' = z0 z1 w2, each divided by max(w0' + w1' + w2', 1e-8)
 *   M4 depth pz = w0' z0 + w1' z1 + w2' z2; faces with pz < 0 are skipped
 *   M5 the face is accepted iff the pixel is inside (w0', w1', w2' all > 0) or its squared distance to the triangle's
 *      boundary (minimum over the three edge SEGMENTS) is < blur_radius
 * Everything in binary32, no fused multiply-add (-ffp-contract=off): the HIP kernel evaluates the same expressions in the same order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float m_pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

static inline float m_edge(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

/* squared distance from p to the segment a-b (pytorch3d PointLineDistanceForward) */
static inline float m_seg_d2(float px, float py, float ax, float ay, float bx, float by) {
    const float abx = bx - ax, aby = by - ay;
    const float l2 = abx * abx + aby * aby;
    if (l2 <= 1e-8f) return (px - bx) * (px - bx) + (py - by) * (py - by);
    float t = (abx * (px - ax) + aby * (py - ay)) / l2;
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    const float qx = ax + t * abx, qy = ay + t * aby;
    return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

int gd_ref_mesh_face_accepts(float px, float py, const float* v0, const float* v1, const float* v2, float blur) {
    const float area = m_edge(v2[0], v2[1], v0[0], v0[1], v1[0], v1[1]);
    if (area <= 1e-8f && area >= -1e-8f) return 0;
    const float den = area + 1e-8f;
    const float w0 = m_edge(px, py, v1[0], v1[1], v2[0], v2[1]) / den;
    const float w1 = m_edge(px, py, v2[0], v2[1], v0[0], v0[1]) / den;
    const float w2 = m_edge(px, py, v0[0], v0[1], v1[0], v1[1]) / den;
    const float z0 = v0[2], z1 = v1[2], z2 = v2[2];
    const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
    float dn = t0 + t1 + t2;
    dn = dn > 1e-8f ? dn : 1e-8f;
    const float b0 = t0 / dn, b1 = t1 / dn, b2 = t2 / dn;
    const float pz = b0 * z0 + b1 * z1 + b2 * z2;
    if (pz < 0.0f) return 0;
    const int inside = b0 > 0.0f && b1 > 0.0f && b2 > 0.0f;
    if (inside) return 1;
    float d = m_seg_d2(px, py, v0[0], v0[1], v1[0], v1[1]);
    const float d1 = m_seg_d2(px, py, v0[0], v0[1], v2[0], v2[1]);
    const float d2 = m_seg_d2(px, py, v1[0], v1[1], v2[0], v2[1]);
    d = d1 < d ? d1 : d;
    d = d2 < d ? d2 : d;
    return d < blur;
}

/* verts [V,3] (x, y already in the rasterizer's convention, z depth), faces [F,3], out [S,S] in {0,1}.  O(F * bbox): each face
 * visits the pixels of its bounding box grown by sqrt(blur) (a superset of the pixels it can accept: the test itself is exact). */
int gd_ref_mesh_coverage(const float* verts, const int32_t* faces, int V, int F, int S, float* out) {
    (void)V;
    memset(out, 0, (size_t)S * S * sizeof(float));
    const float blur = 1e-6f / (float)(2 * S);
    const float grow = sqrtf(blur);
    for (int f = 0; f < F; ++f) {
        const float* v0 = verts + 3 * (size_t)faces[3 * f];
        const float* v1 = verts + 3 * (size_t)faces[3 * f + 1];
        const float* v2 = verts + 3 * (size_t)faces[3 * f + 2];
        const float xmin = fminf(v0[0], fminf(v1[0], v2[0])) - grow, xmax = fmaxf(v0[0], fmaxf(v1[0], v2[0])) + grow;
        const float ymin = fminf(v0[1], fminf(v1[1], v2[1])) - grow, ymax = fmaxf(v0[1], fmaxf(v1[1], v2[1])) + grow;
        /* pixel column c has centre 1 - (2c+1)/S  =>  c = ((1 - x) S - 1) / 2; one extra pixel of margin on each side */
        int c0 = (int)floorf(((1.0f - xmax) * (float)S - 1.0f) * 0.5f) - 1, c1 = (int)ceilf(((1.0f - xmin) * (float)S - 1.0f) * 0.5f) + 1;
        int r0 = (int)floorf(((1.0f - ymax) * (float)S - 1.0f) * 0.5f) - 1, r1 = (int)ceilf(((1.0f - ymin) * (float)S - 1.0f) * 0.5f) + 1;
        c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0; c1 = c1 > S - 1 ? S - 1 : c1; r1 = r1 > S - 1 ? S - 1 : r1;
        for (int r = r0; r <= r1; ++r) {
            const float py = m_pix_to_ndc(S - 1 - r, S);
            for (int c = c0; c <= c1; ++c) {
                if (out[(size_t)r * S + c] != 0.0f) continue;
                const float px = m_pix_to_ndc(S - 1 - c, S);
                if (gd_ref_mesh_face_accepts(px, py, v0, v1, v2, blur)) out[(size_t)r * S + c] = 1.0f;
            }
        }
    }
    return 0;
}

/* brute force over every (pixel, face) pair: validates the bounding-box walk above on small cases */
int gd_ref_mesh_coverage_bruteforce(const float* verts, const int32_t* faces, int V, int F, int S, float* out) {
    (void)V;
    memset(out, 0, (size_t)S * S * sizeof(float));
    const float blur = 1e-6f / (float)(2 * S);
    for (int r = 0; r < S; ++r)
        for (int c = 0; c < S; ++c) {
            const float px = m_pix_to_ndc(S - 1 - c, S), py = m_pix_to_ndc(S - 1 - r, S);
            for (int f = 0; f < F; ++f)
                if (gd_ref_mesh_face_accepts(px, py, verts + 3 * (size_t)faces[3 * f], verts + 3 * (size_t)faces[3 * f + 1],
                                             verts + 3 * (size_t)faces[3 * f + 2], blur)) {
                    out[(size_t)r * S + c] = 1.0f;
                    break;
                }
        }
    return 0;
}
