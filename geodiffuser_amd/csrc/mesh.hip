// R14 (setup, once per edit) — coverage mask of the transformed object surface mesh ("amodal mask").
//
// Replaces pytorch3d.rasterize_meshes(blur_radius = 1e-6 / (2 S), perspective_correct=True) + TexturesVertex.sample_textures as used
// by splatter_mesh (GeoDiffuser/utils/warp_utils.py:235-298): the texture is identically 1 and only face slot 0 is read (:266), so the
// result is 1 wherever ANY face is accepted for the pixel.  PARITY UNPINNED (pytorch3d is absent): the per-(pixel, face) test is the
// published naive mesh rasterizer's, restated in oracle/c/mesh_ref.c (rules M1-M5 there) — this kernel evaluates the SAME binary32
// expressions in the same order (no fused multiply-add), so it is bit-identical to that oracle.
//   accepted  <=>  |area| > 1e-8,  perspective-corrected depth >= 0,  and (all corrected barycentrics > 0  or  squared distance to
//                  the triangle's boundary < blur_radius)
// One thread per face walks the pixels of the face's bounding box grown by sqrt(blur) (+1 pixel margin); stores of 1.0f race benignly.
#include "common.hpp"

#pragma clang fp contract(off)

__device__ __forceinline__ float mesh_pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

__device__ __forceinline__ float mesh_edge(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float mesh_seg_d2(float px, float py, float ax, float ay, float bx, float by) {
    const float abx = bx - ax, aby = by - ay;
    const float l2 = abx * abx + aby * aby;
    if (l2 <= 1e-8f) return (px - bx) * (px - bx) + (py - by) * (py - by);
    float t = (abx * (px - ax) + aby * (py - ay)) / l2;
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    const float qx = ax + t * abx, qy = ay + t * aby;
    return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

__global__ void k_mesh_coverage(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F, int S, float* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    const float x0 = verts[i0 * 3], y0 = verts[i0 * 3 + 1], z0 = verts[i0 * 3 + 2];
    const float x1 = verts[i1 * 3], y1 = verts[i1 * 3 + 1], z1 = verts[i1 * 3 + 2];
    const float x2 = verts[i2 * 3], y2 = verts[i2 * 3 + 1], z2 = verts[i2 * 3 + 2];
    const float area = mesh_edge(x2, y2, x0, y0, x1, y1);
    if (area <= 1e-8f && area >= -1e-8f) return;
    const float den = area + 1e-8f;
    const float blur = 1e-6f / (float)(2 * S);
    const float grow = sqrtf(blur);
    const float xmin = fminf(x0, fminf(x1, x2)) - grow, xmax = fmaxf(x0, fmaxf(x1, x2)) + grow;
    const float ymin = fminf(y0, fminf(y1, y2)) - grow, ymax = fmaxf(y0, fmaxf(y1, y2)) + grow;
    // pixel column c has centre 1 - (2c+1)/S  =>  c = ((1 - x) S - 1) / 2; one extra pixel of margin on each side
    int c0 = (int)floorf(((1.0f - xmax) * (float)S - 1.0f) * 0.5f) - 1, c1 = (int)ceilf(((1.0f - xmin) * (float)S - 1.0f) * 0.5f) + 1;
    int r0 = (int)floorf(((1.0f - ymax) * (float)S - 1.0f) * 0.5f) - 1, r1 = (int)ceilf(((1.0f - ymin) * (float)S - 1.0f) * 0.5f) + 1;
    c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0; c1 = c1 > S - 1 ? S - 1 : c1; r1 = r1 > S - 1 ? S - 1 : r1;
    for (int r = r0; r <= r1; ++r) {
        const float py = mesh_pix_to_ndc(S - 1 - r, S);
        for (int c = c0; c <= c1; ++c) {
            const float px = mesh_pix_to_ndc(S - 1 - c, S);
            const float w0 = mesh_edge(px, py, x1, y1, x2, y2) / den;
            const float w1 = mesh_edge(px, py, x2, y2, x0, y0) / den;
            const float w2 = mesh_edge(px, py, x0, y0, x1, y1) / den;
            const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
            float dn = t0 + t1 + t2;
            dn = dn > 1e-8f ? dn : 1e-8f;
            const float b0 = t0 / dn, b1 = t1 / dn, b2 = t2 / dn;
            const float pz = b0 * z0 + b1 * z1 + b2 * z2;
            if (pz < 0.0f) continue;
            bool ok = b0 > 0.0f && b1 > 0.0f && b2 > 0.0f;
            if (!ok) {
                float d = mesh_seg_d2(px, py, x0, y0, x1, y1);
                const float d1 = mesh_seg_d2(px, py, x0, y0, x2, y2);
                const float d2 = mesh_seg_d2(px, py, x1, y1, x2, y2);
                d = d1 < d ? d1 : d;
                d = d2 < d ? d2 : d;
                ok = d < blur;
            }
            if (ok) out[r * S + c] = 1.0f;
        }
    }
}

extern "C" int gd_mesh_coverage(const float* verts, const int32_t* faces, int V, int F, int S, float* out, void* stream) {
    GD_REQUIRE(verts && out && S > 0 && V >= 0 && F >= 0, GD_EINVAL, "gd_mesh_coverage: bad argument");
    GD_REQUIRE(F == 0 || faces, GD_EINVAL, "gd_mesh_coverage: null faces");
    hipStream_t st = as_stream(stream);
    gd_zero_async(out, (size_t)S * S * sizeof(float), st);
    if (F > 0) k_mesh_coverage<<<(F + 127) / 128, 128, 0, st>>>(verts, faces, F, S, out);
    GD_CHECK_LAUNCH("gd_mesh_coverage");
    return GD_OK;
}
