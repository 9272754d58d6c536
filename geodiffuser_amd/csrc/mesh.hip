// R14 (setup, once per edit) — coverage mask of the transformed object surface mesh ("amodal mask").
//
// Replaces pytorch3d.rasterize_meshes + TexturesVertex.sample_textures as used by splatter_mesh
// (GeoDiffuser/utils/warp_utils.py:235-298): the texture is identically 1 and only slot 0 is read (:266), so the
// result is 1 wherever a pixel centre is covered by any face with non-negative depth.  PARITY UNPINNED (pytorch3d is
// absent): pixel centres as in the point rasterizer (A1); a pixel counts as covered when all three edge functions have
// the same sign or are zero (the reference's blur radius of 1e-6/(2S) admits boundary pixels).
#include "common.hpp"

#pragma clang fp contract(off)

__device__ __forceinline__ float mesh_pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

__global__ void k_mesh_coverage(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F, int S, float* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    const float x0 = verts[i0 * 3], y0 = verts[i0 * 3 + 1], z0 = verts[i0 * 3 + 2];
    const float x1 = verts[i1 * 3], y1 = verts[i1 * 3 + 1], z1 = verts[i1 * 3 + 2];
    const float x2 = verts[i2 * 3], y2 = verts[i2 * 3 + 1], z2 = verts[i2 * 3 + 2];
    const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    if (fabsf(area) <= 1e-8f) return;
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    // pixel column c has centre 1 - (2c+1)/S  =>  c = ((1 - x) S - 1) / 2
    int c0 = (int)floorf(((1.0f - xmax) * S - 1.0f) * 0.5f) - 1, c1 = (int)ceilf(((1.0f - xmin) * S - 1.0f) * 0.5f) + 1;
    int r0 = (int)floorf(((1.0f - ymax) * S - 1.0f) * 0.5f) - 1, r1 = (int)ceilf(((1.0f - ymin) * S - 1.0f) * 0.5f) + 1;
    c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0; c1 = c1 > S - 1 ? S - 1 : c1; r1 = r1 > S - 1 ? S - 1 : r1;
    for (int r = r0; r <= r1; ++r) {
        const float py = mesh_pix_to_ndc(S - 1 - r, S);
        for (int c = c0; c <= c1; ++c) {
            const float px = mesh_pix_to_ndc(S - 1 - c, S);
            const float w0 = (x1 - px) * (y2 - py) - (x2 - px) * (y1 - py);
            const float w1 = (x2 - px) * (y0 - py) - (x0 - px) * (y2 - py);
            const float w2 = (x0 - px) * (y1 - py) - (x1 - px) * (y0 - py);
            const bool pos = w0 >= 0.f && w1 >= 0.f && w2 >= 0.f, neg = w0 <= 0.f && w1 <= 0.f && w2 <= 0.f;
            if (!(pos || neg)) continue;
            const float pz = (w0 * z0 + w1 * z1 + w2 * z2) / area;
            if (pz < 0.f) continue;
            out[r * S + c] = 1.0f;
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
