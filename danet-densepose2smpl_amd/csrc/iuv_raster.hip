// Wavefront-parallel IUV triangle rasteriser for gfx950.
//
// Replaces IUV_Renderer.verts2uvimg (/root/reference/utils/renderer.py:256-278), whose
// arithmetic lives in the third-party CUDA package neural_renderer ('projection' camera, flat
// per-face colour, fill_back=False, no anti-aliasing).  The rule implemented here is the one
// documented in DESIGN.md ("raster rule"): NDC projection exactly as renderer.py:213-226,289 set
// it up; a pixel centre is covered when all three edge functions are >= 0; perspective-correct
// depth; near/far = 0.1/100; the nearest face wins and the lowest face index wins a depth tie.
//
// This translation unit is compiled with -ffp-contract=off: every float operation below is one
// IEEE-754 binary32 operation, in a fixed order, so the integer part-id plane is reproducible
// bit-for-bit (the parity tests compare it with an independent CPU restatement).
//
// Mapping (fills the chip; nothing is serial per image):
//   project  one lane per (image, DensePose vertex): gather through vert_mapping, project to NDC,
//            store [B,NDV,3] in the workspace; the same launch resets the 64-bit depth/id buffer.
//   faces    one lane per (image, triangle): set-up, tight pixel bounding box (the body covers
//            ~2.4k pixels with ~6.9k front faces, so most faces touch 0-2 pixel centres), and a
//            64-bit atomic min on (depth bits << 32 | face id) per covered pixel -- depth > 0, so
//            the unsigned order is the float order and ties fall to the lower face id.
//   resolve  one lane per (image, pixel): decode the winning face, write the three colour planes
//            coalesced (plus the optional face-index / depth maps).
#include "common.h"

namespace {

constexpr float NR_NEAR = 0.1f;
constexpr float NR_FAR = 100.0f;
constexpr unsigned long long EMPTY = 0xFFFFFFFFFFFFFFFFull;

__global__ __launch_bounds__(256) void raster_project_kernel(
    const float* __restrict__ verts, const float* __restrict__ cam, int B, int NV,
    const int* __restrict__ vert_mapping, int NDV, float focal, float orig, int npix,
    float* __restrict__ ndc, unsigned long long* __restrict__ zbuf)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npix) zbuf[(size_t)b * npix + i] = EMPTY;
    if (i >= NDV) return;
    float fx = focal, cx = orig / 2.0f;
    if (orig != 224.0f) {
        const float sc = orig / 224.0f;
        fx = fx * sc;
        cx = cx * sc;
    }
    const float fy = fx, cy = cx;
    const float half = orig / 2.0f;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float tz = (2.0f * focal) / (orig * s + 1e-9f);
    const float* p = verts + ((size_t)b * NV + vert_mapping[i]) * 3;
    const float px = p[0] + tx, py = p[1] + ty, pz = p[2] + tz;
    const float zz = pz + 1e-9f;
    const float x = px / zz, y = py / zz;
    float u = fx * x + cx;
    float v = fy * y + cy;
    v = orig - v;
    u = 2.0f * (u - half) / orig;
    v = 2.0f * (v - half) / orig;
    float* o = ndc + ((size_t)b * NDV + i) * 3;
    o[0] = u; o[1] = v; o[2] = pz;
}

__global__ __launch_bounds__(256) void raster_faces_kernel(
    const float* __restrict__ ndc, int NDV, const int* __restrict__ faces, int F, int S,
    unsigned long long* __restrict__ zbuf)
{
    const int b = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float Sf = (float)S;
    const float* nb = ndc + (size_t)b * NDV * 3;
    const int i0 = faces[f * 3 + 0], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    const float x0 = nb[i0 * 3 + 0], y0 = nb[i0 * 3 + 1], z0 = nb[i0 * 3 + 2];
    const float x1 = nb[i1 * 3 + 0], y1 = nb[i1 * 3 + 1], z1 = nb[i1 * 3 + 2];
    const float x2 = nb[i2 * 3 + 0], y2 = nb[i2 * 3 + 1], z2 = nb[i2 * 3 + 2];
    const float area2 = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
    if (!(area2 > 0.0f)) return;
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    if (!(xmin <= xmax) || !(ymin <= ymax)) return;
    // pixel centre xp(c) = (2c+1-S)/S; conservative candidate range (0.01 px slack >> rounding)
    float cl = ceilf((xmin * Sf + Sf - 1.0f) * 0.5f - 0.01f), ch = floorf((xmax * Sf + Sf - 1.0f) * 0.5f + 0.01f);
    float rl = ceilf((Sf - 1.0f - ymax * Sf) * 0.5f - 0.01f), rh = floorf((Sf - 1.0f - ymin * Sf) * 0.5f + 0.01f);
    if (cl < 0.0f) cl = 0.0f;
    if (rl < 0.0f) rl = 0.0f;
    if (ch > Sf - 1.0f) ch = Sf - 1.0f;
    if (rh > Sf - 1.0f) rh = Sf - 1.0f;
    if (!(cl <= ch) || !(rl <= rh)) return;
    const int c0 = (int)cl, c1 = (int)ch, r0 = (int)rl, r1 = (int)rh;
    unsigned long long* zb = zbuf + (size_t)b * S * S;
    for (int r = r0; r <= r1; ++r) {
        const float yp = (Sf - 1.0f - 2.0f * (float)r) / Sf;
        for (int cc = c0; cc <= c1; ++cc) {
            const float xp = (2.0f * (float)cc + 1.0f - Sf) / Sf;
            const float e0 = (x1 - xp) * (y2 - yp) - (y1 - yp) * (x2 - xp);
            const float e1 = (x2 - xp) * (y0 - yp) - (y2 - yp) * (x0 - xp);
            const float e2 = (x0 - xp) * (y1 - yp) - (y0 - yp) * (x1 - xp);
            if (!(e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f)) continue;
            const float w0 = e0 / area2, w1 = e1 / area2, w2 = e2 / area2;
            const float zp = 1.0f / (w0 / z0 + w1 / z1 + w2 / z2);
            if (!(zp > NR_NEAR && zp < NR_FAR)) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned int)f;
            atomicMin(&zb[r * S + cc], key);
        }
    }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(
    const unsigned long long* __restrict__ zbuf, const float* __restrict__ tex, int npix,
    float* __restrict__ out, int* __restrict__ face_idx, float* __restrict__ depth)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const unsigned long long key = zbuf[(size_t)b * npix + i];
    const int f = key == EMPTY ? -1 : (int)(unsigned int)(key & 0xFFFFFFFFull);
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    if (f >= 0) { c0 = tex[f * 3 + 0]; c1 = tex[f * 3 + 1]; c2 = tex[f * 3 + 2]; }
    out[((size_t)b * 3 + 0) * npix + i] = c0;
    out[((size_t)b * 3 + 1) * npix + i] = c1;
    out[((size_t)b * 3 + 2) * npix + i] = c2;
    if (face_idx) face_idx[(size_t)b * npix + i] = f;
    if (depth) depth[(size_t)b * npix + i] = f >= 0 ? __uint_as_float((unsigned int)(key >> 32)) : __builtin_inff();
}

}  // namespace

// workspace: ndc [B,NDV,3] f32 followed by the 64-bit depth/id buffer [B,S,S] (8-byte aligned)
extern "C" size_t danet_iuv_raster_ws_bytes(int B, int NDV, int S) {
    const size_t ndc = ((size_t)B * NDV * 3 * sizeof(float) + 7) / 8 * 8;
    return ndc + (size_t)B * S * S * 8;
}

extern "C" int danet_iuv_raster_forward(const float* verts, const float* cam, int B, int NV,
                                        const int32_t* vert_mapping, int NDV,
                                        const int32_t* faces, const float* tex, int F,
                                        float focal, float orig, int S,
                                        float* out, int32_t* face_idx, float* depth,
                                        void* ws, size_t ws_bytes, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(B > 0 && B < 65536 && NV > 0 && NDV > 0 && F > 0 && S > 0 && S <= 4096,
                    "iuv_raster_forward: bad sizes B=%d NV=%d NDV=%d F=%d S=%d", B, NV, NDV, F, S);
    DANET_CHECK_ARG(verts && cam && vert_mapping && faces && tex && out && ws, "iuv_raster_forward: null pointer");
    DANET_CHECK_ARG(((uintptr_t)ws & 7) == 0, "iuv_raster_forward: workspace must be 8-byte aligned");
    if (ws_bytes < danet_iuv_raster_ws_bytes(B, NDV, S))
        return danet::fail(DANET_ERR_WORKSPACE, "iuv_raster_forward: workspace %zu < %zu bytes", ws_bytes,
                           danet_iuv_raster_ws_bytes(B, NDV, S));
    hipStream_t st = (hipStream_t)stream;
    float* ndc = (float*)ws;
    unsigned long long* zbuf = (unsigned long long*)((char*)ws + ((size_t)B * NDV * 3 * sizeof(float) + 7) / 8 * 8);
    const int npix = S * S;
    const int n1 = NDV > npix ? NDV : npix;
    hipLaunchKernelGGL(raster_project_kernel, dim3(danet::cdiv(n1, 256), B), dim3(256), 0, st, verts, cam, B, NV,
                       vert_mapping, NDV, focal, orig, npix, ndc, zbuf);
    DANET_CHECK_LAUNCH("raster_project_kernel");
    hipLaunchKernelGGL(raster_faces_kernel, dim3(danet::cdiv(F, 256), B), dim3(256), 0, st, ndc, NDV, faces, F, S, zbuf);
    DANET_CHECK_LAUNCH("raster_faces_kernel");
    hipLaunchKernelGGL(raster_resolve_kernel, dim3(danet::cdiv(npix, 256), B), dim3(256), 0, st, zbuf, tex, npix, out,
                       face_idx, depth);
    DANET_CHECK_LAUNCH("raster_resolve_kernel");
    return DANET_OK;
}
