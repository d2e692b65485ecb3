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
// Mapping: one workgroup of 1024 lanes per (image, row band).  Phase 1 projects the DensePose
// vertices once into LDS (coalesced gather through vert_mapping).  Phase 2 gives every lane a
// stride of faces: a lane walks the few pixels of its face's bounding box (the body covers
// ~2.4k pixels with ~6.9k front faces, so most faces touch 0-2 pixel centres) and resolves
// visibility with a 64-bit LDS atomic min on (depth bits << 32 | face id) -- depth > 0, so the
// unsigned order is the float order and ties fall to the lower face id.  Phase 3 writes the three
// colour planes coalesced.  Nothing but the final image touches HBM.
#include "common.h"

namespace {

constexpr float NR_NEAR = 0.1f;
constexpr float NR_FAR = 100.0f;
constexpr int RASTER_THREADS = 1024;
constexpr int MAX_BAND_PIXELS = 4096;
constexpr unsigned long long EMPTY = 0xFFFFFFFFFFFFFFFFull;

__global__ __launch_bounds__(RASTER_THREADS) void iuv_raster_kernel(
    const float* __restrict__ verts, const float* __restrict__ cam, int NV,
    const int* __restrict__ vert_mapping, int NDV,
    const int* __restrict__ faces, const float* __restrict__ tex, int F,
    float focal, float orig, int S, int band_rows,
    float* __restrict__ out, int* __restrict__ face_idx, float* __restrict__ depth)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* sDepth = reinterpret_cast<unsigned long long*>(smem);              // [band pixels]
    float* sNdc = reinterpret_cast<float*>(smem + (size_t)MAX_BAND_PIXELS * 8);            // [NDV][3]

    const int b = blockIdx.x, band = blockIdx.y, t = threadIdx.x;
    const int row0 = band * band_rows;
    const int rows = min(band_rows, S - row0);
    const int npix = rows * S;

    float fx = focal, cx = orig / 2.0f;
    if (orig != 224.0f) {
        const float sc = orig / 224.0f;
        fx = fx * sc;
        cx = cx * sc;
    }
    const float fy = fx, cy = cx;
    const float half = orig / 2.0f;
    const float Sf = (float)S;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float tz = (2.0f * focal) / (orig * s + 1e-9f);

    for (int i = t; i < npix; i += RASTER_THREADS) sDepth[i] = EMPTY;
    for (int i = t; i < NDV; i += RASTER_THREADS) {
        const float* p = verts + ((size_t)b * NV + vert_mapping[i]) * 3;
        const float px = p[0] + tx, py = p[1] + ty, pz = p[2] + tz;
        const float zz = pz + 1e-9f;
        const float x = px / zz, y = py / zz;
        float u = fx * x + cx;
        float v = fy * y + cy;
        v = orig - v;
        u = 2.0f * (u - half) / orig;
        v = 2.0f * (v - half) / orig;
        sNdc[i * 3 + 0] = u; sNdc[i * 3 + 1] = v; sNdc[i * 3 + 2] = pz;
    }
    __syncthreads();

    const float rlo = (float)row0, rhi = (float)(row0 + rows - 1);
    for (int f = t; f < F; f += RASTER_THREADS) {
        const int i0 = faces[f * 3 + 0], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
        const float x0 = sNdc[i0 * 3 + 0], y0 = sNdc[i0 * 3 + 1], z0 = sNdc[i0 * 3 + 2];
        const float x1 = sNdc[i1 * 3 + 0], y1 = sNdc[i1 * 3 + 1], z1 = sNdc[i1 * 3 + 2];
        const float x2 = sNdc[i2 * 3 + 0], y2 = sNdc[i2 * 3 + 1], z2 = sNdc[i2 * 3 + 2];
        const float area2 = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
        if (!(area2 > 0.0f)) continue;
        const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
        const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
        if (!(xmin <= xmax) || !(ymin <= ymax)) continue;
        float cl = floorf((xmin * Sf + Sf - 1.0f) * 0.5f) - 1.0f, ch = ceilf((xmax * Sf + Sf - 1.0f) * 0.5f) + 1.0f;
        float rl = floorf((Sf - 1.0f - ymax * Sf) * 0.5f) - 1.0f, rh = ceilf((Sf - 1.0f - ymin * Sf) * 0.5f) + 1.0f;
        if (cl < 0.0f) cl = 0.0f;
        if (rl < rlo) rl = rlo;
        if (ch > Sf - 1.0f) ch = Sf - 1.0f;
        if (rh > rhi) rh = rhi;
        if (!(cl <= ch) || !(rl <= rh)) continue;
        const int c0 = (int)cl, c1 = (int)ch, r0 = (int)rl, r1 = (int)rh;
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
                atomicMin(&sDepth[(r - row0) * S + cc], key);
            }
        }
    }
    __syncthreads();

    const size_t plane = (size_t)S * S;
    for (int i = t; i < npix; i += RASTER_THREADS) {
        const unsigned long long key = sDepth[i];
        const int f = key == EMPTY ? -1 : (int)(unsigned int)(key & 0xFFFFFFFFull);
        const size_t pix = (size_t)row0 * S + i;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        if (f >= 0) { c0 = tex[f * 3 + 0]; c1 = tex[f * 3 + 1]; c2 = tex[f * 3 + 2]; }
        out[((size_t)b * 3 + 0) * plane + pix] = c0;
        out[((size_t)b * 3 + 1) * plane + pix] = c1;
        out[((size_t)b * 3 + 2) * plane + pix] = c2;
        if (face_idx) face_idx[(size_t)b * plane + pix] = f;
        if (depth) depth[(size_t)b * plane + pix] = f >= 0 ? __uint_as_float((unsigned int)(key >> 32)) : __builtin_inff();
    }
}

}  // namespace

extern "C" int danet_iuv_raster_forward(const float* verts, const float* cam, int B, int NV,
                                        const int32_t* vert_mapping, int NDV,
                                        const int32_t* faces, const float* tex, int F,
                                        float focal, float orig, int S,
                                        float* out, int32_t* face_idx, float* depth, void* stream)
{
    DANET_CHECK_ARG(B > 0 && NV > 0 && NDV > 0 && F > 0 && S > 0, "iuv_raster_forward: bad sizes B=%d NV=%d NDV=%d F=%d S=%d",
                    B, NV, NDV, F, S);
    DANET_CHECK_ARG(verts && cam && vert_mapping && faces && tex && out, "iuv_raster_forward: null pointer");
    DANET_CHECK_ARG(S <= MAX_BAND_PIXELS, "iuv_raster_forward: S=%d too large", S);
    const size_t lds = (size_t)MAX_BAND_PIXELS * 8 + (size_t)NDV * 3 * sizeof(float);
    DANET_CHECK_ARG(lds <= 160 * 1024, "iuv_raster_forward: %d mesh vertices do not fit LDS (%zu B)", NDV, lds);
    const int band_rows = MAX_BAND_PIXELS / S < S ? MAX_BAND_PIXELS / S : S;
    const int nbands = (S + band_rows - 1) / band_rows;
    static thread_local size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(iuv_raster_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "iuv_raster: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
        lds_set = lds;
    }
    hipLaunchKernelGGL(iuv_raster_kernel, dim3(B, nbands), dim3(RASTER_THREADS), lds, (hipStream_t)stream,
                       verts, cam, NV, vert_mapping, NDV, faces, tex, F, focal, orig, S, band_rows, out, face_idx, depth);
    DANET_CHECK_LAUNCH("iuv_raster_kernel");
    return DANET_OK;
}
