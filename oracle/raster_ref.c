/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of IUV_Renderer.verts2uvimg.
 *
 * What it follows.  Set-up code of the reference: /root/reference/utils/renderer.py
 *   :213-226  K = [[f,0,orig/2],[0,f,orig/2],[0,0,1]], all four entries scaled by orig/224
 *             when orig != 224 (so the principal point becomes orig/2*orig/224 -- kept as is)
 *   :236-249  vert_mapping = All_vertices-1, faces = All_Faces-1, per-face constant texture
 *             (FaceIndex/24, mean U, mean V)
 *   :251-254  nr.Renderer(camera_mode='projection', image_size=out, fill_back=False,
 *             anti_aliasing=False, orig_size=orig), ambient light 1, directional 0
 *   :270      vertices = verts[:, vert_mapping]
 *   :289      t = [cam_x, cam_y, 2*f/(orig*s + 1e-9)],  R = I
 * and the decoder that defines the integer channel: utils/iuvmap.py:111
 * (part = round(ch0*24)).  The rasterisation arithmetic itself lives in the third-party
 * `neural_renderer` (daniilidis-group fork, requirements.txt:1 / README.md:22, UNPINNED,
 * CUDA-only, absent here).  Its published algorithm is restated below:
 *   projection: p = v + t; x = px/(pz+1e-9), y = py/(pz+1e-9); u = fx x + cx, v = fy y + cy;
 *               v = orig - v; (u,v) -> 2*((u,v) - orig/2)/orig        (NDC, +y up)
 *   raster:     pixel (row r from the top, col c) has centre xp=(2c+1-S)/S, yp=(S-1-2r)/S;
 *               back-facing faces (signed NDC area <= 0) are skipped; a pixel is covered when
 *               all three barycentric weights are >= 0 (inclusive edges); depth is the
 *               perspective-correct zp = 1/sum(w_i/z_i); accepted when near < zp < far
 *               (0.1 / 100) and zp < current depth (strict: the lowest face index wins a tie);
 *               colour = the face's single texel; background 0; no anti-aliasing.
 * PARITY UNPINNED against neural_renderer itself (cannot run here; the reference has no
 * tests).  Tie-breaking and edge inclusivity are this build's documented rule.  The HIP
 * kernel must reproduce THIS file bit-exactly on the part-id plane; both are compiled with
 * FP contraction off so that every operation is a single IEEE-754 binary32 operation.
 */
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#define NR_NEAR 0.1f
#define NR_FAR 100.0f

/* verts [B,NV,3] (SMPL vertices), cam [B,3] = (s,tx,ty), vert_mapping [NDV] -> 0..NV-1,
 * faces [F,3] -> 0..NDV-1, tex [F,3].  out [B,3,S,S]; face_idx [B,S,S] (may be NULL, -1 = bg);
 * depth [B,S,S] (may be NULL, +inf = bg).  scratch is allocated internally. */
int iuv_raster_forward_ref(const float* verts, const float* cam, int B, int NV,
                           const int32_t* vert_mapping, int NDV,
                           const int32_t* faces, const float* tex, int F,
                           float focal, float orig, int S,
                           float* out, int32_t* face_idx_out, float* depth_out)
{
    float* ndc = (float*)malloc(sizeof(float) * 3 * (size_t)NDV);
    float* zbuf = (float*)malloc(sizeof(float) * (size_t)S * S);
    int32_t* fbuf = (int32_t*)malloc(sizeof(int32_t) * (size_t)S * S);
    if (!ndc || !zbuf || !fbuf) return -1;
    float fx = focal, cx = orig / 2.0f;
    if (orig != 224.0f) {
        const float sc = orig / 224.0f;
        fx = fx * sc;
        cx = cx * sc;
    }
    const float fy = fx, cy = cx;
    const float half = orig / 2.0f;
    const float Sf = (float)S;
    for (int b = 0; b < B; ++b) {
        const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
        const float tz = (2.0f * focal) / (orig * s + 1e-9f);
        for (int i = 0; i < NDV; ++i) {
            const float* p = verts + ((size_t)b * NV + vert_mapping[i]) * 3;
            const float px = p[0] + tx, py = p[1] + ty, pz = p[2] + tz;
            const float zz = pz + 1e-9f;
            const float x = px / zz, y = py / zz;
            float u = fx * x + cx;
            float v = fy * y + cy;
            v = orig - v;
            u = 2.0f * (u - half) / orig;
            v = 2.0f * (v - half) / orig;
            ndc[i * 3 + 0] = u; ndc[i * 3 + 1] = v; ndc[i * 3 + 2] = pz;
        }
        for (int i = 0; i < S * S; ++i) { zbuf[i] = INFINITY; fbuf[i] = -1; }
        for (int f = 0; f < F; ++f) {
            const float* a = ndc + 3 * (size_t)faces[f * 3 + 0];
            const float* bb = ndc + 3 * (size_t)faces[f * 3 + 1];
            const float* c = ndc + 3 * (size_t)faces[f * 3 + 2];
            const float x0 = a[0], y0 = a[1], z0 = a[2];
            const float x1 = bb[0], y1 = bb[1], z1 = bb[2];
            const float x2 = c[0], y2 = c[1], z2 = c[2];
            const float area2 = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
            if (!(area2 > 0.0f)) continue;                      /* back-facing / degenerate / NaN */
            /* conservative pixel bounding box (pure optimisation: coverage is decided below) */
            const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
            const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
            if (!(xmin <= xmax) || !(ymin <= ymax)) continue;   /* NaN guard */
            /* xp = (2c+1-S)/S  =>  c = ((xp*S)+S-1)/2 */
            float cl = floorf((xmin * Sf + Sf - 1.0f) * 0.5f) - 1.0f, ch = ceilf((xmax * Sf + Sf - 1.0f) * 0.5f) + 1.0f;
            /* yp = (S-1-2r)/S  =>  r = (S-1-yp*S)/2 */
            float rl = floorf((Sf - 1.0f - ymax * Sf) * 0.5f) - 1.0f, rh = ceilf((Sf - 1.0f - ymin * Sf) * 0.5f) + 1.0f;
            if (cl < 0.0f) cl = 0.0f;
            if (rl < 0.0f) rl = 0.0f;
            if (ch > Sf - 1.0f) ch = Sf - 1.0f;
            if (rh > Sf - 1.0f) rh = Sf - 1.0f;
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
                    const int pix = r * S + cc;
                    if (zp < zbuf[pix]) { zbuf[pix] = zp; fbuf[pix] = f; }
                }
            }
        }
        for (int pix = 0; pix < S * S; ++pix) {
            const int f = fbuf[pix];
            for (int ch3 = 0; ch3 < 3; ++ch3)
                out[((size_t)b * 3 + ch3) * S * S + pix] = f >= 0 ? tex[f * 3 + ch3] : 0.0f;
            if (face_idx_out) face_idx_out[(size_t)b * S * S + pix] = f;
            if (depth_out) depth_out[(size_t)b * S * S + pix] = zbuf[pix];
        }
    }
    free(ndc); free(zbuf); free(fbuf);
    return 0;
}
