// SMPL-side losses of the regressor (/root/reference/models/danet/smpl_regressor.py:141-218 with the helpers :233-298) as two
// kernels per pass instead of ~240 tensor-op launches: per sample the squared / absolute errors of the staged rotation and
// joint-position regressors, the SMPL parameters, the projected 2-D key-points (weak-perspective camera -> translation,
// pin-hole projection, :182-193), the pelvis-centred 3-D key-points, the vertices and the camera regulariser; then the
// masked means (rows selected by has_smpl / has_kp3d: boolean indexing in the reference, per-sample weights here) and
// the yaml weights (:213-218).  Backward: one kernel from the gradients of the ten loss scalars.
#include "common.h"

namespace {

constexpr int NT = 10;      // 0,1 joint_rotation{0,1}  2,3 joint_position{0,1}  4 keypoints_2d  5 keypoints_3d  6 smpl_pose  7 smpl_betas  8 smpl_verts  9 cam

struct LossP {
    const float* para; const float* target;                 // [B,229]: cam 3 | betas 10 | rotmat 216
    const float* jrot[2]; const float* jpos[2];             // [B,216], [B,72] (NULL = stage absent)
    const float* gt_pts;                                     // [B,72]
    const float* joints; const float* verts; const float* tverts;     // [B,49,3], [B,V,3], [B,V,3] (verts may be NULL: weight 0)
    const float* kps2d; const float* kps3d;                 // [B,49,3], [B,24,4]
    const float* has_smpl; const float* has_kp3d;           // [B]
    int B, V;
    float focal, img, op_w, gt_w;
    float w[NT], cnt[NT];                                    // yaml weight and per-sample element count of every term
};

__device__ inline float block_sum(float v, float* sm) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ inline float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

// per-sample sums ps[b][NT]
__global__ __launch_bounds__(256) void smpl_loss_fwd_kernel(LossP p, float* __restrict__ ps)
{
    __shared__ float sm[4];
    const int b = blockIdx.x, t = threadIdx.x;
    float a[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) a[k] = 0.f;
    const float* pa = p.para + (size_t)b * 229;
    const float* tg = p.target + (size_t)b * 229;
    if (t < 216) {
        const float g = tg[13 + t];
        const float d = pa[13 + t] - g; a[6] = d * d;
        if (p.jrot[0]) { const float e = p.jrot[0][(size_t)b * 216 + t] - g; a[0] = e * e; }
        if (p.jrot[1]) { const float e = p.jrot[1][(size_t)b * 216 + t] - g; a[1] = e * e; }
    }
    if (t < 72) {
        const float g = p.gt_pts[(size_t)b * 72 + t];
        if (p.jpos[0]) a[2] = fabsf(p.jpos[0][(size_t)b * 72 + t] - g);
        if (p.jpos[1]) a[3] = fabsf(p.jpos[1][(size_t)b * 72 + t] - g);
    }
    if (t < 10) { const float d = pa[3 + t] - tg[3 + t]; a[7] = d * d; }
    const float s = pa[0], tx = pa[1], ty = pa[2];
    const float tz = 2.f * p.focal / (p.img * s + 1e-9f);
    if (t < 49) {
        const float* J = p.joints + ((size_t)b * 49 + t) * 3;
        const float* G = p.kps2d + ((size_t)b * 49 + t) * 3;
        const float ak = p.focal / (p.img * 0.5f);
        const float iz = 1.f / (J[2] + tz);
        const float u = ak * (J[0] + tx) * iz, v = ak * (J[1] + ty) * iz;
        const float conf = G[2] * (t < 25 ? p.op_w : p.gt_w);
        a[4] = conf * ((u - G[0]) * (u - G[0]) + (v - G[1]) * (v - G[1]));
    }
    if (t < 24) {
        const float* Jb = p.joints + ((size_t)b * 49 + 25) * 3;
        const float* Gb = p.kps3d + (size_t)b * 24 * 4;
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pp = Jb[t * 3 + c] - 0.5f * (Jb[2 * 3 + c] + Jb[3 * 3 + c]);
            const float gg = Gb[t * 4 + c] - 0.5f * (Gb[2 * 4 + c] + Gb[3 * 4 + c]);
            e += (pp - gg) * (pp - gg);
        }
        a[5] = Gb[t * 4 + 3] * e;
    }
    if (p.verts) {
        const float* Vp = p.verts + (size_t)b * p.V * 3;
        const float* Vt = p.tverts + (size_t)b * p.V * 3;
        float e = 0.f;
        for (int i = t; i < p.V * 3; i += 256) e += fabsf(Vp[i] - Vt[i]);
        a[8] = e;
    }
    if (t == 0) { const float e = __expf(-10.f * s); a[9] = e * e; }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const float v = block_sum(a[k], sm);
        if (t == 0) ps[(size_t)b * NT + k] = v;
    }
}

__device__ inline float term_mask(const LossP& p, int k, int b) {
    return k == 5 ? p.has_kp3d[b] : ((k == 4 || k == 9) ? 1.f : p.has_smpl[b]);
}

// out[k] = w_k * sum_b m_b ps[b,k] / (max(sum_b m_b, 1) * cnt_k); norm[k] = w_k / (max(sum m, 1) * cnt_k) kept for the backward
__global__ __launch_bounds__(64) void smpl_loss_finalize_kernel(LossP p, const float* __restrict__ ps, float* __restrict__ out, float* __restrict__ norm)
{
    const int k = threadIdx.x;
    if (k >= NT) return;
    float s = 0.f, m = 0.f;
    for (int b = 0; b < p.B; ++b) { const float mb = term_mask(p, k, b); s += mb * ps[(size_t)b * NT + k]; m += mb; }
    const float n = p.w[k] / (fmaxf(m, 1.f) * p.cnt[k]);
    out[k] = s * n;
    norm[k] = n;
}

struct LossG {
    float* dpara; float* djrot[2]; float* djpos[2]; float* djoints; float* dverts;      // same shapes as the inputs; dverts may be NULL
};

__global__ __launch_bounds__(256) void smpl_loss_bwd_kernel(LossP p, const float* __restrict__ gout, const float* __restrict__ norm, LossG g)
{
    __shared__ float sm[4];
    __shared__ float gk[NT];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < NT) gk[t] = gout[t] * norm[t] * term_mask(p, t, b);
    __syncthreads();
    const float* pa = p.para + (size_t)b * 229;
    const float* tg = p.target + (size_t)b * 229;
    float* dpa = g.dpara + (size_t)b * 229;
    if (t < 216) {
        const float gt = tg[13 + t];
        dpa[13 + t] = 2.f * (pa[13 + t] - gt) * gk[6];
        if (g.djrot[0]) g.djrot[0][(size_t)b * 216 + t] = 2.f * (p.jrot[0][(size_t)b * 216 + t] - gt) * gk[0];
        if (g.djrot[1]) g.djrot[1][(size_t)b * 216 + t] = 2.f * (p.jrot[1][(size_t)b * 216 + t] - gt) * gk[1];
    }
    if (t < 72) {
        const float gt = p.gt_pts[(size_t)b * 72 + t];
        if (g.djpos[0]) g.djpos[0][(size_t)b * 72 + t] = sgn(p.jpos[0][(size_t)b * 72 + t] - gt) * gk[2];
        if (g.djpos[1]) g.djpos[1][(size_t)b * 72 + t] = sgn(p.jpos[1][(size_t)b * 72 + t] - gt) * gk[3];
    }
    if (t < 10) dpa[3 + t] = 2.f * (pa[3 + t] - tg[3 + t]) * gk[7];
    // projected key-points: gradients to the 49 joints and (block-reduced) to the camera
    const float s = pa[0], tx = pa[1], ty = pa[2];
    const float den = p.img * s + 1e-9f;
    const float tz = 2.f * p.focal / den;
    float dtx = 0.f, dty = 0.f, dtz = 0.f;
    float dj[3] = {0.f, 0.f, 0.f};
    if (t < 49) {
        const float* J = p.joints + ((size_t)b * 49 + t) * 3;
        const float* G = p.kps2d + ((size_t)b * 49 + t) * 3;
        const float ak = p.focal / (p.img * 0.5f);
        const float iz = 1.f / (J[2] + tz);
        const float u = ak * (J[0] + tx) * iz, v = ak * (J[1] + ty) * iz;
        const float conf = G[2] * (t < 25 ? p.op_w : p.gt_w);
        const float gu = 2.f * conf * (u - G[0]) * gk[4], gv = 2.f * conf * (v - G[1]) * gk[4];
        dj[0] = gu * ak * iz; dj[1] = gv * ak * iz;
        dj[2] = -(gu * u + gv * v) * iz;
        dtx = dj[0]; dty = dj[1]; dtz = dj[2];
    }
    // pelvis-centred 3-D key-points (joints 25..48)
    float e3[3] = {0.f, 0.f, 0.f};
    if (t < 24) {
        const float* Jb = p.joints + ((size_t)b * 49 + 25) * 3;
        const float* Gb = p.kps3d + (size_t)b * 24 * 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pp = Jb[t * 3 + c] - 0.5f * (Jb[2 * 3 + c] + Jb[3 * 3 + c]);
            const float gg = Gb[t * 4 + c] - 0.5f * (Gb[2 * 4 + c] + Gb[3 * 4 + c]);
            e3[c] = 2.f * Gb[t * 4 + 3] * (pp - gg) * gk[5];
        }
    }
    float tot3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) tot3[c] = block_sum(e3[c], sm);
    const float stx = block_sum(dtx, sm), sty = block_sum(dty, sm), stz = block_sum(dtz, sm);
    if (t < 49) {
        float* D = g.djoints + ((size_t)b * 49 + t) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = dj[c];
            if (t >= 25) {
                const int j = t - 25;
                // the thread that owns 3-D joint j is thread j (< 24): fetch its e3 through shared memory instead -- recompute
                const float* Jb = p.joints + ((size_t)b * 49 + 25) * 3;
                const float* Gb = p.kps3d + (size_t)b * 24 * 4;
                const float pp = Jb[j * 3 + c] - 0.5f * (Jb[2 * 3 + c] + Jb[3 * 3 + c]);
                const float gg = Gb[j * 4 + c] - 0.5f * (Gb[2 * 4 + c] + Gb[3 * 4 + c]);
                v += 2.f * Gb[j * 4 + 3] * (pp - gg) * gk[5];
                if (j == 2 || j == 3) v -= 0.5f * tot3[c];
            }
            D[c] = v;
        }
    }
    if (t == 0) {
        const float e = __expf(-10.f * s);
        dpa[0] = stz * (-2.f * p.focal * p.img / (den * den)) + (-20.f * e * e) * gk[9];
        dpa[1] = stx;
        dpa[2] = sty;
    }
    if (g.dverts) {
        const float* Vp = p.verts + (size_t)b * p.V * 3;
        const float* Vt = p.tverts + (size_t)b * p.V * 3;
        float* D = g.dverts + (size_t)b * p.V * 3;
        for (int i = t; i < p.V * 3; i += 256) D[i] = sgn(Vp[i] - Vt[i]) * gk[8];
    }
}

}  // namespace

extern "C" size_t danet_smpl_loss_param_bytes(void) { return sizeof(LossP); }
extern "C" size_t danet_smpl_loss_grad_bytes(void) { return sizeof(LossG); }

// params: a host LossP (layout above; fill with ctypes), ps [B,10] scratch, out [10] losses, norm [10] (kept for the backward)
extern "C" int danet_smpl_loss_forward(const void* params, float* ps, float* out, float* norm, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(params && ps && out && norm, "smpl_loss_forward: null pointer");
    const LossP p = *(const LossP*)params;
    DANET_CHECK_ARG(p.para && p.target && p.gt_pts && p.joints && p.kps2d && p.kps3d && p.has_smpl && p.has_kp3d && p.B > 0 && (!p.verts || (p.tverts && p.V > 0)),
                    "smpl_loss_forward: bad arguments");
    hipLaunchKernelGGL(smpl_loss_fwd_kernel, dim3(p.B), dim3(256), 0, (hipStream_t)stream, p, ps);
    hipLaunchKernelGGL(smpl_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, (const float*)ps, out, norm);
    DANET_CHECK_LAUNCH("smpl_loss_fwd_kernel");
    return DANET_OK;
}

// gout [10]: gradients of the ten losses; grads: a host LossG (output pointers, shapes of the inputs)
extern "C" int danet_smpl_loss_backward(const void* params, const float* gout, const float* norm, const void* grads, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(params && gout && norm && grads, "smpl_loss_backward: null pointer");
    const LossP p = *(const LossP*)params;
    const LossG g = *(const LossG*)grads;
    DANET_CHECK_ARG(g.dpara && g.djoints && p.B > 0, "smpl_loss_backward: bad arguments");
    hipLaunchKernelGGL(smpl_loss_bwd_kernel, dim3(p.B), dim3(256), 0, (hipStream_t)stream, p, gout, norm, g);
    DANET_CHECK_LAUNCH("smpl_loss_bwd_kernel");
    return DANET_OK;
}
