// Small launch-count killers of the train step (no arithmetic of note: each replaces tens of 2-5 us tensor-op launches).
//
//   pad_multi     up to 16 "zero-pad at the end of each dimension" (or crop: the same index map read the other way) copies of small
//                 fp32 tensors in ONE launch.  The parameters of layers whose widths are no multiple of 8 (12 / 21 / 25 / 15 channels:
//                 the heat-map head's Bottleneck(48, 12), the IUV heads) are padded every step -- F.pad is a fill + a copy per tensor
//                 forward and a copy backward.
//   stn_theta     the STN parameters of the 24 part crops from the soft-argmax centres: box / bone scales, learned ratio and offset,
//                 scale jitter, visibility score of the centre (bilinear sample of the part-membership map of the arg-max index plane)
//                 and the [[s,0,cx],[0,s,cy]] matrices -- /root/reference/models/danet/iuv_estimator.py:262-301 (affine_para) and
//                 :176-186 (the single-point grid_sample of the visibility score); one workgroup per batch item.
//                 No gradient: the reference detaches theta before affine_grid (iuv_estimator.py:197).
#include "common.h"

namespace {

constexpr int PAD_MAX = 16;
struct PadJob { const float* src; float* dst; int sd[4]; int dd[4]; };
struct PadMulti { PadJob j[PAD_MAX]; int start[PAD_MAX + 1]; int n; };

__global__ __launch_bounds__(256) void pad_multi_kernel(const PadMulti P)
{
    const int total = P.start[P.n];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int k = 0;
        while (k + 1 < P.n && i >= P.start[k + 1]) ++k;
        const PadJob& J = P.j[k];
        int r = i - P.start[k];
        const int i3 = r % J.dd[3]; r /= J.dd[3];
        const int i2 = r % J.dd[2]; r /= J.dd[2];
        const int i1 = r % J.dd[1];
        const int i0 = r / J.dd[1];
        float v = 0.f;
        if (i0 < J.sd[0] && i1 < J.sd[1] && i2 < J.sd[2] && i3 < J.sd[3])
            v = J.src[(((long)i0 * J.sd[1] + i1) * J.sd[2] + i2) * J.sd[3] + i3];
        J.dst[i - P.start[k]] = v;
    }
}

constexpr int NJ = 24, NPART = 25;

struct StnThetaP {
    const float* centers;        // [B,24,2] in [-1,1]
    const unsigned char* am;     // [B,H,W] arg-max part index or NULL (no visibility test)
    const float* member;         // [24,25] part membership of the joints
    const float* ratio;          // [24]
    const float* offset;         // [24]
    const float* rnd;            // [2,B,24] uniform [0,1) or NULL
    const long* child;           // [24]
    const long* parent;          // [24]
    float* theta;                // [B,24,2,3]
    int B, H, W, align;
    float jit, vis;
};

__global__ __launch_bounds__(64) void stn_theta_kernel(const StnThetaP P)
{
    __shared__ float cx[NJ], cy[NJ];
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < NJ) {
        cx[j] = P.centers[(b * NJ + j) * 2];
        cy[j] = P.centers[(b * NJ + j) * 2 + 1];
    }
    __syncthreads();
    if (j >= NJ) return;
    float xmin = cx[0], xmax = cx[0], ymin = cy[0], ymax = cy[0];
    for (int k = 1; k < NJ; ++k) {
        xmin = fminf(xmin, cx[k]); xmax = fmaxf(xmax, cx[k]);
        ymin = fminf(ymin, cy[k]); ymax = fmaxf(ymax, cy[k]);
    }
    const float scale_box = 0.5f * fmaxf(xmax - xmin, ymax - ymin);
    const int c = (int)P.child[j], p = (int)P.parent[j];
    const float dc = sqrtf((cx[c] - cx[j]) * (cx[c] - cx[j]) + (cy[c] - cy[j]) * (cy[c] - cy[j])) * 0.5f;
    const float dp = sqrtf((cx[p] - cx[j]) * (cx[p] - cx[j]) + (cy[p] - cy[j]) * (cy[p] - cy[j])) * 0.5f;
    const float raw = j == 0 ? scale_box : 2.f * fmaxf(dc, dp);
    const float ra = P.ratio[j], of = P.offset[j];
    float s = raw * fmaxf(ra, 0.f) + fmaxf(of, 0.f);
    float j1 = 1.f, j2 = 1.f;
    if (P.rnd && P.jit > 0.f) {
        j1 = 1.f + P.jit * (P.rnd[b * NJ + j] - 0.5f);
        j2 = 1.f + P.jit * (P.rnd[(P.B + b) * NJ + j] - 0.5f);
    }
    s *= j1;
    bool hidden = false;
    if (P.am && P.vis > 0.f && j > 0) {
        const int H = P.H, W = P.W;
        float ix, iy;
        if (P.align) { ix = (cx[j] + 1.f) * 0.5f * (W - 1); iy = (cy[j] + 1.f) * 0.5f * (H - 1); }
        else { ix = ((cx[j] + 1.f) * W - 1.f) * 0.5f; iy = ((cy[j] + 1.f) * H - 1.f) * 0.5f; }
        const float x0 = floorf(ix), y0 = floorf(iy);
        float score = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float xx = x0 + dx, yy = y0 + dy;
                const float w = (1.f - fabsf(ix - xx)) * (1.f - fabsf(iy - yy));
                if (xx >= 0.f && xx < (float)W && yy >= 0.f && yy < (float)H) {
                    const int part = P.am[((long)b * H + (int)yy) * W + (int)xx];
                    score += w * P.member[j * NPART + part];
                }
            }
        hidden = score < P.vis;
    }
    if (hidden) s = 0.8f * scale_box;
    s *= j2;
    float* t = P.theta + (long)(b * NJ + j) * 6;
    t[0] = s; t[1] = 0.f; t[2] = cx[j];
    t[3] = 0.f; t[4] = s; t[5] = cy[j];
}

// SMPL joint bookkeeping (/root/reference/models/smpl.py:31-37): joints = joints54[:, JOINT_MAP], smpl_joints = joints54[:, :24], joints_J19 =
// joints[:, -24:][:, J24_TO_J19] -- three index ops forward, their scatters and the accumulation of three gradients backward; here one launch each.
__global__ __launch_bounds__(256) void smpl_joints_fwd_kernel(const float* __restrict__ j54, const long* __restrict__ map49, const long* __restrict__ map19,
                                                              int B, int NJ54, int N49, int N19, float* __restrict__ j49, float* __restrict__ j19, float* __restrict__ j24)
{
    const int per = (N49 + N19 + 24) * 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * per) return;
    const int b = i / per, r = i - b * per, row = r / 3, c = r - 3 * row;
    const float* src = j54 + (long)b * NJ54 * 3;
    if (row < N49) j49[((long)b * N49 + row) * 3 + c] = src[map49[row] * 3 + c];
    else if (row < N49 + N19) { const int k = row - N49; j19[((long)b * N19 + k) * 3 + c] = src[map49[N49 - 24 + map19[k]] * 3 + c]; }
    else { const int k = row - N49 - N19; j24[((long)b * 24 + k) * 3 + c] = src[k * 3 + c]; }
}

// g54[b, row] = (row < 24 ? g24[b, row] : 0) + sum over k with map49[k] == row of g49[b, k] + sum over k with map49[N49 - 24 + map19[k]] == row of g19[b, k]
__global__ __launch_bounds__(256) void smpl_joints_bwd_kernel(const float* __restrict__ g49, const float* __restrict__ g19, const float* __restrict__ g24,
                                                              const long* __restrict__ map49, const long* __restrict__ map19,
                                                              int B, int NJ54, int N49, int N19, float* __restrict__ g54)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * NJ54 * 3) return;
    const int b = i / (NJ54 * 3), r = i - b * NJ54 * 3, row = r / 3, c = r - 3 * row;
    float s = (g24 && row < 24) ? g24[((long)b * 24 + row) * 3 + c] : 0.f;
    if (g49)
        for (int k = 0; k < N49; ++k) if ((int)map49[k] == row) s += g49[((long)b * N49 + k) * 3 + c];
    if (g19)
        for (int k = 0; k < N19; ++k) if ((int)map49[N49 - 24 + map19[k]] == row) s += g19[((long)b * N19 + k) * 3 + c];
    g54[i] = s;
}

}  // namespace

// n (<= 16) jobs; job k copies the fp32 tensor src[k] of shape sdims[4k..4k+3] into dst[k] of shape ddims[4k..4k+3] (both dense, row
// major, shapes padded with leading ones): dst[i] = src[i] where i lies inside the source's shape, 0 elsewhere (a source LARGER than the
// destination is cropped: the backward of a pad).  src / dst: host arrays of device pointers.
extern "C" int danet_pad_multi(const void* const* src, void* const* dst, const int* sdims, const int* ddims, int n, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(src && dst && sdims && ddims && n > 0 && n <= PAD_MAX, "pad_multi: %d jobs (1..%d)", n, PAD_MAX);
    PadMulti P;
    long tot = 0;
    for (int k = 0; k < n; ++k) {
        DANET_CHECK_ARG(src[k] && dst[k], "pad_multi: job %d has a NULL tensor", k);
        P.j[k].src = (const float*)src[k];
        P.j[k].dst = (float*)dst[k];
        long nd = 1;
        for (int q = 0; q < 4; ++q) {
            P.j[k].sd[q] = sdims[4 * k + q];
            P.j[k].dd[q] = ddims[4 * k + q];
            DANET_CHECK_ARG(sdims[4 * k + q] > 0 && ddims[4 * k + q] > 0, "pad_multi: job %d has an empty dimension", k);
            nd *= ddims[4 * k + q];
        }
        P.start[k] = (int)tot;
        tot += nd;
        DANET_CHECK_ARG(tot < (1l << 30), "pad_multi: too many elements");
    }
    P.start[n] = (int)tot;
    P.n = n;
    int grid = danet::cdiv(tot, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(pad_multi_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
    DANET_CHECK_LAUNCH("pad_multi_kernel");
    return DANET_OK;
}

// centers fp32 [B,24,2]; am uint8 [B,H,W] or NULL; member fp32 [24,25]; ratio / offset fp32 [24]; rnd fp32 [2,B,24] or NULL; child /
// parent int64 [24]; output theta fp32 [B,24,2,3].
extern "C" int danet_stn_theta_forward(const float* centers, const unsigned char* am, const float* member, const float* ratio,
                                       const float* offset, const float* rnd, const long* child, const long* parent, int B, int H, int W,
                                       int align, float jitter, float vis_score, float* theta, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(centers && member && ratio && offset && child && parent && theta && B > 0, "stn_theta_forward: bad arguments");
    DANET_CHECK_ARG(!am || (H > 0 && W > 0), "stn_theta_forward: index plane %d x %d", H, W);
    StnThetaP P = {centers, am, member, ratio, offset, rnd, child, parent, theta, B, H, W, align, jitter, vis_score};
    hipLaunchKernelGGL(stn_theta_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, P);
    DANET_CHECK_LAUNCH("stn_theta_kernel");
    return DANET_OK;
}

// j54 fp32 [B,NJ54,3]; map49 int64 [N49] (rows of j54), map19 int64 [N19] (indices into the LAST 24 of the N49); outputs j49 [B,N49,3],
// j19 [B,N19,3], j24 [B,24,3] (= j54[:, :24]).
extern "C" int danet_smpl_joints_forward(const float* j54, const long* map49, const long* map19, int B, int NJ54, int N49, int N19,
                                         float* j49, float* j19, float* j24, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(j54 && map49 && map19 && j49 && j19 && j24 && B > 0 && NJ54 >= 24 && N49 >= 24 && N19 > 0, "smpl_joints_forward: bad arguments");
    const long n = (long)B * (N49 + N19 + 24) * 3;
    hipLaunchKernelGGL(smpl_joints_fwd_kernel, dim3(danet::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, j54, map49, map19, B, NJ54, N49, N19, j49, j19, j24);
    DANET_CHECK_LAUNCH("smpl_joints_fwd_kernel");
    return DANET_OK;
}

// g49 / g19 / g24: gradients of the three outputs (any of them NULL = zero); g54 fp32 [B,NJ54,3] is fully written.
extern "C" int danet_smpl_joints_backward(const float* g49, const float* g19, const float* g24, const long* map49, const long* map19,
                                          int B, int NJ54, int N49, int N19, float* g54, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(map49 && map19 && g54 && B > 0 && NJ54 >= 24 && N49 >= 24 && N19 > 0, "smpl_joints_backward: bad arguments");
    const long n = (long)B * NJ54 * 3;
    hipLaunchKernelGGL(smpl_joints_bwd_kernel, dim3(danet::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g49, g19, g24, map49, map19, B, NJ54, N49, N19, g54);
    DANET_CHECK_LAUNCH("smpl_joints_bwd_kernel");
    return DANET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Loss bookkeeping of the estimator (round 6): the fused loss kernels leave raw DOUBLE sums (rows of replicas: sums[r][n], added in
// index order); the reference then scales each by a constant and / or divides by the number of labelled samples
// (/root/reference/models/danet/iuv_estimator.py:325-339, 233-256).  As tensor ops that was ~6 launches per loss and pass
// (select, mul / div, their backward, select-backward = fill + copy, gradient adds): here ONE launch turns the sums into the n
// finished losses  out[i] = sum_r sums[r][i] * a[i] / (b[i] > 0 ? max(sum(w), 1) * b[i] : 1)   (w = per-sample weights, or nw ones)
// and ONE launch turns the n incoming gradients (null = no gradient) into the coefficients the backward kernels multiply by.
struct LossFin { const double* sums; int rows, n; float a[8], b[8]; const float* w; int nw; float* out; const float* g[8]; };

__global__ void loss_finalize_kernel(LossFin p)
{
    const int i = threadIdx.x;
    if (i >= p.n) return;
    float ws = (float)p.nw;
    if (p.w) { ws = 0.f; for (int k = 0; k < p.nw; ++k) ws += p.w[k]; ws = fmaxf(ws, 1.0f); }
    const float f = p.a[i] / (p.b[i] > 0.f ? ws * p.b[i] : 1.0f);
    if (p.sums) {                                   // forward: the finished losses
        double s = 0.0;
        for (int r = 0; r < p.rows; ++r) s += p.sums[(size_t)r * p.n + i];
        p.out[i] = (float)s * f;
    } else {                                        // backward: d loss_i / d sum_i times the incoming gradient
        p.out[i] = p.g[i] ? p.g[i][0] * f : 0.f;
    }
}

extern "C" int danet_loss_finalize(const void* sums, int rows, int n, const float* a, const float* b, const float* w, int nw,
                                   const void* const* grads, float* out, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(out && n >= 1 && n <= 8 && a && b && nw >= 1 && (sums ? rows >= 1 : grads != nullptr), "loss_finalize: bad arguments");
    LossFin p{};
    p.sums = (const double*)sums; p.rows = rows; p.n = n; p.w = w; p.nw = nw; p.out = out;
    for (int i = 0; i < n; ++i) { p.a[i] = a[i]; p.b[i] = b[i]; p.g[i] = grads ? (const float*)grads[i] : nullptr; }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    DANET_CHECK_LAUNCH("loss_finalize_kernel");
    return DANET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 24 part crops of an image as 24 channel groups of ONE map, and back: x [NB * J][HW][V] 16-byte vectors (an NHWC tensor of NB * J
// small maps) <-> y [NB][HW][J][V] (the NHWC tensor [NB, J * C, H, W] the grouped layer4 of the limb branch reads,
// /root/reference/models/danet/smpl_regressor.py:826 `limb_feat.view(nbs, -1, h, w)`).  As tensor operations the view of a channels-last
// tensor was three copies forward and three backward (NCHW-contiguous reshape, back to channels-last, cast).
__global__ __launch_bounds__(256) void regroup_parts_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int J, int HW, int V, long n, int inverse)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // index in the grouped layout [b][p][j][v]
    if (i >= n) return;
    const int v = (int)(i % V);
    long r = i / V;
    const int j = (int)(r % J); r /= J;
    const int p = (int)(r % HW);
    const long b = r / HW;
    const long k = ((b * J + j) * HW + p) * V + v;                // index in the per-crop layout [b][j][p][v]
    if (inverse) y[k] = x[i]; else y[i] = x[k];
}

extern "C" int danet_regroup_parts(const void* x, void* y, int NB, int J, int HW, int row_bytes, int inverse, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && y && NB > 0 && J > 0 && HW > 0 && row_bytes > 0 && row_bytes % 16 == 0, "regroup_parts: bad arguments (a pixel's channels must be a multiple of 16 bytes)");
    const int V = row_bytes / 16;
    const long n = (long)NB * J * HW * V;
    hipLaunchKernelGGL(regroup_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, J, HW, V, n, inverse);
    DANET_CHECK_LAUNCH("regroup_parts_kernel");
    return DANET_OK;
}

// The input image [B, C, H, W] fp32 (NCHW) -> bf16 NHWC with the channels zero-padded to CP (a multiple of 8, <= 8): the operand of the
// first convolution in one launch (as tensor operations: cast, channels-last copy, fill, pad copy).  No gradient (the image is data).
__global__ __launch_bounds__(256) void pack_image_kernel(const float* __restrict__ x, uint4* __restrict__ y, int C, long HW, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // pixel index b * HW + p
    if (i >= n) return;
    const long b = i / HW, p = i % HW;
    unsigned short h[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float f = c < C ? x[(b * C + c) * HW + p] : 0.f;
        unsigned u = __builtin_bit_cast(unsigned, f);
        // round to nearest even, NaN kept quiet (torch's float -> bfloat16)
        h[c] = (f != f) ? (unsigned short)0x7fc0 : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    uint4 o;
    o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16); o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
    y[i] = o;
}

extern "C" int danet_pack_image(const float* x, void* y, int B, int C, int H, int W, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && y && B > 0 && C >= 1 && C <= 8 && H > 0 && W > 0, "pack_image: bad arguments (1 <= C <= 8)");
    const long n = (long)B * H * W;
    hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint4*)y, C, (long)H * W, n);
    DANET_CHECK_LAUNCH("pack_image_kernel");
    return DANET_OK;
}
