// Global (25-class) IUV glue of the estimator and of DaNet.forward as two kernels per pass instead of ~100 tensor ops:
//
//   iuv_global_fwd   per pixel: argmax of the index head -> exact one-hot (iuvmap_clean, /root/reference/utils/
//                    iuvmap.py:6-38), with the part-drop mask of danet.py:194-203 applied first; U, V masked by it; the
//                    three planes written as the zero-padded 80-channel NHWC bf16 operand of the body regressor's
//                    first conv (danet.py:247) -- and, in training, iuv_img2map of the rendered ground truth
//                    (iuvmap.py:103-147: part = round(24 * ch0), merged 15-way Ann classes) and the four sums of
//                    body_uv_losses (models/danet/iuv_estimator.py:304-341): smooth-L1 of U, V at the ground-truth part's
//                    channel, cross-entropy of the 25-way index and the 15-way Ann logits, each weighted per sample.
//   iuv_global_bwd   the gradient of both paths w.r.t. the four head outputs in one pass.
//   softargmax_*     soft-argmax of the 24 joint heat-maps (utils/keypoints.py:334-394, 2-D branch) and its gradient.
//
// Head tensors are fp32 NHWC with a padded channel stride (the conv epilogue's 32 / 16 floats per pixel); one thread
// owns one pixel, every plane it needs sits in one or two 128-byte lines.
#include "common.h"
#include "conv_common.h"

namespace {

using danet_conv::bf16_t;
using danet_conv::f2bf_pk;

constexpr int NP = 25, NA = 15;
__constant__ int kAnnOfPart[NP] = {0, 1, 1, 2, 3, 4, 5, 6, 7, 6, 7, 8, 9, 8, 9, 10, 11, 10, 11, 12, 13, 12, 13, 14, 14};   // Index2mask, iuvmap.py:108-109

__device__ inline float smooth_l1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ inline float smooth_l1_grad(float d) { return d > 1.f ? 1.f : (d < -1.f ? -1.f : d); }

template <int N>
__device__ inline void load_row(const float* __restrict__ p, float* o) {          // N floats, 16-byte aligned row with >= roundup(N,4) readable floats
#pragma unroll
    for (int q = 0; q < (N + 3) / 4; ++q) {
        const float4 f = reinterpret_cast<const float4*>(p)[q];
        if (4 * q + 0 < N) o[4 * q + 0] = f.x;
        if (4 * q + 1 < N) o[4 * q + 1] = f.y;
        if (4 * q + 2 < N) o[4 * q + 2] = f.z;
        if (4 * q + 3 < N) o[4 * q + 3] = f.w;
    }
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void iuv_global_fwd_kernel(
    const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ ix, const float* __restrict__ an, int ld, int lda,
    const float* __restrict__ gt, const float* __restrict__ w, const float* __restrict__ keep, int B, int HW, int want_loss,
    bf16_t* __restrict__ map, unsigned char* __restrict__ am_raw, unsigned char* __restrict__ am_drop, double* __restrict__ sums)
{
    __shared__ float sred[4][4];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < B * HW;
    const int pc = valid ? p : 0;
    const int b = pc / HW, hw = pc - b * HW;
    float U[NP], V[NP], I[NP];
    load_row<NP>(u + (size_t)pc * ld, U);
    load_row<NP>(v + (size_t)pc * ld, V);
    load_row<NP>(ix + (size_t)pc * ld, I);
    float k[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) k[c] = keep ? keep[b * NP + c] : 1.f;
    int ar = 0, ad = 0;
    float mr = I[0], md = I[0] * k[0];
#pragma unroll
    for (int c = 1; c < NP; ++c) {
        if (I[c] > mr) { mr = I[c]; ar = c; }                       // first maximum, like torch.argmax
        const float d = I[c] * k[c];
        if (d > md) { md = d; ad = c; }
    }
    float l[4] = {0.f, 0.f, 0.f, 0.f};
    if (want_loss) {
        const float g0 = gt[((size_t)b * 3 + 0) * HW + hw], gu = gt[((size_t)b * 3 + 1) * HW + hw], gv = gt[((size_t)b * 3 + 2) * HW + hw];
        int part = (int)rintf(g0 * 24.f);
        part = part < 0 ? 0 : (part > 24 ? 24 : part);
        const float wb = w ? w[b] : 1.f;
        float up = 0.f, vp = 0.f, ip = 0.f, se = 0.f;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            if (c == part) { up = U[c]; vp = V[c]; ip = I[c]; }
            se += __expf(I[c] - mr);
        }
        float A[NA];
        load_row<NA>(an + (size_t)pc * lda, A);
        const int at = kAnnOfPart[part];
        float ma = A[0];
#pragma unroll
        for (int c = 1; c < NA; ++c) ma = fmaxf(ma, A[c]);
        float sa = 0.f, ap = 0.f;
#pragma unroll
        for (int c = 0; c < NA; ++c) { sa += __expf(A[c] - ma); if (c == at) ap = A[c]; }
        if (valid) {
            l[0] = smooth_l1(up - gu) * wb;
            l[1] = smooth_l1(vp - gv) * wb;
            l[2] = (mr + __logf(se) - ip) * wb;
            l[3] = (ma + __logf(sa) - ap) * wb;
        }
    }
    if (valid) {
        // U | V | I planes of the cleaned map + 5 zero channels: 80 bf16 = ten 16-byte stores
        unsigned pk[40];
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            float e[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = 2 * q + h;
                float val = 0.f;
                if (ch < NP) { const int c = ch; val = c == ad ? U[c] * k[c] : 0.f; }
                else if (ch < 2 * NP) { const int c = ch - NP; val = c == ad ? V[c] * k[c] : 0.f; }
                else if (ch < 3 * NP) { const int c = ch - 2 * NP; val = c == ad ? 1.f : 0.f; }
                e[h] = val;
            }
            pk[q] = f2bf_pk(e[0], e[1]);
        }
        uint4* dst = reinterpret_cast<uint4*>(map + (size_t)p * 80);
#pragma unroll
        for (int q = 0; q < 10; ++q) dst[q] = uint4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
        am_raw[p] = (unsigned char)ar;
        am_drop[p] = (unsigned char)ad;
    }
    if (want_loss) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float s = wave_sum(l[i]); if (lane == 0) sred[wv][i] = s; }
        __syncthreads();
        // double accumulators: adding the workgroups' fp32 partial sums is exact, so the totals do not depend on the arrival order (conv_common.h)
        if (threadIdx.x < 4)
            __hip_atomic_fetch_add((__attribute__((address_space(1))) double*)(sums + threadIdx.x),
                                   (double)((sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void iuv_global_bwd_kernel(
    const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ ix, const float* __restrict__ an, int ld, int lda,
    const float* __restrict__ gt, const float* __restrict__ w, const float* __restrict__ keep, const unsigned char* __restrict__ am_drop,
    const bf16_t* __restrict__ dmap, const float* __restrict__ coef, int B, int HW, int want_loss,
    float* __restrict__ du, float* __restrict__ dv, float* __restrict__ di, float* __restrict__ da)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= B * HW) return;
    const int b = p / HW, hw = p - b * HW;
    float gU[NP], gV[NP], gI[NP], gA[NA];
#pragma unroll
    for (int c = 0; c < NP; ++c) { gU[c] = 0.f; gV[c] = 0.f; gI[c] = 0.f; }
#pragma unroll
    for (int c = 0; c < NA; ++c) gA[c] = 0.f;
    if (want_loss) {
        const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];
        const float g0 = gt[((size_t)b * 3 + 0) * HW + hw], gu = gt[((size_t)b * 3 + 1) * HW + hw], gv = gt[((size_t)b * 3 + 2) * HW + hw];
        int part = (int)rintf(g0 * 24.f);
        part = part < 0 ? 0 : (part > 24 ? 24 : part);
        const float wb = w ? w[b] : 1.f;
        float U[NP], V[NP], I[NP], A[NA];
        load_row<NP>(u + (size_t)p * ld, U);
        load_row<NP>(v + (size_t)p * ld, V);
        load_row<NP>(ix + (size_t)p * ld, I);
        load_row<NA>(an + (size_t)p * lda, A);
        float mi = I[0], ma = A[0];
#pragma unroll
        for (int c = 1; c < NP; ++c) mi = fmaxf(mi, I[c]);
#pragma unroll
        for (int c = 1; c < NA; ++c) ma = fmaxf(ma, A[c]);
        float si = 0.f, sa = 0.f;
#pragma unroll
        for (int c = 0; c < NP; ++c) { I[c] = __expf(I[c] - mi); si += I[c]; }
#pragma unroll
        for (int c = 0; c < NA; ++c) { A[c] = __expf(A[c] - ma); sa += A[c]; }
        const float ri = c2 * wb / si, ra = c3 * wb / sa;
        const int at = kAnnOfPart[part];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            gI[c] = I[c] * ri - (c == part ? c2 * wb : 0.f);
            if (c == part) { gU[c] = c0 * wb * smooth_l1_grad(U[c] - gu); gV[c] = c1 * wb * smooth_l1_grad(V[c] - gv); }
        }
#pragma unroll
        for (int c = 0; c < NA; ++c) gA[c] = A[c] * ra - (c == at ? c3 * wb : 0.f);
    }
    if (dmap) {
        // clean path: only the surviving channel of U and V passes its gradient (the one-hot itself is a constant)
        const int ad = am_drop[p];
        const float kk = keep ? keep[b * NP + ad] : 1.f;
        const unsigned short* row = dmap + (size_t)p * 80;
        const float gu_ = __uint_as_float((unsigned)row[ad] << 16) * kk, gv_ = __uint_as_float((unsigned)row[NP + ad] << 16) * kk;
#pragma unroll
        for (int c = 0; c < NP; ++c) if (c == ad) { gU[c] += gu_; gV[c] += gv_; }
    }
    float4* ou = reinterpret_cast<float4*>(du + (size_t)p * ld);
    float4* ov = reinterpret_cast<float4*>(dv + (size_t)p * ld);
    float4* oi = reinterpret_cast<float4*>(di + (size_t)p * ld);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (4 * q >= ld) break;
        auto pick = [&](const float* g, int c) { return c < NP ? g[c] : 0.f; };
        ou[q] = float4{pick(gU, 4 * q), pick(gU, 4 * q + 1), pick(gU, 4 * q + 2), pick(gU, 4 * q + 3)};
        ov[q] = float4{pick(gV, 4 * q), pick(gV, 4 * q + 1), pick(gV, 4 * q + 2), pick(gV, 4 * q + 3)};
        oi[q] = float4{pick(gI, 4 * q), pick(gI, 4 * q + 1), pick(gI, 4 * q + 2), pick(gI, 4 * q + 3)};
    }
    float4* oa = reinterpret_cast<float4*>(da + (size_t)p * lda);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (4 * q >= lda) break;
        auto pick = [&](int c) { return c < NA ? gA[c] : 0.f; };
        oa[q] = float4{pick(4 * q), pick(4 * q + 1), pick(4 * q + 2), pick(4 * q + 3)};
    }
}

// ---- soft-argmax: one workgroup per (sample, joint) ---------------------------------------------------------------------
__device__ inline float block_reduce(float v, float* sm, bool is_max) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o, 64); v = is_max ? fmaxf(v, t) : v + t; }
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])) : (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void softargmax_fwd_kernel(const float* __restrict__ hm, int ld, int J, int H, int W, float scale,
                                                             float* __restrict__ out /* [B*J][2] */, float* __restrict__ saved /* [B*J][4]: max, sum, ex, ey */)
{
    __shared__ float sm[4];
    const int bj = blockIdx.x, b = bj / J, j = bj - b * J, HW = H * W;
    const float* src = hm + (size_t)b * HW * ld + j;
    float m = -3.0e38f;
    for (int p = threadIdx.x; p < HW; p += 256) m = fmaxf(m, scale * src[(size_t)p * ld]);
    m = block_reduce(m, sm, true);
    float s = 0.f, sx = 0.f, sy = 0.f;
    for (int p = threadIdx.x; p < HW; p += 256) {
        const float e = __expf(scale * src[(size_t)p * ld] - m);
        const int y = p / W, x = p - y * W;
        s += e; sx += e * (float)x; sy += e * (float)y;
    }
    s = block_reduce(s, sm, false);
    sx = block_reduce(sx, sm, false);
    sy = block_reduce(sy, sm, false);
    if (threadIdx.x == 0) {
        const float ex = sx / s, ey = sy / s;
        out[bj * 2 + 0] = ex; out[bj * 2 + 1] = ey;
        saved[bj * 4 + 0] = m; saved[bj * 4 + 1] = s; saved[bj * 4 + 2] = ex; saved[bj * 4 + 3] = ey;
    }
}

// d hm[b,p,j] = scale * heat * ((x - ex) * gx + (y - ey) * gy); one thread per (pixel, joint), dense [B*HW][J] output
__global__ __launch_bounds__(256) void softargmax_bwd_kernel(const float* __restrict__ hm, int ld, int J, int H, int W, float scale,
                                                             const float* __restrict__ saved, const float* __restrict__ gout /* [B*J][2] */,
                                                             long total, float* __restrict__ dhm)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % J);
    const long bp = i / J;
    const int HW = H * W;
    const int b = (int)(bp / HW), p = (int)(bp - (long)b * HW);
    const int y = p / W, x = p - y * W;
    const int bj = b * J + j;
    const float m = saved[bj * 4], s = saved[bj * 4 + 1], ex = saved[bj * 4 + 2], ey = saved[bj * 4 + 3];
    const float heat = __expf(scale * hm[bp * ld + j] - m) / s;
    dhm[i] = scale * heat * (((float)x - ex) * gout[bj * 2] + ((float)y - ey) * gout[bj * 2 + 1]);
}

}  // namespace

// u, v, ix: fp32 [B*HW][ld] (25 valid, ld % 4 == 0, ld >= 28); an: [B*HW][lda] (15 valid, lda >= 16); gt: [B,3,H,W] fp32 NCHW
// (ignored unless want_loss); w: [B] per-sample weights or NULL; keep: [B,25] part-drop mask or NULL.
// Outputs: map bf16 [B*HW][80] (U | V | one-hot | zeros), am_raw / am_drop uint8 [B*HW] (argmax without / with the drop
// mask), sums[4] += (sum smooth-L1 U, sum smooth-L1 V, sum CE index, sum CE ann), each term weighted by w.
extern "C" int danet_iuv_global_forward(const float* u, const float* v, const float* ix, const float* an, int ld, int lda,
                                        const float* gt, const float* w, const float* keep, int B, int H, int W, int want_loss,
                                        void* map, unsigned char* am_raw, unsigned char* am_drop, double* sums, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(u && v && ix && map && am_raw && am_drop && B > 0 && H > 0 && W > 0, "iuv_global_forward: bad arguments");
    DANET_CHECK_ARG(ld % 4 == 0 && ld >= 28 && ld <= 32 && (!want_loss || (an && gt && sums && lda % 4 == 0 && lda >= 16)), "iuv_global_forward: channel strides %d / %d", ld, lda);
    const long n = (long)B * H * W;
    hipLaunchKernelGGL(iuv_global_fwd_kernel, dim3(danet::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, u, v, ix, an, ld, lda, gt, w, keep,
                       B, H * W, want_loss, (danet_conv::bf16_t*)map, am_raw, am_drop, sums);
    DANET_CHECK_LAUNCH("iuv_global_fwd_kernel");
    return DANET_OK;
}

// coef[4] (device): dL/d sums; dmap: gradient of the cleaned map (bf16 [B*HW][80]) or NULL.  du, dv, di: fp32 [B*HW][ld],
// da: [B*HW][lda] -- fully written (padding channels as zeros).
extern "C" int danet_iuv_global_backward(const float* u, const float* v, const float* ix, const float* an, int ld, int lda,
                                         const float* gt, const float* w, const float* keep, const unsigned char* am_drop,
                                         const void* dmap, const float* coef, int B, int H, int W, int want_loss,
                                         float* du, float* dv, float* di, float* da, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(u && v && ix && an && am_drop && du && dv && di && da && B > 0 && H > 0 && W > 0, "iuv_global_backward: bad arguments");
    DANET_CHECK_ARG(ld % 4 == 0 && ld >= 28 && ld <= 32 && lda % 4 == 0 && lda >= 16 && lda <= 16 && (!want_loss || (gt && coef)), "iuv_global_backward: channel strides %d / %d", ld, lda);
    const long n = (long)B * H * W;
    hipLaunchKernelGGL(iuv_global_bwd_kernel, dim3(danet::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, u, v, ix, an, ld, lda, gt, w, keep,
                       am_drop, (const danet_conv::bf16_t*)dmap, coef, B, H * W, want_loss, du, dv, di, da);
    DANET_CHECK_LAUNCH("iuv_global_bwd_kernel");
    return DANET_OK;
}

// hm: fp32 [B*H*W][ld] (J valid); out [B,J,2] = expected (x, y) pixel index of softmax(scale * hm) per joint; saved [B,J,4].
extern "C" int danet_softargmax_forward(const float* hm, int ld, int B, int J, int H, int W, float scale, float* out, float* saved, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(hm && out && saved && B > 0 && J > 0 && J <= ld && H > 0 && W > 0, "softargmax_forward: bad arguments");
    hipLaunchKernelGGL(softargmax_fwd_kernel, dim3(B * J), dim3(256), 0, (hipStream_t)stream, hm, ld, J, H, W, scale, out, saved);
    DANET_CHECK_LAUNCH("softargmax_fwd_kernel");
    return DANET_OK;
}

// dhm: dense fp32 [B*H*W][J]
extern "C" int danet_softargmax_backward(const float* hm, int ld, int B, int J, int H, int W, float scale, const float* saved,
                                         const float* gout, float* dhm, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(hm && saved && gout && dhm && B > 0 && J > 0 && J <= ld && H > 0 && W > 0, "softargmax_backward: bad arguments");
    const long total = (long)B * H * W * J;
    hipLaunchKernelGGL(softargmax_bwd_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, hm, ld, J, H, W, scale, saved, gout, total, dhm);
    DANET_CHECK_LAUNCH("softargmax_bwd_kernel");
    return DANET_OK;
}
