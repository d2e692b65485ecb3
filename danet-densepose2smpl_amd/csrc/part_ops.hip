// Fused element-wise work of the partial-IUV ("limb") path: everything between the grouped
// predict_partial_iuv convolution and the limb regressor's first convolution, and the three
// partial-IUV losses.  The reference does this on [B,24,3,7,H,W] fp32 tensors with ~40 tensor ops
// (/root/reference/models/danet/danet.py:264-283 part drop + iuvmap_clean per part;
//  /root/reference/models/danet/iuv_estimator.py:206-246 part_iuv_simp + affine_grid + grid_sample of
//  the ground truth + body_uv_losses per part): 264 MB per intermediate at the benchmark size.
// Here one thread owns one (pixel, joint) pair -- 21 prediction channels -- and nothing but the
// prediction itself (NHWC bf16 [B,H,W,24*21], channel = (joint*3 + {u,v,index})*7 + class) and the
// 3-channel ground-truth IUV image is read.
//
//   part_clean        x24[b*24+j, h, w, 0:21] = iuvmap_clean(keep * pred) as bf16 NHWC, channels 21..23 = 0
//                     (24-channel layout = the zero-padded operand the limb regressor's 1x1 conv reads)
//   part_clean_bwd    d pred = onehot * keep * d x24 (U, V channels; the index channels get no gradient)
//   part_loss         sums of: smooth-L1(U), smooth-L1(V) over foreground classes, index cross-entropy,
//                     against the ground truth resampled by the joint's affine theta
//   part_loss_bwd     d pred for the three sums, scaled by three device-side factors
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

constexpr int NJ = 24, NC = 7;     // joints, classes per joint map; a joint's 21 channels sit at stride cpj (21, or 24 when
                                   // the grouped conv's zero-padded output is consumed as it is): pixel stride NJ*cpj
constexpr int NREP = 32;

__device__ inline float norm_coord(int o, int n, int align) {
    return align ? (n > 1 ? -1.0f + 2.0f * (float)o / (float)(n - 1) : 0.0f) : (2.0f * (float)o + 1.0f) / (float)n - 1.0f;
}
__device__ inline float unnorm_coord(float g, int n, int align) {
    return align ? (g + 1.0f) * 0.5f * (float)(n - 1) : ((g + 1.0f) * (float)n - 1.0f) * 0.5f;
}

struct Pred { float u[NC], v[NC], ix[NC]; };

__device__ inline Pred load_pred(const bf16_t* __restrict__ src) {
    Pred p;
#pragma unroll
    for (int c = 0; c < NC; ++c) { p.u[c] = bf2f(src[c]); p.v[c] = bf2f(src[NC + c]); p.ix[c] = bf2f(src[2 * NC + c]); }
    return p;
}

// the same from the zero-padded layout (cpj == 24: 48 bytes per (pixel, joint), 16-byte aligned): three 16-byte loads instead of 21
// two-byte ones
__device__ inline Pred load_pred24(const bf16_t* __restrict__ src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    const uint4 a = q[0], b = q[1], c = q[2];
    const unsigned w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    float f[24];
#pragma unroll
    for (int i = 0; i < 12; ++i) { f[2 * i] = __builtin_bit_cast(float, w[i] << 16); f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
    Pred p;
#pragma unroll
    for (int k = 0; k < NC; ++k) { p.u[k] = f[k]; p.v[k] = f[NC + k]; p.ix[k] = f[2 * NC + k]; }
    return p;
}
__device__ inline void store24(bf16_t* __restrict__ dst, const float* f /* [24] */) {
    unsigned w[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = (unsigned)(unsigned short)f2bf(f[2 * i]) | ((unsigned)(unsigned short)f2bf(f[2 * i + 1]) << 16);
    uint4* q = reinterpret_cast<uint4*>(dst);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]); q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}

__device__ inline int argmax7(const float* x) {          // first maximum, like torch.argmax
    int am = 0; float best = x[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) if (x[c] > best) { best = x[c]; am = c; }
    return am;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void part_clean_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ keep,
                                                         int B, int HW, int cpj, bf16_t* __restrict__ x24)
{
    const long total = (long)B * HW * NJ;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % NJ);
    const long pix = i / NJ;
    const int b = (int)(pix / HW), hw = (int)(pix - (long)b * HW);
    Pred p = load_pred(pred + (pix * NJ + j) * cpj);
    float k[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) k[c] = keep ? keep[((size_t)b * NJ + j) * NC + c] : 1.f;
    float sx[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) sx[c] = p.ix[c] * k[c];
    const int am = argmax7(sx);
    float o[24];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        o[c] = c == am ? p.u[c] * k[c] : 0.f;
        o[NC + c] = c == am ? p.v[c] * k[c] : 0.f;
        o[2 * NC + c] = c == am ? 1.f : 0.f;
    }
    o[21] = o[22] = o[23] = 0.f;
    uint4* dst = reinterpret_cast<uint4*>(x24 + (((size_t)b * NJ + j) * HW + hw) * 24);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint4 r;
        r.x = f2bf_pk(o[q * 8 + 0], o[q * 8 + 1]); r.y = f2bf_pk(o[q * 8 + 2], o[q * 8 + 3]);
        r.z = f2bf_pk(o[q * 8 + 4], o[q * 8 + 5]); r.w = f2bf_pk(o[q * 8 + 6], o[q * 8 + 7]);
        dst[q] = r;
    }
}

__global__ __launch_bounds__(256) void part_clean_bwd_kernel(const bf16_t* __restrict__ g24, const bf16_t* __restrict__ pred,
                                                             const float* __restrict__ keep, int B, int HW, int cpj, bf16_t* __restrict__ gpred)
{
    const long total = (long)B * HW * NJ;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % NJ);
    const long pix = i / NJ;
    const int b = (int)(pix / HW), hw = (int)(pix - (long)b * HW);
    const bf16_t* src = pred + (pix * NJ + j) * cpj;
    float k[NC], sx[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { k[c] = keep ? keep[((size_t)b * NJ + j) * NC + c] : 1.f; sx[c] = bf2f(src[2 * NC + c]) * k[c]; }
    const int am = argmax7(sx);
    const bf16_t* g = g24 + (((size_t)b * NJ + j) * HW + hw) * 24;
    const float gu = bf2f(g[am]) * k[am], gv = bf2f(g[NC + am]) * k[am];
    bf16_t* dst = gpred + (pix * NJ + j) * cpj;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dst[c] = c == am ? f2bf(gu) : (bf16_t)0;
        dst[NC + c] = c == am ? f2bf(gv) : (bf16_t)0;
        dst[2 * NC + c] = 0;
    }
    for (int c = 3 * NC; c < cpj; ++c) dst[c] = 0;
}

// ---------------------------------------------------------------------------------------------
// Ground truth of joint j at output pixel (oh, ow): bilinear resampling (zero padding) of the 7-class
// simplified maps built from the IUV image (class 0 = none of the joint's 6 DensePose parts).
struct Gt { float I[NC], U[NC], V[NC]; };

__device__ inline Gt part_gt(const float* __restrict__ img /* [3][H][W] of sample b */, const float* __restrict__ th /* [2][3] */,
                             const int* __restrict__ sel /* [6] */, int H, int W, int oh, int ow, int align)
{
    Gt g;
#pragma unroll
    for (int c = 0; c < NC; ++c) { g.I[c] = 0.f; g.U[c] = 0.f; g.V[c] = 0.f; }
    const float xn = norm_coord(ow, W, align), yn = norm_coord(oh, H, align);
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    const float ix = unnorm_coord(gx, W, align), iy = unnorm_coord(gy, H, align);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy;
    int s[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) s[c] = sel[c];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int xx = x0 + (n & 1), yy = y0 + (n >> 1);
        if ((unsigned)xx >= (unsigned)W || (unsigned)yy >= (unsigned)H) continue;
        const float wgt = ((n & 1) ? wx1 : 1.f - wx1) * ((n >> 1) ? wy1 : 1.f - wy1);
        const size_t o = (size_t)yy * W + xx;
        const float i0 = img[o];
        int part = (int)rintf(i0 * 24.f);
        part = part < 0 ? 0 : (part > 24 ? 24 : part);
        const float uval = img[(size_t)H * W + o], vval = img[2 * (size_t)H * W + o];
        bool any = false;
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (s[c] == part) { any = true; g.I[c + 1] += wgt; g.U[c + 1] += wgt * uval; g.V[c + 1] += wgt * vval; }
        if (!any) g.I[0] += wgt;
    }
    return g;
}

__device__ inline float smooth_l1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }

// Round 6: a workgroup owns a tile of 256 consecutive pixels of ONE sample and walks its 256 x 24 (pixel, joint) items (consecutive lanes =
// consecutive joints of a pixel: the prediction is read in order); the sample's ground-truth image -- 3 x H x W floats, 48 KB at 64 x 64 --
// is copied to LDS first, so the four bilinear taps of every item (12 gathers at addresses that differ from joint to joint) are LDS reads
// instead of scattered global loads (IMG_LDS; larger images: the global loads as before).  The padded layout's 24 channels of an item
// travel as three 16-byte accesses (load_pred24 / store24) instead of 21 - 24 two-byte ones.  One atomic triple per tile instead of one
// per 256 items.  fwd 136 -> see DESIGN 3.2b; the arithmetic per item is unchanged.
constexpr int PL_TILE = 256;
constexpr int PL_LDS_FLOATS = 3 * 4096;          // largest image kept in LDS: 64 x 64

template <bool IMG_LDS>
__global__ __launch_bounds__(256) void part_loss_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ img,
                                                        const float* __restrict__ theta, const float* __restrict__ wsample,
                                                        const int* __restrict__ sel, int B, int H, int W, int align, int cpj,
                                                        double* __restrict__ sums /* [NREP][3] */)
{
    __shared__ float simg[IMG_LDS ? PL_LDS_FLOATS : 4];
    const int HW = H * W, tiles = (HW + PL_TILE - 1) / PL_TILE;
    const int b = blockIdx.x / tiles, hw0 = (blockIdx.x % tiles) * PL_TILE, t = threadIdx.x;
    const float* gimg = img + (size_t)b * 3 * HW;
    if (IMG_LDS) {
        for (int i = t; i < 3 * HW / 4; i += 256) reinterpret_cast<float4*>(simg)[i] = reinterpret_cast<const float4*>(gimg)[i];
        __syncthreads();
    }
    const float w = wsample ? wsample[b] : 1.f;
    const int npix = HW - hw0 < PL_TILE ? HW - hw0 : PL_TILE;
    float lu = 0.f, lv = 0.f, li = 0.f;
    for (int i = t; i < npix * NJ; i += 256) {
        const int j = i % NJ, hw = hw0 + i / NJ;
        const long pix = (long)b * HW + hw;
        const bf16_t* src = pred + (pix * NJ + j) * cpj;
        const Pred p = cpj == 24 ? load_pred24(src) : load_pred(src);
        const Gt g = part_gt(IMG_LDS ? simg : gimg, theta + ((size_t)b * NJ + j) * 6, sel + j * 6, H, W, hw / W, hw % W, align);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float fg = g.I[c] > 0.f ? w : 0.f;
            lu += smooth_l1(p.u[c] - g.U[c]) * fg;
            lv += smooth_l1(p.v[c] - g.V[c]) * fg;
        }
        const int tgt = argmax7(g.I);
        float mx = p.ix[0];
#pragma unroll
        for (int c = 1; c < NC; ++c) mx = fmaxf(mx, p.ix[c]);
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) se += expf(p.ix[c] - mx);
        li += (logf(se) + mx - p.ix[tgt]) * w;
    }
    __shared__ float red[3][4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lu += __shfl_xor(lu, o); lv += __shfl_xor(lv, o); li += __shfl_xor(li, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = lu; red[1][wave] = lv; red[2][wave] = li; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        // (double accumulators: exact, order-independent adds -- conv_common.h)
        __hip_atomic_fetch_add((__attribute__((address_space(1))) double*)(sums + (blockIdx.x % NREP) * 3 + threadIdx.x), (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// CLEAN: the gradient that arrives through part_clean (g24 [B*24][HW][24], keep [B][24][7] or NULL) is added in the same pass -- the two
// consumers of the prediction (the three losses, the regressor's cleaned operand) then cost ONE read of the prediction and ONE write of its
// gradient instead of two kernels and autograd's add over the 151 MB tensors (round 6: 93 + 125 + 74 us -> one launch).
template <bool IMG_LDS, bool CLEAN>
__global__ __launch_bounds__(256) void part_loss_bwd_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ img,
                                                            const float* __restrict__ theta, const float* __restrict__ wsample,
                                                            const int* __restrict__ sel, const float* __restrict__ scale /* [3] */,
                                                            int B, int H, int W, int align, int cpj, bf16_t* __restrict__ gpred,
                                                            const bf16_t* __restrict__ g24, const float* __restrict__ keep)
{
    __shared__ float simg[IMG_LDS ? PL_LDS_FLOATS : 4];
    const int HW = H * W, tiles = (HW + PL_TILE - 1) / PL_TILE;
    const int b = blockIdx.x / tiles, hw0 = (blockIdx.x % tiles) * PL_TILE, t = threadIdx.x;
    const float* gimg = img + (size_t)b * 3 * HW;
    if (IMG_LDS) {
        for (int i = t; i < 3 * HW / 4; i += 256) reinterpret_cast<float4*>(simg)[i] = reinterpret_cast<const float4*>(gimg)[i];
        __syncthreads();
    }
    const float w = wsample ? wsample[b] : 1.f;
    const float su = scale[0], sv = scale[1], si = scale[2];
    const int npix = HW - hw0 < PL_TILE ? HW - hw0 : PL_TILE;
    for (int i = t; i < npix * NJ; i += 256) {
        const int j = i % NJ, hw = hw0 + i / NJ;
        const long pix = (long)b * HW + hw;
        const bf16_t* src = pred + (pix * NJ + j) * cpj;
        const Pred p = cpj == 24 ? load_pred24(src) : load_pred(src);
        const Gt g = part_gt(IMG_LDS ? simg : gimg, theta + ((size_t)b * NJ + j) * 6, sel + j * 6, H, W, hw / W, hw % W, align);
        const int tgt = argmax7(g.I);
        float mx = p.ix[0];
#pragma unroll
        for (int c = 1; c < NC; ++c) mx = fmaxf(mx, p.ix[c]);
        float ex[NC], se = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { ex[c] = expf(p.ix[c] - mx); se += ex[c]; }
        const float inv = 1.f / se;
        float out[24];
        out[21] = out[22] = out[23] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float fg = g.I[c] > 0.f ? w : 0.f;
            const float du = fminf(fmaxf(p.u[c] - g.U[c], -1.f), 1.f), dv = fminf(fmaxf(p.v[c] - g.V[c], -1.f), 1.f);
            out[c] = su * fg * du;
            out[NC + c] = sv * fg * dv;
            out[2 * NC + c] = si * w * (ex[c] * inv - (c == tgt ? 1.f : 0.f));
        }
        if (CLEAN) {          // part_clean_bwd_kernel's term: the arg-max class of keep * index receives keep * d x24 on its U and V channels
            float k[NC], sx[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { k[c] = keep ? keep[((size_t)b * NJ + j) * NC + c] : 1.f; sx[c] = p.ix[c] * k[c]; }
            const int am = argmax7(sx);
            const Pred g = load_pred24(g24 + (((size_t)b * NJ + j) * HW + hw) * 24);       // (.u / .v = the gradient's U / V channels)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c == am) { out[c] += bf2f(f2bf(g.u[c] * k[c])); out[NC + c] += bf2f(f2bf(g.v[c] * k[c])); }
        }
        bf16_t* dst = gpred + (pix * NJ + j) * cpj;
        if (cpj == 24) store24(dst, out);
        else {
#pragma unroll
            for (int c = 0; c < 3 * NC; ++c) dst[c] = f2bf(out[c]);
            for (int c = 3 * NC; c < cpj; ++c) dst[c] = 0;
        }
    }
}

}  // namespace

extern "C" int danet_part_clean_forward(const void* pred, const float* keep, int B, int H, int W, int cpj, void* x24, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(pred && x24 && B > 0 && H > 0 && W > 0 && cpj >= 3 * NC, "part_clean_forward: bad arguments");
    const long total = (long)B * H * W * NJ;
    hipLaunchKernelGGL(part_clean_kernel, dim3((unsigned)danet::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)pred, keep, B, H * W, cpj, (bf16_t*)x24);
    DANET_CHECK_LAUNCH("part_clean_kernel");
    return DANET_OK;
}

extern "C" int danet_part_clean_backward(const void* g24, const void* pred, const float* keep, int B, int H, int W, int cpj, void* gpred, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(g24 && pred && gpred && B > 0 && H > 0 && W > 0 && cpj >= 3 * NC, "part_clean_backward: bad arguments");
    const long total = (long)B * H * W * NJ;
    hipLaunchKernelGGL(part_clean_bwd_kernel, dim3((unsigned)danet::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)g24, (const bf16_t*)pred, keep, B, H * W, cpj, (bf16_t*)gpred);
    DANET_CHECK_LAUNCH("part_clean_bwd_kernel");
    return DANET_OK;
}

// sums: [32][3] floats, zeroed by the caller; the loss terms are the column sums.
extern "C" int danet_part_loss_forward(const void* pred, const float* iuv_img, const float* theta, const float* sample_w,
                                       const int* sel, int B, int H, int W, int align, int cpj, double* sums, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(pred && iuv_img && theta && sel && sums && B > 0 && H > 0 && W > 0 && cpj >= 3 * NC, "part_loss_forward: bad arguments");
    const int HW = H * W, tiles = (HW + PL_TILE - 1) / PL_TILE;
    if (3 * HW <= PL_LDS_FLOATS && HW % 4 == 0)
        hipLaunchKernelGGL(part_loss_kernel<true>, dim3((unsigned)(B * tiles)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)pred, iuv_img, theta, sample_w, sel, B, H, W, align, cpj, sums);
    else
        hipLaunchKernelGGL(part_loss_kernel<false>, dim3((unsigned)(B * tiles)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)pred, iuv_img, theta, sample_w, sel, B, H, W, align, cpj, sums);
    DANET_CHECK_LAUNCH("part_loss_kernel");
    return DANET_OK;
}

static int part_bwd_launch(const void* pred, const float* iuv_img, const float* theta, const float* sample_w, const int* sel, const float* scale,
                           int B, int H, int W, int align, int cpj, void* gpred, const void* g24, const float* keep, void* stream)
{
    const int HW = H * W, tiles = (HW + PL_TILE - 1) / PL_TILE;
    const bool lds = 3 * HW <= PL_LDS_FLOATS && HW % 4 == 0;
#define PART_BWD(L, C) hipLaunchKernelGGL((part_loss_bwd_kernel<L, C>), dim3((unsigned)(B * tiles)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, \
                                          iuv_img, theta, sample_w, sel, scale, B, H, W, align, cpj, (bf16_t*)gpred, (const bf16_t*)g24, keep)
    if (g24) { if (lds) PART_BWD(true, true); else PART_BWD(false, true); }
    else { if (lds) PART_BWD(true, false); else PART_BWD(false, false); }
#undef PART_BWD
    DANET_CHECK_LAUNCH("part_loss_bwd_kernel");
    return DANET_OK;
}

extern "C" int danet_part_loss_backward(const void* pred, const float* iuv_img, const float* theta, const float* sample_w,
                                        const int* sel, const float* scale, int B, int H, int W, int align, int cpj, void* gpred, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(pred && iuv_img && theta && sel && scale && gpred && B > 0 && H > 0 && W > 0 && cpj >= 3 * NC, "part_loss_backward: bad arguments");
    return part_bwd_launch(pred, iuv_img, theta, sample_w, sel, scale, B, H, W, align, cpj, gpred, nullptr, nullptr, stream);
}

// d pred of BOTH consumers of the prediction in one pass: danet_part_loss_backward + danet_part_clean_backward (g24 [B*24,H,W,24] bf16,
// keep [B,24,7] or NULL); the padded layout only (cpj == 24).
extern "C" int danet_part_backward_fused(const void* pred, const float* iuv_img, const float* theta, const float* sample_w,
                                         const int* sel, const float* scale, const void* g24, const float* keep,
                                         int B, int H, int W, int align, int cpj, void* gpred, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(pred && iuv_img && theta && sel && scale && gpred && g24 && B > 0 && H > 0 && W > 0 && cpj == 24, "part_backward_fused: bad arguments (cpj must be 24)");
    return part_bwd_launch(pred, iuv_img, theta, sample_w, sel, scale, B, H, W, align, cpj, gpred, g24, keep, stream);
}
