// Joint-centric part decomposition: 24 affine bilinear resamplings of the backbone feature map,
// written straight into the channel-concatenated NHWC tensor the grouped partial-IUV conv reads.
//
// Replaces the loop of 24 x (F.affine_grid + F.grid_sample) + torch.cat at
// /root/reference/models/danet/iuv_estimator.py:193-204 (bilinear, zero padding; the thetas are
// detached there, so only the feature map gets a gradient).
//
//   forward   y[b,oh,ow,p*C+c] = bilinear(x[b,:,:,c] at theta[b,p] . (xn(ow), yn(oh), 1))
//   backward  dx as a GATHER (no atomics): affine_para (iuv_estimator.py:293-296) only builds
//             axis-aligned thetas [[sx,0,cx],[0,sy,cy]], so the sample position is separable and
//             monotone in (oh, ow); each input pixel sums tent(iy-y)*tent(ix-x)*dy over the output
//             window that can reach it.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

__device__ inline float norm_coord(int o, int n, int align) {
    return align ? (n > 1 ? -1.0f + 2.0f * (float)o / (float)(n - 1) : 0.0f) : (2.0f * (float)o + 1.0f) / (float)n - 1.0f;
}
__device__ inline float unnorm_coord(float g, int n, int align) {
    return align ? (g + 1.0f) * 0.5f * (float)(n - 1) : ((g + 1.0f) * (float)n - 1.0f) * 0.5f;
}

struct V8 { float v[8]; };
__device__ inline V8 load8(const bf16_t* p) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    V8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o.v[2 * j] = __uint_as_float(w[j] << 16); o.v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    return o;
}
__device__ inline void store8(bf16_t* p, const V8& a) {
    uint4 r;
    r.x = f2bf_pk(a.v[0], a.v[1]); r.y = f2bf_pk(a.v[2], a.v[3]);
    r.z = f2bf_pk(a.v[4], a.v[5]); r.w = f2bf_pk(a.v[6], a.v[7]);
    *reinterpret_cast<uint4*>(p) = r;
}

// fp32 tensors (BASELINE config C4's arithmetic type): the same eight channels per lane, two 16-byte accesses
__device__ inline V8 load8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    V8 o;
    o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; o.v[4] = b.x; o.v[5] = b.y; o.v[6] = b.z; o.v[7] = b.w;
    return o;
}
__device__ inline void store8(float* p, const V8& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// x [B,H,W,C] (T = bf16 or fp32), theta [B,P,2,3] f32 -> y [B,OH,OW,P*C].  C % 8 == 0.
// One workgroup per output row (b, oh); its OW * P * C/8 items (8 channels each) lie contiguously in y in item order, and a lane
// walks them with a stride of 256 -- (ow, p, cv) advance incrementally (round 6: the per-item 64-bit divisions of the first version
// were ~10x the useful instructions: 151 us for a 302 MB write).  The thetas of image b sit in LDS.
template <typename T>
__global__ __launch_bounds__(256) void stn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ theta,
                                                      int B, int H, int W, int C, int P, int OH, int OW, int align,
                                                      T* __restrict__ y)
{
    extern __shared__ float stn_smem[];
    float* const sTh = stn_smem;                                   // [P][6]
    const int b = (int)blockIdx.x / OH, oh = (int)blockIdx.x - b * OH;
    const int t = threadIdx.x;
    for (int i = t; i < P * 6; i += 256) sTh[i] = theta[(size_t)b * P * 6 + i];
    __syncthreads();
    const int CV = C / 8, PC = P * CV, n = OW * PC;
    const float inv_cv = 1.0f / (float)CV;
    const float yn = norm_coord(oh, OH, align);
    const int dq = 256 / PC, dr = 256 - dq * PC;
    int ow = t / PC, pc = t - ow * PC;
    const T* const xb = x + (size_t)b * H * W * C;
    T* const yrow = y + ((size_t)b * OH + oh) * OW * ((size_t)P * C);
    for (int i = t; i < n; i += 256) {
        const int p = (int)(((float)pc + 0.5f) * inv_cv);           // pc / CV (exact: pc < 2^20)
        const int cv = pc - p * CV;
        const float* th = sTh + p * 6;
        const float xn = norm_coord(ow, OW, align);
        const float gx = th[0] * xn + th[1] * yn + th[2];
        const float gy = th[3] * xn + th[4] * yn + th[5];
        const float ix = unnorm_coord(gx, W, align), iy = unnorm_coord(gy, H, align);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        V8 acc;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = y0 + dy, xx = x0 + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float w = (dy ? wy1 : wy0) * (dx ? wx1 : wx0);
                    const V8 a = load8(xb + ((size_t)yy * W + xx) * C + cv * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc.v[j] += w * a.v[j];
                }
            }
        store8(yrow + (size_t)i * 8, acc);                          // item i = (ow, p, cv) IS the layout order of the row
        pc += dr; ow += dq;
        if (pc >= PC) { pc -= PC; ++ow; }
    }
}

// range of output indices o in [0,n_out) with |a*o + b0 - target| < 1
__device__ inline void reach(float a, float b0, float target, int n_out, int* lo, int* hi) {
    if (fabsf(a) < 1e-12f) {
        if (fabsf(b0 - target) < 1.0f) { *lo = 0; *hi = n_out - 1; } else { *lo = 1; *hi = 0; }
        return;
    }
    float l = (target - 1.0f - b0) / a, h = (target + 1.0f - b0) / a;
    if (l > h) { const float t = l; l = h; h = t; }
    // one-index slack on both sides; the tent weight zeroes anything outside
    const float lf = floorf(l) - 1.0f, hf = ceilf(h) + 1.0f;
    *lo = lf < 0.0f ? 0 : (lf > (float)n_out ? n_out : (int)lf);
    *hi = hf > (float)(n_out - 1) ? n_out - 1 : (hf < -1.0f ? -1 : (int)hf);
}

// dx [B,H,W,C] from dy [B,OH,OW,P*C]; axis-aligned thetas only.
// One workgroup per input row (b, yy), a lane per (xx, 8 channels).  Per part the sample coordinates ix(ow) / iy(oh) -- evaluated with
// exactly the forward's expressions -- are tabulated in LDS once per workgroup, and so is the window of output rows that can reach
// row yy together with their row weights: the lanes' loops hold one table read, the tent weight and one independent 16-byte load per
// candidate (no branch inside: a candidate outside the tent gets weight 0, which adds an exact zero), round 6: the first version
// recomputed coordinates and window bounds (two float divisions per part) in every lane and serialised its loads behind `continue`s
// -- 323 us for a 302 MB read.
template <typename T>
__global__ __launch_bounds__(1024) void stn_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ theta,
                                                       int B, int H, int W, int C, int P, int OH, int OW, int align,
                                                       T* __restrict__ dx)
{
    extern __shared__ float stn_smem[];
    float* const sIx = stn_smem;                                    // [P][OW]
    float* const sWy = sIx + P * OW;                                // [P][OH]: weight of output row oh on input row yy (0: out of reach)
    float* const sXa = sWy + P * OH;                                // [P][2]: ix(ow) = x_at0 + (x_at1 - x_at0) * ow, as (slope, offset)
    int* const sOh = reinterpret_cast<int*>(sXa + 2 * P);           // [P][2]: first / last output row with a non-zero weight
    int* const sOw = sOh + 2 * P;                                   // [P][W]: first | last << 16 output column with a non-zero weight on input column xx
    const int b = (int)blockIdx.x / H, yy = (int)blockIdx.x - b * H;
    const int t = threadIdx.x, nt = blockDim.x;
    const float* const thb = theta + (size_t)b * P * 6;
    for (int i = t; i < P * OW; i += nt) {
        const int p = i / OW, ow = i - p * OW;
        sIx[i] = unnorm_coord(thb[p * 6 + 0] * norm_coord(ow, OW, align) + thb[p * 6 + 2], W, align);
    }
    for (int i = t; i < P * OH; i += nt) {
        const int p = i / OH, oh = i - p * OH;
        const float iy = unnorm_coord(thb[p * 6 + 4] * norm_coord(oh, OH, align) + thb[p * 6 + 5], H, align);
        // bilinear corner weights as the forward computes them (floor-based), so that the gradient matches the forward bit pattern
        const float fy = floorf(iy);
        sWy[i] = (int)fy == yy ? 1.0f - (iy - fy) : ((int)fy + 1 == yy ? iy - fy : 0.0f);
    }
    for (int p = t; p < P; p += nt) {
        const float x_at0 = unnorm_coord(thb[p * 6 + 0] * norm_coord(0, OW, align) + thb[p * 6 + 2], W, align);
        const float x_at1 = unnorm_coord(thb[p * 6 + 0] * norm_coord(OW > 1 ? 1 : 0, OW, align) + thb[p * 6 + 2], W, align);
        const float y_at0 = unnorm_coord(thb[p * 6 + 4] * norm_coord(0, OH, align) + thb[p * 6 + 5], H, align);
        const float y_at1 = unnorm_coord(thb[p * 6 + 4] * norm_coord(OH > 1 ? 1 : 0, OH, align) + thb[p * 6 + 5], H, align);
        sXa[2 * p] = x_at1 - x_at0; sXa[2 * p + 1] = x_at0;
        int oh0, oh1;
        reach(y_at1 - y_at0, y_at0, (float)yy, OH, &oh0, &oh1);
        sOh[2 * p] = oh0; sOh[2 * p + 1] = oh1;
    }
    __syncthreads();
    // exact column windows (the same for every row of image b; recomputed per workgroup: 1 536 short scans): reach() brackets the
    // candidates with slack, the tent test of the forward's floor-based weights trims them -- ix(ow) is monotone, the set is contiguous
    for (int i = t; i < P * W; i += nt) {
        const int p = i / W, xx = i - p * W;
        int lo, hi;
        reach(sXa[2 * p], sXa[2 * p + 1], (float)xx, OW, &lo, &hi);
        const float* const ixp = sIx + p * OW;
        auto hit = [&](int ow) { const int f = (int)floorf(ixp[ow]); return f == xx || f + 1 == xx; };
        while (lo <= hi && !hit(lo)) ++lo;
        while (hi >= lo && !hit(hi)) --hi;
        sOw[i] = lo <= hi ? (lo | (hi << 16)) : (1 | (0 << 16));
    }
    __syncthreads();
    const int CV = C / 8;
    const size_t PCs = (size_t)P * C;
    for (int i = t; i < W * CV; i += nt) {
        const int xx = i / CV, cv = i - xx * CV;
        V8 acc;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
        for (int p = 0; p < P; ++p) {
            const int oh0 = sOh[2 * p], oh1 = sOh[2 * p + 1];      // (workgroup-uniform)
            if (oh0 > oh1) continue;
            const int oww = sOw[p * W + xx];
            const int ow0 = oww & 0xffff, ow1 = oww >> 16;
            const float* const ixp = sIx + p * OW;
            for (int oh = oh0; oh <= oh1; ++oh) {
                const float wy = sWy[p * OH + oh];
                if (wy == 0.0f) continue;                           // (uniform: the same row for every lane)
                const T* const drow = dy + (((size_t)b * OH + oh) * OW) * PCs + (size_t)p * C + cv * 8;
                // four candidates at a time, their loads in flight together (a window is 2 / scale + 1 columns: mostly ONE group);
                // a slot past the window reads the window's last column with weight 0 (adds an exact zero)
                for (int ow = ow0; ow <= ow1; ow += 4) {
                    V8 g[4];
                    float w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int o = ow + k <= ow1 ? ow + k : ow1;
                        const float ix = ixp[o];
                        const float fx = floorf(ix);
                        const float wx = (int)fx == xx ? 1.0f - (ix - fx) : ((int)fx + 1 == xx ? ix - fx : 0.0f);
                        w[k] = ow + k <= ow1 ? wy * wx : 0.0f;
                        g[k] = load8(drow + (size_t)o * PCs);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc.v[j] += w[k] * g[k].v[j];
                }
            }
        }
        store8(dx + (((size_t)b * H + yy) * W + xx) * C + cv * 8, acc);
    }
}

// ---- the round-1 kernels (grid-stride, one item per lane with 64-bit index arithmetic), kept selectable for A-B timing: DANET_STN_V1=1
// x [B,H,W,C] (T = bf16 or fp32), theta [B,P,2,3] f32 -> y [B,OH,OW,P*C].  C % 8 == 0.
template <typename T>
__global__ __launch_bounds__(256) void stn_fwd_kernel_v1(const T* __restrict__ x, const float* __restrict__ theta,
                                                      int B, int H, int W, int C, int P, int OH, int OW, int align,
                                                      T* __restrict__ y)
{
    const int CV = C / 8;
    const long total = (long)B * OH * OW * P * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long r = i / CV;
        const int p = (int)(r % P); r /= P;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const int b = (int)(r / OH);
        const float* th = theta + ((size_t)b * P + p) * 6;
        const float xn = norm_coord(ow, OW, align), yn = norm_coord(oh, OH, align);
        const float gx = th[0] * xn + th[1] * yn + th[2];
        const float gy = th[3] * xn + th[4] * yn + th[5];
        const float ix = unnorm_coord(gx, W, align), iy = unnorm_coord(gy, H, align);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        V8 acc;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = y0 + dy, xx = x0 + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float w = (dy ? wy1 : wy0) * (dx ? wx1 : wx0);
                    const V8 a = load8(x + (((size_t)b * H + yy) * W + xx) * C + cv * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc.v[j] += w * a.v[j];
                }
            }
        store8(y + (((size_t)b * OH + oh) * OW + ow) * ((size_t)P * C) + (size_t)p * C + cv * 8, acc);
    }
}

// dx [B,H,W,C] from dy [B,OH,OW,P*C]; axis-aligned thetas only.
template <typename T>
__global__ __launch_bounds__(256) void stn_bwd_kernel_v1(const T* __restrict__ dy, const float* __restrict__ theta,
                                                      int B, int H, int W, int C, int P, int OH, int OW, int align,
                                                      T* __restrict__ dx)
{
    const int CV = C / 8;
    const long total = (long)B * H * W * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long r = i / CV;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        V8 acc;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
        for (int p = 0; p < P; ++p) {
            const float* th = theta + ((size_t)b * P + p) * 6;
            // ix(ow) and iy(oh) evaluated with exactly the forward's expressions
            const float x_at0 = unnorm_coord(th[0] * norm_coord(0, OW, align) + th[2], W, align);
            const float x_at1 = unnorm_coord(th[0] * norm_coord(OW > 1 ? 1 : 0, OW, align) + th[2], W, align);
            const float y_at0 = unnorm_coord(th[4] * norm_coord(0, OH, align) + th[5], H, align);
            const float y_at1 = unnorm_coord(th[4] * norm_coord(OH > 1 ? 1 : 0, OH, align) + th[5], H, align);
            int ow0, ow1, oh0, oh1;
            reach(x_at1 - x_at0, x_at0, (float)xx, OW, &ow0, &ow1);
            reach(y_at1 - y_at0, y_at0, (float)yy, OH, &oh0, &oh1);
            for (int oh = oh0; oh <= oh1; ++oh) {
                const float iy = unnorm_coord(th[4] * norm_coord(oh, OH, align) + th[5], H, align);
                // bilinear corner weights as the forward computes them (floor-based), so that the
                // gradient matches the forward bit pattern of the weights
                const float fy = floorf(iy);
                float wy;
                if ((int)fy == yy) wy = 1.0f - (iy - fy);
                else if ((int)fy + 1 == yy) wy = iy - fy;
                else continue;
                for (int ow = ow0; ow <= ow1; ++ow) {
                    const float ix = unnorm_coord(th[0] * norm_coord(ow, OW, align) + th[2], W, align);
                    const float fx = floorf(ix);
                    float wx;
                    if ((int)fx == xx) wx = 1.0f - (ix - fx);
                    else if ((int)fx + 1 == xx) wx = ix - fx;
                    else continue;
                    const float w = wy * wx;
                    const V8 g = load8(dy + (((size_t)b * OH + oh) * OW + ow) * ((size_t)P * C) + (size_t)p * C + cv * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc.v[j] += w * g.v[j];
                }
            }
        }
        store8(dx + (((size_t)b * H + yy) * W + xx) * C + cv * 8, acc);
    }
}

}  // namespace

template <typename T>
int stn_forward(const void* x, const float* theta, int B, int H, int W, int C, int P, int OH, int OW, int align_corners, void* y, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && theta && y && B > 0 && H > 0 && W > 0 && P > 0 && OH > 0 && OW > 0 && C > 0 && C % 8 == 0,
                    "stn_gather_forward: bad arguments (C=%d must be a multiple of 8)", C);
    static const bool v1 = getenv("DANET_STN_V1") != nullptr && atoi(getenv("DANET_STN_V1")) != 0;
    if (v1) {
        const long total = (long)B * OH * OW * P * (C / 8);
        long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(stn_fwd_kernel_v1<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, theta, B, H, W, C, P, OH, OW, align_corners, (T*)y);
        DANET_CHECK_LAUNCH("stn_fwd_kernel_v1");
        return DANET_OK;
    }
    DANET_CHECK_ARG((long)OW * P * (C / 8) < (1L << 20) && P * 6 * 4 <= 48 * 1024, "stn_gather_forward: row of %d x %d x %d items is too long", OW, P, C / 8);
    hipLaunchKernelGGL(stn_fwd_kernel<T>, dim3((unsigned)(B * OH)), dim3(256), (size_t)P * 6 * sizeof(float), (hipStream_t)stream, (const T*)x,
                       theta, B, H, W, C, P, OH, OW, align_corners, (T*)y);
    DANET_CHECK_LAUNCH("stn_fwd_kernel");
    return DANET_OK;
}

template <typename T>
int stn_backward(const void* dy, const float* theta, int B, int H, int W, int C, int P, int OH, int OW, int align_corners, void* dx, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(dy && theta && dx && B > 0 && H > 0 && W > 0 && P > 0 && OH > 0 && OW > 0 && C > 0 && C % 8 == 0,
                    "stn_gather_backward: bad arguments");
    static const bool v1 = getenv("DANET_STN_V1") != nullptr && atoi(getenv("DANET_STN_V1")) != 0;
    if (v1) {
        const long total = (long)B * H * W * (C / 8);
        long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(stn_bwd_kernel_v1<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)dy, theta, B, H, W, C, P, OH, OW, align_corners, (T*)dx);
        DANET_CHECK_LAUNCH("stn_bwd_kernel_v1");
        return DANET_OK;
    }
    const size_t lds = ((size_t)P * OW + (size_t)P * OH + 4 * (size_t)P + (size_t)P * W) * sizeof(float);
    DANET_CHECK_ARG(lds <= 60 * 1024 && OW < 32768, "stn_gather_backward: %d parts x %d x %d outputs do not fit the coordinate tables", P, OH, OW);
    int threads = (W * (C / 8) + 63) / 64 * 64;                     // a lane per (xx, 8 channels) of the row, whole waves, <= 1024
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(stn_bwd_kernel<T>, dim3((unsigned)(B * H)), dim3((unsigned)threads), lds, (hipStream_t)stream, (const T*)dy,
                       theta, B, H, W, C, P, OH, OW, align_corners, (T*)dx);
    DANET_CHECK_LAUNCH("stn_bwd_kernel");
    return DANET_OK;
}

extern "C" int danet_stn_gather_forward(const void* x, const float* theta, int B, int H, int W, int C, int P,
                                        int OH, int OW, int align_corners, void* y, void* stream)
{ return stn_forward<danet_conv::bf16_t>(x, theta, B, H, W, C, P, OH, OW, align_corners, y, stream); }
extern "C" int danet_stn_gather_backward(const void* dy, const float* theta, int B, int H, int W, int C, int P,
                                         int OH, int OW, int align_corners, void* dx, void* stream)
{ return stn_backward<danet_conv::bf16_t>(dy, theta, B, H, W, C, P, OH, OW, align_corners, dx, stream); }
// fp32 NHWC tensors (conv.precision('fp32'))
extern "C" int danet_stn_gather_forward_f32(const void* x, const float* theta, int B, int H, int W, int C, int P,
                                            int OH, int OW, int align_corners, void* y, void* stream)
{ return stn_forward<float>(x, theta, B, H, W, C, P, OH, OW, align_corners, y, stream); }
extern "C" int danet_stn_gather_backward_f32(const void* dy, const float* theta, int B, int H, int W, int C, int P,
                                             int OH, int OW, int align_corners, void* dx, void* stream)
{ return stn_backward<float>(dy, theta, B, H, W, C, P, OH, OW, align_corners, dx, stream); }
