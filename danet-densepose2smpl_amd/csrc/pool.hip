// 3x3 / stride-2 / pad-1 max pooling of the two regressor stems (/root/reference/models/module/res_module.py:303,
// nn.MaxPool2d(kernel_size=3, stride=2, padding=1)) on NHWC tensors, bf16 or fp32.
//
// HBM-bound streams.  The forward writes, next to the maxima, the position (0..8, row-major in the window) of each
// maximum -- the FIRST one in scan order, as torch's kernel breaks ties -- as one byte per element; the backward is a gather
// over the <= 4 windows that contain an input pixel (no atomics, deterministic): dx = sum of gy over the windows whose
// recorded position is this pixel.  A lane owns 16 bytes of channels (8 bf16 / 4 fp32).
// Algorithmic bytes (768 crops x 64 ch, 32x32 -> 16x16, bf16): forward 100 MB read + 25 MB + 12.6 MB written; backward 25 + 12.6
// read + 100 MB written.
#include "common.h"
#include "conv_common.h"

namespace {

using danet_conv::bf16_t;

template <typename T> struct Pack;
template <> struct Pack<bf16_t> {
    static constexpr int N = 8;
    __device__ static void load(const bf16_t* p, float* v) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    }
    __device__ static void store(bf16_t* p, const float* v) {      // (values are exact bf16 numbers or sums rounded once)
        uint4 r;
        r.x = danet_conv::f2bf_pk(v[0], v[1]); r.y = danet_conv::f2bf_pk(v[2], v[3]);
        r.z = danet_conv::f2bf_pk(v[4], v[5]); r.w = danet_conv::f2bf_pk(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = r;
    }
};
template <> struct Pack<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float* v) { const float4 r = *reinterpret_cast<const float4*>(p); v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
    __device__ static void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                                          int B, int H, int W, int C, int OH, int OW)
{
    constexpr int N = Pack<T>::N;
    const int CV = C / N;
    const long total = (long)B * OH * OW * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long r = i / CV;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        float best[N];
        unsigned char pos[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { best[j] = -INFINITY; pos[j] = 0; }
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    float v[N];
                    Pack<T>::load(x + (((size_t)b * H + iy) * W + ix) * C + cv * N, v);
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (first || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; pos[j] = (unsigned char)(ky * 3 + kx); }      // (NaN wins, as in torch)
                    first = false;
                }
            }
        const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + cv * N;
        Pack<T>::store(y + o, best);
        if constexpr (N == 8) {
            uint2 q;
            q.x = pos[0] | (pos[1] << 8) | (pos[2] << 16) | ((unsigned)pos[3] << 24);
            q.y = pos[4] | (pos[5] << 8) | (pos[6] << 16) | ((unsigned)pos[7] << 24);
            *reinterpret_cast<uint2*>(idx + o) = q;
        } else {
            *reinterpret_cast<unsigned*>(idx + o) = pos[0] | (pos[1] << 8) | (pos[2] << 16) | ((unsigned)pos[3] << 24);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ gy, const unsigned char* __restrict__ idx, T* __restrict__ dx,
                                                          int B, int H, int W, int C, int OH, int OW)
{
    constexpr int N = Pack<T>::N;
    const int CV = C / N;
    const long total = (long)B * H * W * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long r = i / CV;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        // windows (oy, ox) with 2 oy - 1 <= iy <= 2 oy + 1: oy in {iy / 2, (iy + 1) / 2} (equal for odd iy)
        const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = a ? oy1 : oy0;
            if ((a && oy1 == oy0) || oy >= OH) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ox = c ? ox1 : ox0;
                if ((c && ox1 == ox0) || ox >= OW) continue;
                const unsigned me = (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)));
                const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + cv * N;
                float g[N];
                Pack<T>::load(gy + o, g);
                unsigned char pos[N];
                if constexpr (N == 8) {
                    const uint2 q = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pos[j] = (q.x >> (8 * j)) & 255u; pos[4 + j] = (q.y >> (8 * j)) & 255u; }
                } else {
                    const unsigned q = *reinterpret_cast<const unsigned*>(idx + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pos[j] = (q >> (8 * j)) & 255u;
                }
#pragma unroll
                for (int j = 0; j < N; ++j) if (pos[j] == me) acc[j] += g[j];
            }
        }
        Pack<T>::store(dx + (((size_t)b * H + iy) * W + ix) * C + cv * N, acc);
    }
}

template <typename T>
int pool_forward(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && C % Pack<T>::N == 0, "maxpool3x3s2_forward: bad arguments (C=%d)", C);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)B * OH * OW * (C / Pack<T>::N);
    long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, (unsigned char*)idx, B, H, W, C, OH, OW);
    DANET_CHECK_LAUNCH("maxpool_fwd_kernel");
    return DANET_OK;
}
template <typename T>
int pool_backward(const void* gy, const void* idx, void* dx, int B, int H, int W, int C, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(gy && idx && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % Pack<T>::N == 0, "maxpool3x3s2_backward: bad arguments (C=%d)", C);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)B * H * W * (C / Pack<T>::N);
    long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)gy, (const unsigned char*)idx, (T*)dx, B, H, W, C, OH, OW);
    DANET_CHECK_LAUNCH("maxpool_bwd_kernel");
    return DANET_OK;
}

}  // namespace

// x [B,H,W,C] -> y [B,OH,OW,C], OH = (H - 1) / 2 + 1; idx: B*OH*OW*C bytes (position 0..8 of each maximum in its window, first
// maximum in row-major scan order).  C % 8 == 0 (bf16) / C % 4 == 0 (fp32).
extern "C" int danet_maxpool3x3s2_forward(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream)
{ return pool_forward<bf16_t>(x, y, idx, B, H, W, C, stream); }
extern "C" int danet_maxpool3x3s2_backward(const void* gy, const void* idx, void* dx, int B, int H, int W, int C, void* stream)
{ return pool_backward<bf16_t>(gy, idx, dx, B, H, W, C, stream); }
extern "C" int danet_maxpool3x3s2_forward_f32(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream)
{ return pool_forward<float>(x, y, idx, B, H, W, C, stream); }
extern "C" int danet_maxpool3x3s2_backward_f32(const void* gy, const void* idx, void* dx, int B, int H, int W, int C, void* stream)
{ return pool_backward<float>(gy, idx, dx, B, H, W, C, stream); }
