// Is an elementwise bf16 stream (BatchNorm apply: y = relu(x * sc + sh)) faster with 16-byte than with 8-byte accesses per
// lane at the sizes of the HRNet branches (12.6 MB ... 1.6 MB tensors, 23.6 MB for the four of a block level)?
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/stream_width.hip -o tools/experiments/stream_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ inline unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ inline float lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ inline unsigned op(unsigned u, float sc, float sh) { return pk(fmaxf(fmaf(lo(u), sc, sh), 0.f), fmaxf(fmaf(hi(u), sc, sh), 0.f)); }

template <int W, int UNR>       // W = 2 (8 bytes) or 4 (16 bytes) dwords per lane and access
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ x, unsigned* __restrict__ y, long ndw, float sc, float sh)
{
    const long stride = (long)gridDim.x * 256 * W;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * W; i < ndw; i += stride * UNR) {
        unsigned v[UNR][W];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * stride;
            if (j < ndw) {
                if (W == 2) { const i32x2 q = *reinterpret_cast<const i32x2*>(x + j); v[u][0] = q.x; v[u][1] = q.y; }
                else { const i32x4 q = *reinterpret_cast<const i32x4*>(x + j); v[u][0] = q.x; v[u][1] = q.y; v[u][2 % W] = q.z; v[u][3 % W] = q.w; }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long j = i + u * stride;
            if (j < ndw) {
                if (W == 2) *reinterpret_cast<i32x2*>(y + j) = i32x2{(int)op(v[u][0], sc, sh), (int)op(v[u][1], sc, sh)};
                else *reinterpret_cast<i32x4*>(y + j) = i32x4{(int)op(v[u][0], sc, sh), (int)op(v[u][1], sc, sh), (int)op(v[u][2 % W], sc, sh), (int)op(v[u][3 % W], sc, sh)};
            }
        }
    }
}

template <int W, int UNR>
void run(const unsigned* x, unsigned* y, long bytes, int grid, const char* what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long ndw = bytes / 4;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<W, UNR>), dim3(grid), dim3(256), 0, 0, x, y, ndw, 1.5f, -0.1f);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<W, UNR>), dim3(grid), dim3(256), 0, 0, x, y, ndw, 1.5f, -0.1f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-10s %5.1f MB  %d B/lane unroll %d grid %4d: %6.2f us  %.2f TB/s (read + write)\n", what, bytes / 1e6, W * 4, UNR, grid, us, 2.0 * bytes / us / 1e6);
}

int main()
{
    unsigned *x, *y;
    CK(hipMalloc(&x, 64 << 20)); CK(hipMalloc(&y, 64 << 20));
    CK(hipMemset(x, 0x3c, 64 << 20));
    for (long mb : {23.6e6, 12.6e6, 3.1e6}) {
        const long bytes = (long)mb / 1024 * 1024;
        for (int grid : {512, 1024, 2048}) {
            run<2, 4>(x, y, bytes, grid, "8-byte");
            run<4, 2>(x, y, bytes, grid, "16-byte");
            run<4, 4>(x, y, bytes, grid, "16-byte");
        }
    }
    return 0;
}
