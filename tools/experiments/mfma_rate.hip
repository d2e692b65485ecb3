// Issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD, by where the accumulators live (arch VGPRs vs AGPRs) and by
// how the MFMAs are ordered (the streamed 3x3 kernel's k-step: 12 independent accumulators, A fragment shared by 4, B by 3).
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_rate.hip -o tools/experiments/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MF_V(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF_A(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, long long* cyc, int iters)
{
    f32x4 acc[4][3];
    bf16x8 a[3], b[4];
    for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const bf16x8*>(in + (threadIdx.x + i * 256) * 4);
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const bf16x8*>(in + (threadIdx.x + 1024 + i * 256) * 4);
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {              // VGPR accumulators, nt-major (the kernel's order)
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) MF_V(acc[m][n], a[n], b[m]);
        } else if (MODE == 1) {       // AGPR accumulators, nt-major
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) MF_A(acc[m][n], a[n], b[m]);
        } else if (MODE == 2) {       // the compiler's builtin (its own register choice), nt-major
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], b[m], acc[m][n], 0, 0, 0);
        } else if (MODE == 3) {       // VGPR, mt-major
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) MF_V(acc[m][n], a[n], b[m]);
        } else {                      // AGPR, mt-major
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) MF_A(acc[m][n], a[n], b[m]);
        }
    }
    const long long t1 = clock64();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 3; ++n) s += acc[m][n];
    *reinterpret_cast<f32x4*>(out + (blockIdx.x * 256 + threadIdx.x) * 4) = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* in, float* out, long long* cyc, int grid)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    long long h[1024];
    CK(hipMemcpy(h, cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < grid; ++i) avg += (double)h[i];
    printf("{\"mfma_rate\": \"%s\", \"cycles_per_mfma\": %.2f}\n", name, avg / grid / iters / 12.0);
}

int main()
{
    float *in, *out; long long* cyc;
    const int grid = 256;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, grid * 256 * 16)); CK(hipMalloc(&cyc, 8 * 1024));
    CK(hipMemset(in, 0x3c, 1 << 20));
    run<0>("vgpr acc, nt-major", in, out, cyc, grid);
    run<1>("agpr acc, nt-major", in, out, cyc, grid);
    run<2>("builtin, nt-major", in, out, cyc, grid);
    run<3>("vgpr acc, mt-major", in, out, cyc, grid);
    run<4>("agpr acc, mt-major", in, out, cyc, grid);
    return 0;
}
