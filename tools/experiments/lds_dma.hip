// Hardware questions behind the loader-wave design of csrc/conv3x3.hip (gfx950):
//  1. what does `buffer_load_dwordx4 ... lds` write for lanes whose offset is out of range (zeros or nothing)?
//  2. are exec-masked lanes skipped?  3. does the scalar offset take part in the range check?
//  4. how fast does ONE loader wave per workgroup stream 38 KB tiles into LDS when every CU does the same?
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/lds_dma.hip -o tools/experiments/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void sem_kernel(const int* __restrict__ x, int* __restrict__ y, int nbytes, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int t = threadIdx.x;
    for (int i = t; i < 256; i += 64) reinterpret_cast<int*>(sm)[i] = -7;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(x), 0, nbytes, 0x00020000);
    int voff = t * 16, soff = 0;
    if (mode == 1 && (t & 3) == 1) voff = 0x7fffffff;
    if (mode == 3) soff = nbytes - 512;                 // lanes >= 32: voff + soff beyond the resource, voff alone inside
    if (mode == 4) { voff = t * 16 + 4096; soff = -4096; }   // voff alone beyond (nbytes = 2048), the sum inside
    if (mode == 2) {
        if ((t & 3) != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)sm, 16, voff, soff, 0, 0);
    } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)sm, 16, voff, soff, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = t; i < 256; i += 64) y[i] = reinterpret_cast<int*>(sm)[i];
}

// One loader wave per workgroup: tiles of NI x 1 KB pieces, DEPTH tiles in flight, per-lane offsets precomputed.
#define STREAM_KERNEL(NI, DEPTH)                                                                                              \
__global__ __launch_bounds__(64) void stream_kernel_##NI##_##DEPTH(const int* __restrict__ x, int nbytes, int ntiles, int* __restrict__ sink) \
{                                                                                                                             \
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];                                                        \
    const int t = threadIdx.x;                                                                                                \
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(x), 0, nbytes, 0x00020000);           \
    int voff[NI];                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) voff[i] = (i * 64 + t) * 16;                                               \
    const int tile_bytes = NI * 1024;                                                                                         \
    for (int k = 0; k < ntiles; ++k) {                                                                                        \
        const int tile = blockIdx.x + k * gridDim.x;                                                                          \
        const int soff = (int)(((long)tile * tile_bytes) % (long)(nbytes - tile_bytes)) & ~15;                                \
        unsigned char* dst = sm + (k % DEPTH) * tile_bytes;                                                                   \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(dst + i * 1024), 16, voff[i], soff, 0, 0);                  \
        if (k >= DEPTH - 1) __builtin_amdgcn_s_waitcnt(WAITVM(NI * (DEPTH - 1)));                                             \
    }                                                                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                          \
    if (sink && t == 0) sink[blockIdx.x] = reinterpret_cast<int*>(sm)[blockIdx.x & 63];                                       \
}                                                                                                                             \
void run_stream_##NI##_##DEPTH(const int* x, int nbytes, int grid, int ntiles, int* sink)                                    \
{                                                                                                                             \
    const size_t lds = (size_t)NI * 1024 * DEPTH;                                                                             \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel_##NI##_##DEPTH), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    hipEvent_t e0, e1;                                                                                                        \
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));                                                                         \
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_kernel_##NI##_##DEPTH, dim3(grid), dim3(64), lds, 0, x, nbytes, ntiles, sink); \
    CK(hipEventRecord(e0));                                                                                                   \
    const int reps = 10;                                                                                                      \
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(stream_kernel_##NI##_##DEPTH, dim3(grid), dim3(64), lds, 0, x, nbytes, ntiles, sink); \
    CK(hipEventRecord(e1));                                                                                                   \
    CK(hipEventSynchronize(e1));                                                                                              \
    float ms = 0.f;                                                                                                           \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                                                     \
    const double bytes = (double)grid * ntiles * NI * 1024;                                                                   \
    printf("stream %d KB tiles, depth %d, grid %d, tiles/wg %d: %.2f us per launch, %.2f TB/s, %.1f GB/s per workgroup\n",   \
           NI, DEPTH, grid, ntiles, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e12, bytes * reps / (ms * 1e-3) / 1e9 / grid); \
}
#define WAITVM(n) ((((n) > 63 ? 63 : (n)) & 15) | ((((n) > 63 ? 63 : (n)) >> 4) << 14) | (7 << 4) | (15 << 8))
STREAM_KERNEL(38, 1)
STREAM_KERNEL(38, 2)
STREAM_KERNEL(19, 2)
STREAM_KERNEL(19, 4)
STREAM_KERNEL(16, 1)

int main()
{
    const int n = 1 << 20;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = i + 1;
    int *x, *y;
    CK(hipMalloc(&x, 512 << 20)); CK(hipMalloc(&y, 4096));
    CK(hipMemset(x, 1, 512 << 20));
    CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int mode = 0; mode <= 4; ++mode) {
        const int nbytes = mode == 3 ? 1024 : (mode == 4 ? 2048 : 1024);
        hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 4096, 0, x, y, nbytes, mode);
        int out[256];
        CK(hipMemcpy(out, y, 1024, hipMemcpyDeviceToHost));
        printf("mode %d:", mode);
        for (int i = 0; i < 256; i += (mode == 3 || mode == 4 ? 16 : 4)) printf(" %d", out[i]);
        printf("\n");
    }
    const int nbytes = 256 << 20;          // beyond L2, inside the Infinity Cache on the second pass
    for (int grid : {256, 512}) {
        run_stream_38_1(x, nbytes, grid, 16, y);
        run_stream_38_2(x, nbytes, grid, 16, y);
        run_stream_19_2(x, nbytes, grid, 32, y);
        run_stream_19_4(x, nbytes, grid, 32, y);
        run_stream_16_1(x, nbytes, grid, 32, y);
    }
    run_stream_38_2(x, nbytes, 768, 16, y);
    run_stream_38_2(x, nbytes, 1024, 16, y);
    return 0;
}
