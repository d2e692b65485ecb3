// Does writing M0 between LDS-DMA instructions serialise them?  One wave per workgroup, 32 x 1 KB copies of an L2-warm
// buffer; cycles (s_memtime) for issuing them and for their completion, three ways of addressing the LDS destination.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/lds_dma2.hip -o tools/experiments/lds_dma2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int N = 32;

template <int MODE>
__global__ __launch_bounds__(64) void k(const int* __restrict__ x, int nbytes, long long* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int t = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(x), 0, nbytes, 0x00020000);
    const unsigned long long xa = reinterpret_cast<unsigned long long>(x);
    const i32x4 desc = {__builtin_amdgcn_readfirstlane((int)(unsigned)xa), __builtin_amdgcn_readfirstlane((int)((xa >> 32) & 0xffffu)), nbytes, 0x00020000};
    const int base = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * N * 1024));
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)sm;
    long long t0 = 0, t1 = 0, t2 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        __syncthreads();
        t0 = clock64();
        if (MODE == 0) {                    // builtin, constant destinations: the compiler folds them into the immediate offset where it can
#pragma unroll
            for (int i = 0; i < N; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sm + i * 1024), 16, t * 16, base + i * 1024, 0, 0);
        } else if (MODE == 1) {             // M0 rewritten before every copy
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int so = base + i * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds0 + i * 1024), "v"(t * 16), "s"(desc), "s"(so) : "memory");
            }
        } else if (MODE >= 3) {             // builtin with: 3 = every 7th lane out of range; 4 = every lane out of range through the scalar offset;
                                            // 5 = 96-byte pixel records at a 192-byte stride (6 chunks per pixel); 6 = 5 plus 2 padding lanes per pixel
            int vo = t * 16;
            if (MODE == 3 && t % 7 == 3) vo = 0x7fffffff;
            if (MODE == 5) vo = (t / 6) * 192 + (t % 6) * 16;
            if (MODE == 6) vo = (t % 8) < 6 ? (t / 8) * 192 + (t % 8) * 16 : 0x7fffffff;
#pragma unroll
            for (int i = 0; i < N; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sm + i * 1024), 16, vo, MODE == 4 ? 0x40000000 : base + i * 2048, 0, 0);
        } else {                            // M0 rewritten every fourth copy, the others through the immediate offset (which also moves the source)
#pragma unroll
            for (int i = 0; i < N; i += 4) {
                const int so = base + i * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                             "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                             "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                             "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n\t"
                             "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds" :: "s"(lds0 + i * 1024), "v"(t * 16), "s"(desc), "s"(so) : "memory");
            }
        }
        t1 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t2 = clock64();
    }
    if (t == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
    if (t == 1) out[1000 + blockIdx.x] = reinterpret_cast<int*>(sm)[(blockIdx.x * 7) & 1023];
}

template <int MODE>
void run(const int* x, int nbytes, long long* out, int grid, const char* name)
{
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 40 * 1024, 0, x, nbytes, out);
    long long h[64];
    CK(hipMemcpy(h, out, sizeof(long long) * 2 * (grid < 32 ? grid : 32), hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    const int n = grid < 32 ? grid : 32;
    for (int i = 0; i < n; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    printf("%-44s grid %4d: issue %7.0f cycles (%5.1f per copy), issue + landed %7.0f\n", name, grid, a / n, a / n / N, b / n);
}

int main()
{
    int* x; long long* out;
    const int nbytes = 64 << 20;
    CK(hipMalloc(&x, nbytes)); CK(hipMalloc(&out, 65536));
    CK(hipMemset(x, 1, nbytes));
    for (int grid : {8, 256, 512}) {
        run<0>(x, nbytes, out, grid, "builtin (compiler-chosen M0 / immediates)");
        run<1>(x, nbytes, out, grid, "asm, M0 written before every copy");
        run<2>(x, nbytes, out, grid, "asm, M0 written every 4th copy");
        run<3>(x, nbytes, out, grid, "builtin, every 7th lane out of range");
        run<4>(x, nbytes, out, grid, "builtin, all lanes out of range (soffset)");
        run<5>(x, nbytes, out, grid, "builtin, 96 B records at 192 B stride");
        run<6>(x, nbytes, out, grid, "builtin, 96 B records + 2 OOB lanes / pixel");
    }
    return 0;
}
