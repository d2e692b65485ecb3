// How many workgroups of a given shape are resident per CU?  Every workgroup records its CU (HW_ID + XCC_ID), start and
// end time while spinning ~20 us; the host counts overlapping workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/residency.hip -o tools/experiments/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int THREADS, int WPE>
__global__ __launch_bounds__(THREADS, WPE) void k(long long* out, int spin)
{
    extern __shared__ unsigned char sm[];
    const long long t0 = clock64();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        out[blockIdx.x * 4 + 0] = ((long long)(xcc & 15) << 32) | (hw & 0xff00);      // cu, sh, se bits (wave / simd / pipe dropped)
        out[blockIdx.x * 4 + 1] = t0;
    }
    sm[threadIdx.x] = (unsigned char)threadIdx.x;
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 2] = clock64() + sm[7];
}

template <int THREADS, int WPE>
void run(long long* out, int grid, int lds)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<THREADS, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((k<THREADS, WPE>), dim3(grid), dim3(THREADS), lds, 0, out, 40000);
    std::vector<long long> h(grid * 4);
    CK(hipMemcpy(h.data(), out, sizeof(long long) * 4 * grid, hipMemcpyDeviceToHost));
    std::map<long long, std::vector<std::pair<long long, long long>>> cu;
    for (int b = 0; b < grid; ++b) cu[h[b * 4]].push_back({h[b * 4 + 1], h[b * 4 + 2]});
    int maxc = 0; double avg = 0;
    for (auto& kv : cu) {
        int best = 0;
        for (auto& a : kv.second) { int c = 0; for (auto& b : kv.second) if (b.first <= a.first && a.first < b.second) ++c; best = std::max(best, c); }
        maxc = std::max(maxc, best); avg += best;
    }
    long long tmin = h[1], tmax = h[2];
    for (int b = 0; b < grid; ++b) { tmin = std::min(tmin, h[b * 4 + 1]); tmax = std::max(tmax, h[b * 4 + 2]); }
    printf("threads %d waves/EU %d lds %6d grid %4d: %3zu CUs seen, resident workgroups per CU max %d avg %.2f\n", THREADS, WPE, lds, grid, cu.size(), maxc, avg / cu.size());
}

int main()
{
    long long* out;
    CK(hipMalloc(&out, 1 << 20));
    for (int lds : {81920, 81408, 80896, 79872, 65536, 40960}) {
        run<320, 3>(out, 512, lds);
        run<256, 2>(out, 512, lds);
    }
    run<320, 3>(out, 768, 40960);
    run<384, 3>(out, 512, 79872);
    return 0;
}
