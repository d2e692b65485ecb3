// Do LDS-DMA loads (buffer_load ... lds) and register loads (buffer_load ...) of ONE wave return in issue order with respect to each other?
// Both count in vmcnt; the hand-counted rings of conv3x3s.hip / conv_stem*.hip / conv3x3a.hip wait with partial counts.
//   A: R register loads from COLD lines (never touched: HBM), then M LDS-DMA loads from HOT lines (L2), s_waitcnt vmcnt(M): in order would
//      mean every register load has landed.  The destination registers are pre-set to a sentinel and snapshotted right after the wait.
//   B: M LDS-DMA loads from COLD lines into sentinel-filled LDS, then R register loads from HOT lines, s_waitcnt vmcnt(R - 4) (the streamed
//      kernel's pattern: copies older than the ring's loads, a ring wait D - 1 k-steps later), then the LDS words are read back.
// Counts the waves that saw a sentinel = a load that had NOT landed although the count said so under the in-order assumption.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/dma_order.hip -o tools/experiments/dma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int SENT = 0x7fc0dead;

__device__ inline i32x4 mkdesc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)), (int)bytes, 0x00020000};
}

// cold: [nblk * rounds][8 KB] of distinct lines per (block, round); hot: 8 KB re-read by everybody.  Every buffer word holds its own word index.
// oob = 1: the copies are requested OUT OF RANGE (the way halo cells and rows outside the image are zero-filled): no memory access behind them
__global__ __launch_bounds__(64) void order_a(const int* __restrict__ cold, const int* __restrict__ hot, int rounds, int* __restrict__ stale, int oob)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int t = threadIdx.x;
    const i32x4 dc = mkdesc(cold, 0x7ffffff0u), dh = mkdesc(hot, 8192);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)sm;
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const int so = __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * rounds + r) * 8192));
        const int vo = t * 16;
        const int hoff = oob ? 0x40000000 : 0;
        int v0 = SENT, v1 = SENT, v2 = SENT, v3 = SENT;
        int snap;
        asm volatile(
            "buffer_load_dword %0, %5, %6, %7 offen\n\t"
            "buffer_load_dword %1, %5, %6, %7 offen offset:1024\n\t"
            "buffer_load_dword %2, %5, %6, %7 offen offset:2048\n\t"
            "buffer_load_dword %3, %5, %6, %7 offen offset:3072\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %5, %9, %10 offen lds\n\t"
            "buffer_load_dwordx4 %5, %9, %10 offen offset:1024 lds\n\t"
            "buffer_load_dwordx4 %5, %9, %10 offen offset:2048 lds\n\t"
            "buffer_load_dwordx4 %5, %9, %10 offen offset:3072 lds\n\t"
            "s_waitcnt vmcnt(4)\n\t"                       // in order: the four register loads (older) have landed
            "v_mov_b32 %4, %0\n\t"                          // snapshot the OLDEST load's destination (a read of a pending load's register sees the old value)
            "s_waitcnt vmcnt(0)"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&v"(snap)
            : "v"(vo), "s"(dc), "s"(so), "s"(lds0), "s"(dh), "s"(hoff)
            : "memory");
        if (snap == SENT) ++bad;
        if (v0 != (so + vo) / 4 || v3 != (so + 3072 + vo) / 4) bad += 1000;            // (sanity: after vmcnt(0) the data is the word index)
        __syncthreads();
    }
    if (bad) atomicAdd(&stale[(oob ? 6 : 0) + (t == 0 ? 0 : 1)], bad >= 1000 ? 1000000 : bad);
}

__global__ __launch_bounds__(64) void order_b(const int* __restrict__ cold, const int* __restrict__ hot, int rounds, int* __restrict__ stale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int t = threadIdx.x;
    const i32x4 dc = mkdesc(cold, 0x7ffffff0u), dh = mkdesc(hot, 32768);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)sm;
    int* const sw = reinterpret_cast<int*>(sm);
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = t; i < 1024; i += 64) sw[i] = SENT;     // 4 KB of LDS = four copies
        __syncthreads();
        const int so = __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * rounds + r) * 8192));
        const int vo = t * 16;
        i32x4 v[16];
        asm volatile(
            "s_mov_b32 m0, %16\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %17, %18, %19 offen lds\n\t"
            "buffer_load_dwordx4 %17, %18, %19 offen offset:1024 lds\n\t"
            "buffer_load_dwordx4 %17, %18, %19 offen offset:2048 lds\n\t"
            "buffer_load_dwordx4 %17, %18, %19 offen offset:3072 lds\n\t"
            "buffer_load_dwordx4 %0, %17, %20, 0 offen\n\t"
            "buffer_load_dwordx4 %1, %17, %20, 0 offen offset:1024\n\t"
            "buffer_load_dwordx4 %2, %17, %20, 0 offen offset:2048\n\t"
            "buffer_load_dwordx4 %3, %17, %20, 0 offen offset:3072\n\t"
            "buffer_load_dwordx4 %4, %17, %20, 0 offen\n\t"
            "buffer_load_dwordx4 %5, %17, %20, 0 offen offset:1024\n\t"
            "buffer_load_dwordx4 %6, %17, %20, 0 offen offset:2048\n\t"
            "buffer_load_dwordx4 %7, %17, %20, 0 offen offset:3072\n\t"
            "buffer_load_dwordx4 %8, %17, %20, 0 offen\n\t"
            "buffer_load_dwordx4 %9, %17, %20, 0 offen offset:1024\n\t"
            "buffer_load_dwordx4 %10, %17, %20, 0 offen offset:2048\n\t"
            "buffer_load_dwordx4 %11, %17, %20, 0 offen offset:3072\n\t"
            "buffer_load_dwordx4 %12, %17, %20, 0 offen\n\t"
            "buffer_load_dwordx4 %13, %17, %20, 0 offen offset:1024\n\t"
            "buffer_load_dwordx4 %14, %17, %20, 0 offen offset:2048\n\t"
            "buffer_load_dwordx4 %15, %17, %20, 0 offen offset:3072\n\t"
            "s_waitcnt vmcnt(12)"                           // in order: the four copies (oldest) and the first four register loads have landed
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
              "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
            : "s"(lds0), "v"(vo), "s"(dc), "s"(so), "s"(dh)
            : "memory");
        const int w0 = sw[t * 4], w3 = sw[768 + t * 4];      // this lane's own first words of the first and the last copy
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[15]) :: "memory");
        if (w0 == SENT || w3 == SENT) ++bad;
        if (v[15][0] != (3072 + vo) / 4) bad += 1000;
        __syncthreads();
    }
    if (bad) atomicAdd(&stale[t == 0 ? 2 : 3], bad >= 1000 ? 1000000 : bad);
}

// C: the first forward stem kernel's ring turn exactly: 20 register loads (HOT: the weights) in flight, then 23 LDS-DMA copies (COLD: the
//    activations), then s_waitcnt vmcnt(16 + 23): in order would mean the four OLDEST register loads have landed.
__global__ __launch_bounds__(64) void order_c(const int* __restrict__ cold, const int* __restrict__ hot, int rounds, int* __restrict__ stale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int t = threadIdx.x;
    const i32x4 dc = mkdesc(cold, 0x7ffffff0u), dh = mkdesc(hot, 32768);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)sm;
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const int so = __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * rounds + r) * 8192));
        const int vo = t * 16, vo4 = t * 4;
        int v[20], snap0, snap3;
#pragma unroll
        for (int i = 0; i < 20; ++i) v[i] = SENT;
#pragma unroll
        for (int i = 0; i < 20; ++i)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(v[i]) : "v"(vo4), "s"(dh), "s"(i * 256) : "memory");
#pragma unroll
        for (int i = 0; i < 23; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds0 + (i & 7) * 1024), "v"(vo), "s"(dc), "s"(so + (i % 8) * 1024) : "memory");
        asm volatile("s_waitcnt vmcnt(39)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(snap0), "=&v"(snap3) : "v"(v[0]), "v"(v[3]) : "memory");
        if (snap0 == SENT || snap3 == SENT) ++bad;
#pragma unroll
        for (int i = 0; i < 20; ++i) asm volatile("" : "+v"(v[i]));
        if (v[19] != (19 * 256 + vo4) / 4) bad += 1000;
        __syncthreads();
    }
    if (bad) atomicAdd(&stale[t == 0 ? 4 : 5], bad >= 1000 ? 1000000 : bad);
}

int main()
{
    const int nblk = 1024, rounds = 64;
    const size_t coldb = (size_t)nblk * rounds * 8192;       // 512 MB
    int *cold, *hot, *stale;
    CK(hipMalloc(&cold, coldb)); CK(hipMalloc(&hot, 32768)); CK(hipMalloc(&stale, 64));
    int* h = (int*)malloc(coldb);
    for (size_t i = 0; i < coldb / 4; ++i) h[i] = (int)i;
    CK(hipMemcpy(cold, h, coldb, hipMemcpyHostToDevice)); CK(hipMemcpy(hot, h, 32768, hipMemcpyHostToDevice));
    int res[8];
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(stale, 0, 64));
        hipLaunchKernelGGL(order_a, dim3(nblk), dim3(64), 8192, 0, cold, hot, rounds, stale, 0);
        hipLaunchKernelGGL(order_a, dim3(nblk), dim3(64), 8192, 0, cold, hot, rounds, stale, 1);
        hipLaunchKernelGGL(order_b, dim3(nblk), dim3(64), 8192, 0, cold, hot, rounds, stale);
        hipLaunchKernelGGL(order_c, dim3(nblk), dim3(64), 8192, 0, cold, hot, rounds, stale);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(res, stale, 32, hipMemcpyDeviceToHost));
        printf("{\"waves\": %d, \"rounds_per_wave\": %d, \"A_register_load_not_landed_after_vmcnt\": {\"lane0\": %d, \"other_lanes\": %d}, "
               "\"B_copy_not_landed_after_vmcnt\": {\"lane0\": %d, \"other_lanes\": %d}, \"C_oldest_of_20_register_loads_not_landed_after_vmcnt39_behind_23_copies\": {\"lane0\": %d, \"other_lanes\": %d}, \"A_with_out_of_range_copies\": {\"lane0\": %d, \"other_lanes\": %d}}\n", nblk, rounds, res[0], res[1], res[2], res[3], res[4], res[5], res[6], res[7]);
    }
    return 0;
}
