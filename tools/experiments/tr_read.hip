// Discover the semantics of ds_read_b64_tr_b16 on gfx950: LDS[i] = i (u16); print what each lane gets.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int mode, unsigned short* out) {
    __shared__ volatile unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr_elems;
    if (mode == 0) addr_elems = 0;                                   // uniform
    else if (mode == 1) addr_elems = l * 4;                          // contiguous 8 B per lane
    else if (mode == 2) addr_elems = (l & 15) * 64 + (l >> 4) * 4;   // row per lane (stride 64 elems), 4 elems per lane group
    else if (mode == 3) addr_elems = (l & 15) * 4 + (l >> 4) * 256;  // 16 lanes contiguous, groups far apart
    else addr_elems = (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 512;   // 4x4 blocks
    unsigned base = (unsigned)(size_t)(&lds[0]);
    unsigned byte_addr = base + addr_elems * 2;
    if (l == 0 && mode == 0) printf("lds base %u lds[5]=%d\n", base, (int)lds[5]);
    typedef __attribute__((ext_vector_type(2))) unsigned v2u;
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(byte_addr) : "memory");
    out[l * 4 + 0] = r.x & 0xffff; out[l * 4 + 1] = r.x >> 16; out[l * 4 + 2] = r.y & 0xffff; out[l * 4 + 3] = r.y >> 16;
}
int main() {
    unsigned short* d; if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) { printf("malloc failed\n"); return 1; }
    int ndev = 0; hipGetDeviceCount(&ndev); printf("devices %d\n", ndev);
    unsigned short h[256];
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        hipError_t e1 = hipDeviceSynchronize(); hipError_t e2 = hipGetLastError();
        hipError_t e3 = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("sync=%s last=%s copy=%s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
        printf("mode %d\n", mode); if (mode == 0) continue;
        for (int l = 0; l < 64; ++l) { printf(" l%02d:[%4d %4d %4d %4d]", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 8 == 7) printf("\n"); }
    }
    return 0;
}
