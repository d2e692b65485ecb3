// How much does the shape of a wave's store matter?  A conv epilogue holds, per lane (li = pixel, lg), 4 consecutive channels of
// each of NT channel tiles.  Pattern A (today): per tile one 8-byte store, a pixel's 64 channels are written by 4 instructions x
// 4 lanes as 32-byte pieces.  Pattern B (channel-permuted weights): a lane owns 16 consecutive channels: two 16-byte stores, a
// pixel's 128-byte line is written by 4 lanes of ONE instruction pair.  Pattern C: plain coalesced 16 B per lane (reference).
// Writes `bytes` per launch; prints GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// C channels per pixel (bf16), M pixels; a workgroup of 4 waves covers 256 pixels x 64 channels per iteration
template <int PAT>
__global__ __launch_bounds__(256) void k(unsigned short* y, int M, int C)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((long)M * C * 2), 0x00020000);
    const int nb = C / 64;
    const long tiles = (long)(M / 256) * nb;
    for (long tt = blockIdx.x; tt < tiles; tt += gridDim.x) {
        const int mb = (int)(tt / nb), cb = (int)(tt % nb);
        for (int mt = 0; mt < 4; ++mt) {
            const int pix = mb * 256 + wave * 64 + mt * 16 + li;
            const int base = (pix * C + cb * 64) * 2;
            if (PAT == 0) {
                for (int nt = 0; nt < 4; ++nt) {
                    const i32x2 v = {pix, nt};
                    __builtin_amdgcn_raw_buffer_store_b64(v, yr, base + (nt * 16 + lg * 4) * 2, 0, 0);
                }
            } else if (PAT == 1) {
                for (int h = 0; h < 2; ++h) {
                    const i32x4 v = {pix, h, lg, 0};
                    __builtin_amdgcn_raw_buffer_store_b128(v, yr, base + (lg * 16 + h * 8) * 2, 0, 0);
                }
            } else {
                // fully linear: lane l of instruction i writes 16 B at (i * 64 + l) * 16 within the wave's 64-pixel x 64-channel block
                for (int h = 0; h < 2; ++h) {
                    const i32x4 v = {pix, h, lg, 0};
                    const int wbase = ((mb * 256 + wave * 64 + mt * 16) * C + cb * 64) * 2;
                    __builtin_amdgcn_raw_buffer_store_b128(v, yr, C == 64 ? wbase + (h * 64 + lane) * 16 : base + (lg * 16 + h * 8) * 2, 0, 0);
                }
            }
        }
    }
}

int main()
{
    const int M = 32 * 64 * 64;
    for (int C : {64, 256}) {
        unsigned short* y;
        CK(hipMalloc(&y, (size_t)M * C * 2));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int pat = 0; pat < 3; ++pat)
            for (int grid : {512, 1024, 2048}) {
                float best = 1e9;
                for (int it = 0; it < 6; ++it) {
                    hipEventRecord(e0);
                    for (int r = 0; r < 10; ++r) {
                        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, y, M, C);
                        if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, y, M, C);
                        if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, y, M, C);
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms / 10 < best) best = ms / 10;
                }
                printf("C=%d pattern %d grid %d: %.1f us  %.0f GB/s\n", C, pat, grid, best * 1e3, (double)M * C * 2 / (best * 1e-3) / 1e9);
            }
        hipFree(y);
    }
    return 0;
}
