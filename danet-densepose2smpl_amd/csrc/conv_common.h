// Types shared by the convolution kernels (gfx950, bf16 storage, fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>

namespace danet_conv {

typedef unsigned short bf16_t;                                       // storage type
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;           // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;             // MFMA C/D fragment (16x16)

__device__ inline bf16_t f2bf(float f) {                             // round to nearest even
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ inline float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }

// replicas of per-channel BatchNorm accumulators: block b adds into replica b % BN_NCOPY
constexpr int BN_NCOPY = 32;

struct ConvP {
    const bf16_t* x; const bf16_t* w; const float* bias; void* y;
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, dil, groups, transposed;
    int Cin_g, Cout_g, Cout_pad, K, Kp;
    int relu, out_fp32, sshift, parity;
    float* stats;      // optional [BN_NCOPY][2][Cout] (pre-zeroed): per-channel sum / sum of squares of the bf16 output
    long M;
};

}  // namespace danet_conv
