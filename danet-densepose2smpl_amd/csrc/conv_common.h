// Types shared by the convolution kernels (gfx950, bf16 storage, fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>

namespace danet_conv {

typedef unsigned short bf16_t;                                       // storage type
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;           // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;             // MFMA C/D fragment (16x16)

// fp32 -> bf16, round to nearest even: one v_cvt_pk_bf16_f32 (gfx950) for two values
typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ inline unsigned f2bf_pk(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{lo, hi}, bf16x2_));
}
__device__ inline bf16_t f2bf(float f) { return (bf16_t)(f2bf_pk(f, 0.f) & 0xffffu); }
__device__ inline float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }

// Replicas of per-channel BatchNorm accumulators ([BN_NCOPY][2][C] workspaces): block b adds into replica b % bn_ncopy(C)
// (same-address atomic contention was 52 % of the step with one copy).  Every consumer block sums the replicas again,
// so a wide layer keeps fewer of them: the product replicas x channels -- what a consumer block reads before it can
// start -- stays ~2 K values (with 32 copies a block of the 384-channel BatchNorm read 98 KB to normalise 1.5 KB).
constexpr int BN_NCOPY = 32;                        // copies a workspace has room for
#ifndef DANET_BN_NCOPY_DIV
#define DANET_BN_NCOPY_DIV 2
#endif
// (round 5: with 8-byte accumulators half as many replicas for the narrow layers -- 16 / 8 / 4 / 4 -- measured -0.25 ms/step against
// 32 / 16 / 8 / 4: a consumer workgroup's prologue reads replicas x 2C x 8 bytes before it can start; DANET_BN_NCOPY_DIV = 1 / 2 / 4: A-B builds)
__host__ __device__ inline int bn_ncopy(int C) {
    const int n = (C <= 64 ? 32 : (C <= 128 ? 16 : (C <= 256 ? 8 : 4))) / DANET_BN_NCOPY_DIV;
    return n < 4 ? 4 : n;
}   // (8/4 and 16/8/4 measured the same step time)

// The accumulators are DOUBLES (DANET_BN_ACC32: floats, the round-1..4 form, kept for A-B timing).  A workgroup's partial sum is an
// fp32 value; up to 2^4 of them (workgroups / replicas) are added into one replica with global_atomic_add_f64.  Every such
// addition is EXACT while the non-zero partials of a channel lie within 2^25 of each other in magnitude (24-bit significands,
// 4 bits of carries, 53-bit accumulator), and exact additions commute: the replica's value -- and with it every BatchNorm
// statistic of the step -- does not depend on the order in which the workgroups arrive.  (fp32 atomics did: last-bit
// differences between two executions of the same step, which a deep random-weight net amplifies to per cents)
// Consumers add the replicas in index order in double precision and round once.  Partials further apart than 2^25 lose
// bits below 2^-29 of the sum, i.e. could change the fp32 result by one ulp with probability ~2^-29 per sum.
#ifdef DANET_BN_ACC32
typedef float bn_acc_t;
#else
typedef double bn_acc_t;
#endif
constexpr int BN_ACC_FLOATS = (int)(sizeof(bn_acc_t) / sizeof(float));       // workspace floats per accumulator
// ws: a [BN_NCOPY][2][Ctot] accumulator workspace (danet_bn_ws_floats(Ctot) floats, 8-byte aligned, zeroed); adds v to
// channel c of row `which` (0: first sum, 1: second) of replica rep % bn_ncopy(Ctot).  A GLOBAL (address space 1) atomic:
// a flat one would count against the LDS waits of the calling kernel.
__device__ inline void bn_acc_add(float* ws, int rep, int which, int Ctot, int c, float v) {
    bn_acc_t* const p = reinterpret_cast<bn_acc_t*>(ws) + ((size_t)(rep % bn_ncopy(Ctot)) * 2 + which) * Ctot + c;
    __hip_atomic_fetch_add((__attribute__((address_space(1))) bn_acc_t*)p, (bn_acc_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The BatchNorm (+ residual) (+ ReLU) a convolution launch applies to its own output after a grid-wide barrier (conv3x3s.hip
// s3_bn_tail; /root/reference/models/module/res_module.py:39-56 conv -> bn -> relu): what norm_act.hip's bn_apply_body does in a
// launch of its own, with the same arithmetic -- out = [relu](fmaf(y, sc, sh) [+ res]), sc = invstd * gamma, sh = fmaf(-mean, sc, beta).
struct BnApply {
    const bf16_t* res; bf16_t* out;                 // residual (optional) and result, both shaped like the convolution's output
    const float* gamma; const float* beta;          // [C]
    float* running_mean; float* running_var;        // [C], optional (updated with `momentum`)
    float* saved;                                   // [2][C]: mean, invstd (the backward pass reads them)
    unsigned char* mask;                            // optional: the ReLU gate, one byte per four channels (norm_act.hip stmask)
    int relu;
    float momentum, eps;                            // (one value per launch: the first problem's)
    unsigned* bar;                                  // grid-barrier state (grid_barrier.h; one per launch: the first problem's)
};

struct ConvP {
    const bf16_t* x; const bf16_t* w; const float* bias; void* y;
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, dil, groups, transposed;
    int Cin_g, Cout_g, Cout_pad, K, Kp;
    int relu, out_fp32, sshift, parity;
    float* stats;      // optional [BN_NCOPY][2][Cout] (pre-zeroed): per-channel sum / sum of squares of the bf16 output
    long M;
    // optional (data-gradient launches): reduce the BatchNorm-backward sums of the BN that produced this conv's
    // input into bn_red [BN_NCOPY][2][Cout] while the gradient tile is still in registers: sum(dy') and
    // sum(dy' * xhat) with dy' = dy * (bn_y > 0) under ReLU, xhat = (bn_x - mean) * invstd, bn_saved = [mean[Cout] | invstd[Cout]]
    const bf16_t* bn_x; const bf16_t* bn_y; const float* bn_saved; float* bn_red;
    // ReLU gate of that reduction: 0 = from bn_y (NULL: no ReLU); 2 = bn_y is the byte mask the BatchNorm forward wrote
    // (norm_act.hip; LDS-tile 3x3 kernel only: one byte per lane instead of eight).
    int bn_gate;
    long x_bytes, y_bytes;   // extents of the gathered / written tensors (buffer resources of conv_fast.hip)
    const bf16_t* addend;    // optional (3x3 LDS kernels and conv_fast.hip; bf16 outputs): bf16 tensor shaped like y, added before rounding
    const BnApply* bna;   // optional (streamed 3x3 kernel, forward with `stats`): the training-mode BatchNorm that follows, applied by the same launch
};


// conv_fast.hip: the lean kernel for the common cases (conv_igemm.hip keeps the general one)
bool conv_fast_ok(const ConvP& p, bool vec8, int mt);
int conv_fast_launch(const ConvP& p, int mt, int nt, void* stream);
int conv_fast_launch_multi(const ConvP* ps, const int* mts, int n, int nt, void* stream);

// conv_pw.hip: 1x1 / stride-1 layers with the whole weight operand in LDS and persistent workgroups (X read once, Y written once)
bool conv_pw_ok(const ConvP& p, bool vec8);
int conv_pw_config(const ConvP& p);                  // NKS * 10 + NTB of the instantiation, 0 = not supported
int conv_pw_launch(const ConvP& p, void* stream);

// One problem of the multi-problem weight-gradient entry points (include/danet_hip.h danet_conv_wgrad_multi)
struct WgJob { const void* x; const void* dy; float* dw; int B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups; };
// conv_pw_wgrad.hip: weight gradients of 1x1 / stride-1 layers (chunks staged as they lie, LDS transpose reads, deterministic partials)
bool conv_pw_wgrad_ok(const WgJob& j);
size_t conv_pw_wgrad_ws_floats(const WgJob* jobs, const int* idx, int cnt, long target);
int conv_pw_wgrad_launch(const WgJob* jobs, const int* idx, int cnt, float* ws, float beta, long target, void* stream);

// conv_g3.hip: grouped 3x3 / stride-1 layers with narrow groups (the 24-group partial-IUV head: 48 -> 24 forward, 24 -> 48 data gradient)
bool conv_g3_ok(const ConvP& p, bool vec8);
int conv_g3_launch(const ConvP& p, void* stream);

// conv3x3.hip: 3x3 / stride-1 / pad-1 forward and data gradient on an LDS-resident halo tile (persistent workgroups)
bool conv3x3_ok(const ConvP& p, bool vec8);
int conv3x3_config(const ConvP& p, bool vec8, int nprob);      // MT*100 + NT*10 + KW, 0 = not supported
int conv3x3_launch(const ConvP* ps, int n, void* stream, bool dry = false);      // n <= 4 problems in one launch; dry: only check

// conv3x3s.hip: the streamed successor (LDS-DMA stage copies into a two-slot LDS ring); conv3x3_launch tries it first.  0 = launched (or, dry, would be)
int conv3x3s_launch(const ConvP* ps, int n, void* stream, bool dry);
bool conv3x3_stream_first();
int* conv3x3_debug_buffer();

// Run-time knobs (A-B timing, tests) behind the ONE entry point danet_knob (include/danet_hip.h, DANET_KNOB_*): every translation
// unit keeps its own switches and answers for its ids; value < 0 only queries; the previous value is returned.
long conv3x3_knob(int id, long value);          // conv3x3.hip:  C3_ENABLE, C3_MT, C3_KW, C3_BLOCKS, C3_WANT
long conv3x3s_knob(int id, long value);         // conv3x3s.hip: C3S_ENABLE, C3S_BLOCKS, C3S_KW, C3S_WANT
long conv_pw_knob(long value);
long conv_pw_wgrad_knob(long value);
long conv_stem_knob(long value);
long conv_stem_dgrad_knob(long value);
long conv3x3a_knob(long value);
long conv_g3_knob(long value);
long bn_block_bytes_knob(long value);        // danet_conv3x3_debug's buffer (NULL: off); the streamed kernel writes 16 ints per workgroup

}  // namespace danet_conv
