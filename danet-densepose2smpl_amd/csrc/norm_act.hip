// BatchNorm (+ReLU, +residual add) and the HRNet multi-resolution fuse, on NHWC bf16.
//
// Replaces the ~300 BatchNorm2d / ReLU / add / nearest-Upsample launches per forward of the
// reference's backbone (/root/reference/models/module/hr_module.py:111-177,
// res_module.py:27-97).  All kernels are HBM-bound streams over the flat [M*C] array: a lane
// owns one 16-byte run of 8 channels whose channel index never changes across its grid-stride
// loop (the stride is a multiple of C/8), so per-channel statistics live in registers.
//
//   bn_stats        per-channel sum / sum of squares (fp32, block-reduced, one atomic per block)
//   bn_apply        y = [relu]( (x-mean)*invstd*gamma + beta [+ res] ); also writes mean/invstd
//                   for the backward and updates the running statistics (training)
//   bn_bwd_reduce   per-channel sum(dy') and sum(dy' * xhat), dy' = dy * (y > 0) under ReLU
//   bn_bwd_apply    dx = gamma*invstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)), d_res = dy'
//   sum_relu        y = relu(sum_t nearest_upsample_{f_t}(in_t)) (HRNet fuse layer), and its
//                   per-term backward (window sum of the masked gradient)
#include "common.h"
#include "conv_common.h"
#include "grid_barrier.h"

namespace {

using namespace danet_conv;

constexpr int NCOPY = danet_conv::BN_NCOPY;   // replicas of the per-channel accumulators: block b adds into replica b % NCOPY (cuts same-address atomic contention)
constexpr int VW = 4;     // channels per lane (8-byte runs in bf16, 16-byte runs in fp32; every BN width on the path is a multiple of 4)

// This file is compiled twice: as it is for the bf16 production path, and through norm_act_f32.hip (NA_F32) for fp32 NHWC
// tensors -- BASELINE config C4's arithmetic type, the same kernels with 4-byte elements, exported with an _f32 suffix.
#ifdef NA_F32
typedef float elem_t;
constexpr int ES = 4;         // bytes per element
constexpr int MSH = 4;        // byte offset -> ReLU-mask byte (one per lane-run of VW channels)
#define NA_NAME(n) n##_f32
#else
typedef bf16_t elem_t;
constexpr int ES = 2;
constexpr int MSH = 3;
#define NA_NAME(n) n
#endif

struct Vec { float v[VW]; };

#ifdef NA_F32
__device__ inline Vec load_bf(const elem_t* p) {
    const float4 r = *reinterpret_cast<const float4*>(p);
    Vec o; o.v[0] = r.x; o.v[1] = r.y; o.v[2] = r.z; o.v[3] = r.w;
    return o;
}
__device__ inline void store_bf(elem_t* p, const Vec& a) { *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }
#else
__device__ inline Vec load_bf(const elem_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    Vec o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    return o;
}
__device__ inline void store_bf(elem_t* p, const Vec& a) {
    uint2 r;
    r.x = f2bf_pk(a.v[0], a.v[1]);
    r.y = f2bf_pk(a.v[2], a.v[3]);
    *reinterpret_cast<uint2*>(p) = r;
}
#endif

// Flat mapping: vector id v = blockIdx.x*span + t + k*gridspan, span = RY*CV (CV = C/VW).
// Requires t < span; then v % CV == t % CV for every k: a lane keeps its channel vector cv = t % CV and
// walks rows row0, row0 + rstep, ...  All offsets are 32-bit byte offsets into buffer resources (the
// host checks the tensors are < 2 GB): rows past the end get an out-of-range offset, for which the
// hardware returns zeros on loads and drops stores -- no bounds branches in the loops.
struct FlatMap { int M; int CV; int span; int rstep; int ldv; int coff; int bytes; };

typedef __attribute__((ext_vector_type(2))) int i32x2;
constexpr int OOB = 0x7fffffff;
constexpr int UNR = 4;       // rows per trip; the next trip's loads are issued before the current trip's stores

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#ifdef NA_F32
typedef __attribute__((ext_vector_type(4))) int i32x4_;
__device__ inline Vec ldv(__amdgpu_buffer_rsrc_t r, int off) {
    const i32x4_ q = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    Vec o;
#pragma unroll
    for (int j = 0; j < VW; ++j) o.v[j] = __int_as_float(q[j]);
    return o;
}
__device__ inline void stv(__amdgpu_buffer_rsrc_t r, int off, const Vec& a) {
    const i32x4_ q = {__float_as_int(a.v[0]), __float_as_int(a.v[1]), __float_as_int(a.v[2]), __float_as_int(a.v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, off, 0, 0);
}
#else
__device__ inline Vec ldv(__amdgpu_buffer_rsrc_t r, int off) {
    const i32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    Vec o;
    o.v[0] = __uint_as_float((unsigned)q.x << 16); o.v[1] = __uint_as_float((unsigned)q.x & 0xffff0000u);
    o.v[2] = __uint_as_float((unsigned)q.y << 16); o.v[3] = __uint_as_float((unsigned)q.y & 0xffff0000u);
    return o;
}
__device__ inline void stv(__amdgpu_buffer_rsrc_t r, int off, const Vec& a) {
    const i32x2 q = {(int)f2bf_pk(a.v[0], a.v[1]), (int)f2bf_pk(a.v[2], a.v[3])};
    __builtin_amdgcn_raw_buffer_store_b64(q, r, off, 0, 0);
}
#endif

// ReLU gate of the backward passes (mask_mode): 0 = from the saved output (y > 0); 1 = from the byte mask the forward
// wrote (one byte per lane-run of VW channels, bit j = y_j > 0: the residual case, where y cannot be recomputed
// without reading the residual); 2 = recomputed from x with the forward's own expression fmaf(x, sc, sh) (no
// residual).  Modes 1 and 2 save the read of y: 1/5 .. 1/4 of the backward traffic.
__device__ inline int ldmask(__amdgpu_buffer_rsrc_t r, int off) { return (int)__builtin_amdgcn_raw_buffer_load_b8(r, off >> MSH, 0, 0); }
__device__ inline void stmask(__amdgpu_buffer_rsrc_t r, int off, int m) { __builtin_amdgcn_raw_buffer_store_b8((unsigned char)m, r, off >> MSH, 0, 0); }

struct RowIter {
    int row, off, rstep, ostride, M;
    __device__ inline void init(const FlatMap& fm, int bid, int t, int& cv) {
        cv = t % fm.CV;
        row = (bid * fm.span + t) / fm.CV;
        off = ((row * fm.ldv + fm.coff + cv) * VW) * ES;
        rstep = fm.rstep; ostride = fm.rstep * fm.ldv * VW * ES; M = fm.M;
    }
    __device__ inline int offset(int u) const { return row + u * rstep < M ? off + u * ostride : OOB; }
    __device__ inline bool more() const { return row < M; }
    __device__ inline void next() { row += rstep * UNR; off += ostride * UNR; }
};

using danet_conv::bn_acc_t;
template <typename T>
__device__ inline void block_channel_reduce(float (*sm)[VW], Vec a, int t, int CV, int span, T* dst /* [C] */) {
    // sm: [256][VW]; lanes with equal (t % CV) are summed in a fixed order, then one atomic per channel (T = bn_acc_t: a
    // double accumulator, whose value does not depend on the order of these atomics -- conv_common.h)
    if (t < span) {
#pragma unroll
        for (int j = 0; j < VW; ++j) sm[t][j] = a.v[j];
    }
    __syncthreads();
    if (t < CV) {
        Vec s;
#pragma unroll
        for (int j = 0; j < VW; ++j) s.v[j] = 0.f;
        for (int u = t; u < span; u += CV)
#pragma unroll
            for (int j = 0; j < VW; ++j) s.v[j] += sm[u][j];
#pragma unroll
        for (int j = 0; j < VW; ++j)
            __hip_atomic_fetch_add((__attribute__((address_space(1))) T*)(dst + t * VW + j), (T)s.v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

constexpr int SLAB = 1024;   // channels per launch (wider tensors are processed in channel slabs)

// Sum the NCOPY replicas of a [2][Cst] accumulator for the Cs channels of this slab into LDS: the
// 2*Cs sums are spread over the block's lanes, each issuing NCOPY independent loads (fixed order).
__device__ inline void reduce_replicas(const bn_acc_t* __restrict__ rep, int Cst, int Cs, int t, float (*sStat)[SLAB]) {
    const int ncopy = danet_conv::bn_ncopy(Cst);             // 4 .. 32 (a multiple of 4): the replicas this width uses
    for (int i = t; i < 2 * Cs; i += 256) {
        const int which = i >= Cs ? 1 : 0, c = i - which * Cs;
        bn_acc_t s = 0;
        for (int r0 = 0; r0 < ncopy; r0 += 4) {
            bn_acc_t v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rep[(size_t)(r0 + r) * 2 * Cst + (size_t)which * Cst + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) s += v[r];
        }
        sStat[which][c] = (float)s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const elem_t* __restrict__ x, FlatMap fm, bn_acc_t* __restrict__ sums /* [2][Cst] */, int Cst)
{
    __shared__ float sm[256][VW];
    const int t = threadIdx.x;
    Vec s, q;
#pragma unroll
    for (int j = 0; j < VW; ++j) { s.v[j] = 0.f; q.v[j] = 0.f; }
    if (t < fm.span) {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, fm.bytes);
        int cv; RowIter it; it.init(fm, blockIdx.x, t, cv);
#ifdef NA_F32
        // fp32 tensors: sums of x - k with k = the channel's value in row 0 (bn_apply_body adds it back), so that
        // E[(x-k)^2] - E[x-k]^2 does not cancel for channels whose mean is large against their spread
        const Vec k = ldv(xr, (fm.coff + cv) * VW * ES);
#endif
        for (; it.more(); it.next()) {
            Vec a[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) a[u] = ldv(xr, it.offset(u));
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int j = 0; j < VW; ++j) {
#ifdef NA_F32
                    const float d = it.offset(u) != OOB ? a[u].v[j] - k.v[j] : 0.f;       // (rows past the end load zeros)
                    s.v[j] += d; q.v[j] += d * d;
#else
                    s.v[j] += a[u].v[j]; q.v[j] += a[u].v[j] * a[u].v[j];
#endif
                }
        }
    }
    bn_acc_t* dst = sums + (size_t)(blockIdx.x % danet_conv::bn_ncopy(Cst)) * 2 * Cst;
    block_channel_reduce(sm, s, t, fm.CV, fm.span, dst);
    block_channel_reduce(sm, q, t, fm.CV, fm.span, dst + Cst);
}

// per-channel sums of a [M, C] tensor (bias gradients): one atomic per channel and block into out[C] (DOUBLES, zeroed by the host: the
// additions of the workgroups' fp32 partial sums are exact, so the totals do not depend on their order -- conv_common.h)
__global__ __launch_bounds__(256) void channel_sum_kernel(const elem_t* __restrict__ x, FlatMap fm, double* __restrict__ out, int ncopy, int Ctot)
{
    __shared__ float sm[256][VW];
    const int t = threadIdx.x;
    Vec s;
#pragma unroll
    for (int j = 0; j < VW; ++j) s.v[j] = 0.f;
    if (t < fm.span) {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, fm.bytes);
        int cv; RowIter it; it.init(fm, blockIdx.x, t, cv);
        for (; it.more(); it.next()) {
            Vec a[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) a[u] = ldv(xr, it.offset(u));              // (rows past the end load zeros)
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int j = 0; j < VW; ++j) s.v[j] += a[u].v[j];
        }
    }
    block_channel_reduce(sm, s, t, fm.CV, fm.span, out + (size_t)(blockIdx.x % ncopy) * Ctot);
}

// mode 0: training (sums -> mean/invstd, update running stats); mode 1: eval (running stats)
__device__ __forceinline__ void bn_apply_body(
    const int bid, const elem_t* __restrict__ x, const elem_t* __restrict__ res, elem_t* __restrict__ y, const FlatMap& fm,
    const bn_acc_t* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ saved /* [2][Cst] mean, invstd */,
    int Cst, float inv_count, float unbias, float momentum, float eps, int mode, int relu, unsigned char* __restrict__ mask)
{
    const int t = threadIdx.x;
    const int C = Cst;
    __shared__ float sStat[2][SLAB];
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, fm.bytes), rr = make_rsrc(res ? res : x, fm.bytes), yr = make_rsrc(y, fm.bytes);
    const __amdgpu_buffer_rsrc_t mr = make_rsrc(mask ? (const void*)mask : (const void*)x, mask ? fm.bytes >> MSH : 0);
    int cv = 0; RowIter it; it.init(fm, bid, t < fm.span ? t : 0, cv);
    Vec a[UNR], r[UNR];
    int o[UNR];
    auto fetch = [&]() {
#pragma unroll
        for (int u = 0; u < UNR; ++u) { o[u] = it.offset(u); a[u] = ldv(xr, o[u]); }
        if (res) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) r[u] = ldv(rr, o[u]);
        }
    };
    if (t < fm.span) fetch();                       // first rows are in flight while the statistics are reduced
#ifdef NA_F32
    Vec kshift;                                     // bn_stats_kernel's shift: the channel's value in row 0
    if (t < fm.span && mode == 0) kshift = ldv(xr, (fm.coff + cv) * VW * ES);
#endif
    if (mode == 0) reduce_replicas(sums, C, fm.CV * VW, t, sStat);
    if (t >= fm.span) return;
    const int c0 = cv * VW;
    float sc[VW], sh[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        float mean, var;
        if (mode == 0) {
            mean = sStat[0][c0 + j] * inv_count;
            var = fmaxf(sStat[1][c0 + j] * inv_count - mean * mean, 0.f);
#ifdef NA_F32
            mean += kshift.v[j];
#endif
        } else {
            mean = running_mean[c0 + j];
            var = running_var[c0 + j];
        }
#ifdef NA_F32
        const float invstd = 1.0f / sqrtf(var + eps);           // correctly rounded, as the reference's fp32 BatchNorm computes it
#else
        const float invstd = rsqrtf(var + eps);
#endif
        const float g = gamma ? gamma[c0 + j] : 1.f, b = beta ? beta[c0 + j] : 0.f;
        sc[j] = invstd * g;
        sh[j] = fmaf(-mean, sc[j], b);
        if (mode == 0 && bid == 0 && t < fm.CV) {
            if (saved) { saved[c0 + j] = mean; saved[C + c0 + j] = invstd; }
            if (running_mean) {
                running_mean[c0 + j] = (1.f - momentum) * running_mean[c0 + j] + momentum * mean;
                running_var[c0 + j] = (1.f - momentum) * running_var[c0 + j] + momentum * var * unbias;
            }
        }
    }
    while (it.more()) {
        Vec outv[UNR];
        int oo[UNR], mb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            oo[u] = o[u];
            mb[u] = 0;
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                float v = fmaf(a[u].v[j], sc[j], sh[j]);           // (the backward's mask_mode 2 repeats exactly this)
                if (res) v += r[u].v[j];
                mb[u] |= (v > 0.f ? 1 : 0) << j;
                outv[u].v[j] = relu ? fmaxf(v, 0.f) : v;
            }
        }
        it.next();
        if (it.more()) fetch();
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            stv(yr, oo[u], outv[u]);
            if (mask) stmask(mr, oo[u], mb[u]);
        }
    }
}

// GATE: where the ReLU gate comes from -- 0 / 1 / 2 = mask_mode (see ldmask), 3 = no ReLU.  A compile-time constant: with the
// mode chosen per element through a run-time select chain the fp32 build lost the gated value altogether (the compiler
// folded `on ? dy : 0` to 0 in bn_bwd_reduce_kernel), and the specialised loops are shorter anyway.
template <int GATE>
__device__ __forceinline__ void bn_bwd_reduce_body_g(
    const int bid, const elem_t* __restrict__ dy, const elem_t* __restrict__ x, const elem_t* __restrict__ y, const FlatMap& fm,
    const float* __restrict__ saved, int Cst, bn_acc_t* __restrict__ red /* [2][Cst]: sum dy', sum dy'*xhat */,
    const unsigned char* __restrict__ mask, const float* __restrict__ gamma, const float* __restrict__ beta, float (*sm)[VW])
{
    constexpr int relu = GATE != 3, mask_mode = GATE == 3 ? 0 : GATE;
    const int t = threadIdx.x;
    const int C = Cst;
    Vec s1, s2;
#pragma unroll
    for (int j = 0; j < VW; ++j) { s1.v[j] = 0.f; s2.v[j] = 0.f; }
    if (t < fm.span) {
        const bool from_y = relu && mask_mode == 0;
        const __amdgpu_buffer_rsrc_t gr = make_rsrc(dy, fm.bytes), xr = make_rsrc(x, fm.bytes), yr = make_rsrc(from_y ? y : x, fm.bytes);
        const __amdgpu_buffer_rsrc_t mr = make_rsrc(mask_mode == 1 ? (const void*)mask : (const void*)x, mask_mode == 1 ? fm.bytes >> MSH : 0);
        int cv; RowIter it; it.init(fm, bid, t, cv);
        const int c0 = cv * VW;
        float mean[VW], invstd[VW], sc[VW], sh[VW];
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            mean[j] = saved[c0 + j]; invstd[j] = saved[C + c0 + j];
            sc[j] = invstd[j] * (gamma ? gamma[c0 + j] : 1.f);
            sh[j] = fmaf(-mean[j], sc[j], beta ? beta[c0 + j] : 0.f);
        }
        for (; it.more(); it.next()) {
            Vec g[UNR], a[UNR], o[UNR];
            int mb[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int off = it.offset(u);
                g[u] = ldv(gr, off);
                a[u] = ldv(xr, off);
                if (from_y) o[u] = ldv(yr, off);
                if (relu && mask_mode == 1) mb[u] = ldmask(mr, off);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const bool on = !relu || (mask_mode == 0 ? o[u].v[j] > 0.f : mask_mode == 1 ? ((mb[u] >> j) & 1) != 0
                                                                                                 : fmaf(a[u].v[j], sc[j], sh[j]) > 0.f);
                    const float gv = on ? g[u].v[j] : 0.f;        // rows past the end load dy = 0
                    s1.v[j] += gv; s2.v[j] += gv * (a[u].v[j] - mean[j]) * invstd[j];
                }
        }
    }
    bn_acc_t* dst = red + (size_t)(bid % danet_conv::bn_ncopy(C)) * 2 * C;
    block_channel_reduce(sm, s1, t, fm.CV, fm.span, dst);
    block_channel_reduce(sm, s2, t, fm.CV, fm.span, dst + C);
}
__device__ __forceinline__ void bn_bwd_reduce_body(
    const int bid, const elem_t* __restrict__ dy, const elem_t* __restrict__ x, const elem_t* __restrict__ y, const FlatMap& fm,
    const float* __restrict__ saved, int Cst, int relu, bn_acc_t* __restrict__ red,
    int mask_mode, const unsigned char* __restrict__ mask, const float* __restrict__ gamma, const float* __restrict__ beta)
{
    __shared__ float sm[256][VW];
    switch (relu ? mask_mode : 3) {
        case 0: bn_bwd_reduce_body_g<0>(bid, dy, x, y, fm, saved, Cst, red, mask, gamma, beta, sm); break;
        case 1: bn_bwd_reduce_body_g<1>(bid, dy, x, y, fm, saved, Cst, red, mask, gamma, beta, sm); break;
        case 2: bn_bwd_reduce_body_g<2>(bid, dy, x, y, fm, saved, Cst, red, mask, gamma, beta, sm); break;
        default: bn_bwd_reduce_body_g<3>(bid, dy, x, y, fm, saved, Cst, red, mask, gamma, beta, sm); break;
    }
}

template <int GATE>
__device__ __forceinline__ void bn_bwd_apply_body_g(
    const int bid, const elem_t* __restrict__ dy, const elem_t* __restrict__ x, const elem_t* __restrict__ y, const FlatMap& fm,
    const float* __restrict__ saved, const float* __restrict__ gamma, const bn_acc_t* __restrict__ red,
    int Cst, float inv_count, elem_t* __restrict__ dx, elem_t* __restrict__ dres, float* __restrict__ dparam,
    const unsigned char* __restrict__ mask, const float* __restrict__ beta, float (*sStat)[SLAB])
{
    constexpr int relu = GATE != 3, mask_mode = GATE == 3 ? 0 : GATE;
    const int t = threadIdx.x;
    const int C = Cst;
    const bool from_y = relu && mask_mode == 0;
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(dy, fm.bytes), xr = make_rsrc(x, fm.bytes), yr = make_rsrc(from_y ? y : x, fm.bytes);
    const __amdgpu_buffer_rsrc_t mr = make_rsrc(mask_mode == 1 ? (const void*)mask : (const void*)x, mask_mode == 1 ? fm.bytes >> MSH : 0);
    const __amdgpu_buffer_rsrc_t dxr = make_rsrc(dx, fm.bytes), drr = make_rsrc(dres ? dres : dx, fm.bytes);
    int cv = 0; RowIter it; it.init(fm, bid, t < fm.span ? t : 0, cv);
    Vec g[UNR], a[UNR], o[UNR];
    int of[UNR], mb[UNR];
    auto fetch = [&]() {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            of[u] = it.offset(u);
            g[u] = ldv(gr, of[u]);
            a[u] = ldv(xr, of[u]);
            if (from_y) o[u] = ldv(yr, of[u]);
            if (relu && mask_mode == 1) mb[u] = ldmask(mr, of[u]);
        }
    };
    if (t < fm.span) fetch();
    reduce_replicas(red, C, fm.CV * VW, t, sStat);
    if (t >= fm.span) return;
    const int c0 = cv * VW;
    float mean[VW], invstd[VW], k0[VW], m1[VW], m2[VW], sc[VW], sh[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        mean[j] = saved[c0 + j]; invstd[j] = saved[C + c0 + j];
        k0[j] = (gamma ? gamma[c0 + j] : 1.f) * invstd[j];
        sc[j] = invstd[j] * (gamma ? gamma[c0 + j] : 1.f);
        sh[j] = fmaf(-mean[j], sc[j], beta ? beta[c0 + j] : 0.f);
        const float s0 = sStat[0][c0 + j], s1 = sStat[1][c0 + j];
        m1[j] = s0 * inv_count; m2[j] = s1 * inv_count;
        if (bid == 0 && t < fm.CV && dparam) { dparam[c0 + j] = s0; dparam[C + c0 + j] = s1; }
    }
    while (it.more()) {
        Vec d[UNR], gm[UNR];
        int oo[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            oo[u] = of[u];
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                const bool on = !relu || (mask_mode == 0 ? o[u].v[j] > 0.f : mask_mode == 1 ? ((mb[u] >> j) & 1) != 0
                                                                                             : fmaf(a[u].v[j], sc[j], sh[j]) > 0.f);
                const float gv = on ? g[u].v[j] : 0.f;
                gm[u].v[j] = gv;
                d[u].v[j] = k0[j] * (gv - m1[j] - (a[u].v[j] - mean[j]) * invstd[j] * m2[j]);
            }
        }
        it.next();
        if (it.more()) fetch();
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (dres) stv(drr, oo[u], gm[u]);
            stv(dxr, oo[u], d[u]);
        }
    }
}
__device__ __forceinline__ void bn_bwd_apply_body(
    const int bid, const elem_t* __restrict__ dy, const elem_t* __restrict__ x, const elem_t* __restrict__ y, const FlatMap& fm,
    const float* __restrict__ saved, const float* __restrict__ gamma, const bn_acc_t* __restrict__ red,
    int Cst, float inv_count, int relu, elem_t* __restrict__ dx, elem_t* __restrict__ dres, float* __restrict__ dparam,
    int mask_mode, const unsigned char* __restrict__ mask, const float* __restrict__ beta)
{
    __shared__ float sStat[2][SLAB];
    switch (relu ? mask_mode : 3) {
        case 0: bn_bwd_apply_body_g<0>(bid, dy, x, y, fm, saved, gamma, red, Cst, inv_count, dx, dres, dparam, mask, beta, sStat); break;
        case 1: bn_bwd_apply_body_g<1>(bid, dy, x, y, fm, saved, gamma, red, Cst, inv_count, dx, dres, dparam, mask, beta, sStat); break;
        case 2: bn_bwd_apply_body_g<2>(bid, dy, x, y, fm, saved, gamma, red, Cst, inv_count, dx, dres, dparam, mask, beta, sStat); break;
        default: bn_bwd_apply_body_g<3>(bid, dy, x, y, fm, saved, gamma, red, Cst, inv_count, dx, dres, dparam, mask, beta, sStat); break;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(
    const elem_t* __restrict__ x, const elem_t* __restrict__ res, elem_t* __restrict__ y, FlatMap fm,
    const bn_acc_t* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ saved,
    int Cst, float inv_count, float unbias, float momentum, float eps, int mode, int relu, unsigned char* __restrict__ mask)
{
    bn_apply_body(blockIdx.x, x, res, y, fm, sums, gamma, beta, running_mean, running_var, saved, Cst, inv_count, unbias, momentum, eps, mode, relu, mask);
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const elem_t* __restrict__ dy, const elem_t* __restrict__ x, const elem_t* __restrict__ y, FlatMap fm,
    const float* __restrict__ saved, int Cst, int relu, bn_acc_t* __restrict__ red,
    int mask_mode, const unsigned char* __restrict__ mask, const float* __restrict__ gamma, const float* __restrict__ beta)
{
    bn_bwd_reduce_body(blockIdx.x, dy, x, y, fm, saved, Cst, relu, red, mask_mode, mask, gamma, beta);
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const elem_t* __restrict__ dy, const elem_t* __restrict__ x, const elem_t* __restrict__ y, FlatMap fm,
    const float* __restrict__ saved, const float* __restrict__ gamma, const bn_acc_t* __restrict__ red,
    int Cst, float inv_count, int relu, elem_t* __restrict__ dx, elem_t* __restrict__ dres, float* __restrict__ dparam,
    int mask_mode, const unsigned char* __restrict__ mask, const float* __restrict__ beta)
{
    bn_bwd_apply_body(blockIdx.x, dy, x, y, fm, saved, gamma, red, Cst, inv_count, relu, dx, dres, dparam, mask_mode, mask, beta);
}

// Up to 4 independent BatchNorms in one launch (the HRNet branches advance in lockstep: nn.multi_batch_norm): the
// small branches' launches are dominated by the per-launch floor, one launch over all of them is not.
constexpr int NBM = 12;
struct BnFwdOne {
    const elem_t* x; const elem_t* res; elem_t* y; const bn_acc_t* sums; const float* gamma; const float* beta;
    float* running_mean; float* running_var; float* saved; unsigned char* mask; FlatMap fm; int C; float inv_count, unbias; int relu;
};
struct BnFwdMulti { BnFwdOne a[NBM]; int start[NBM + 1]; int n; float momentum, eps; int mode; };
__global__ __launch_bounds__(256) void bn_apply_multi_kernel(BnFwdMulti m)
{
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const BnFwdOne& a = m.a[i];
    bn_apply_body(blockIdx.x - m.start[i], a.x, a.res, a.y, a.fm, a.sums, a.gamma, a.beta, a.running_mean, a.running_var, a.saved,
                  a.C, a.inv_count, a.unbias, m.momentum, m.eps, m.mode, a.relu, a.mask);
}
struct BnBwdOne {
    const elem_t* dy; const elem_t* x; const elem_t* y; const float* saved; const float* gamma; bn_acc_t* red;
    elem_t* dx; elem_t* dres; float* dparam; const float* beta; const unsigned char* mask;
    FlatMap fm; int C; float inv_count; int relu; int have_red; int mask_mode;
};
struct BnBwdMulti { BnBwdOne a[NBM]; int start[NBM + 1]; int n; };
__global__ __launch_bounds__(256) void bn_bwd_reduce_multi_kernel(BnBwdMulti m)
{
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const BnBwdOne& a = m.a[i];
    if (a.have_red) return;                              // already reduced by the consumer conv's data-gradient epilogue
    bn_bwd_reduce_body(blockIdx.x - m.start[i], a.dy, a.x, a.y, a.fm, a.saved, a.C, a.relu, a.red, a.mask_mode, a.mask, a.gamma, a.beta);
}
__global__ __launch_bounds__(256) void bn_bwd_apply_multi_kernel(BnBwdMulti m)
{
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const BnBwdOne& a = m.a[i];
    bn_bwd_apply_body(blockIdx.x - m.start[i], a.dy, a.x, a.y, a.fm, a.saved, a.gamma, a.red, a.C, a.inv_count, a.relu, a.dx, a.dres, a.dparam,
                      a.mask_mode, a.mask, a.beta);
}

#ifndef NA_F32      // (bf16 only: the register / LDS budget of the one-pass form is sized for 2-byte elements)
// ------------------------------------------------------------------------------------------
// One-pass BatchNorm backward: the two-kernel form reads dy and x twice (once for the per-channel sums, once to apply
// them).  Here every lane keeps its share of dy / x (and the gate) in registers across a grid-wide barrier: phase 1
// accumulates sum(dy') and sum(dy' * xhat) (block reduction, one atomic per channel and block), all workgroups meet at
// the barrier, phase 2 applies the sums to the registers and stores dx (and d_res).  Traffic drops from 5..6 to 3..4
// tensor passes.  Conditions (host): at most OP_NV rows per lane with a grid of at most OP_MAX_BLOCKS workgroups -- which
// all fit on the chip at once (2 workgroups per CU), so the barrier cannot wait for a workgroup that has
// not started -- and launches of this kernel never overlap each other (the caller issues them on ONE stream).  The spin
// is bounded: if the barrier is not met in time the launch sets an error flag instead of hanging (results are then
// garbage; bar[2] reports it).
constexpr int OP_NV = 10;             // rows (8-byte channel vectors of dy and x) per lane held in registers: the instantiation for the whole device ...
constexpr int OP_NV_BIG = 14;         // ... and the one a smaller co-residency budget asks for (177 instead of 128 registers; its four extra row slots cost a
                                      // full-size launch ~1.5 us when they are not needed, so the planner only picks it when it saves a launch)
constexpr int OP_NL = 15;             // ... and in the lane's private LDS slots (17 bytes per row: 64 KB per workgroup; with the 12 KB of
                                      // reduction scratch two workgroups fit a CU's 160 KB)
#ifndef DANET_OP_CH
#define DANET_OP_CH 5
#endif
constexpr int OP_CH = DANET_OP_CH;              // LDS rows loaded per batch
static_assert(OP_NL % OP_CH == 0, "OP_NL");
constexpr int OP_ROWS = OP_NV + OP_NL, OP_ROWS_BIG = OP_NV_BIG + OP_NL;     // 25 / 29 rows per lane
constexpr int OP_MAX_BLOCKS = 512;    // 2 workgroups per CU (<= 256 VGPRs each)

using danet::grid_barrier;
constexpr int OP_BAR_WORDS = danet::GRID_BAR_WORDS;

// reduce_replicas with agent-scope loads (sc1: served by L2, never by this CU's L1): the sums other workgroups added
// before a grid barrier.  The loads of a lane are independent buffer loads (a loop of __hip_atomic_load was serialised
// one round trip each: 30 us per launch).
__device__ inline void ld_acc_sc1(__amdgpu_buffer_rsrc_t rr, int off, double& v) { v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, off, 0, 16 /* sc1 */)); }
__device__ inline void ld_acc_sc1(__amdgpu_buffer_rsrc_t rr, int off, float& v) { v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off, 0, 16 /* sc1 */)); }
__device__ inline void reduce_replicas_sc1(const bn_acc_t* __restrict__ rep, int Cst, int Cs, int t, float (*sStat)[SLAB]) {
    const int ncopy = danet_conv::bn_ncopy(Cst);
    constexpr int AB = (int)sizeof(bn_acc_t);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(rep, (int)((size_t)NCOPY * 2 * Cst * AB));
    for (int i = t; i < 2 * Cs; i += 256) {
        const int which = i >= Cs ? 1 : 0, c = i - which * Cs;
        bn_acc_t s = 0;
        for (int r0 = 0; r0 < ncopy; r0 += 4) {
            bn_acc_t v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (((r0 + r) * 2 + which) * Cst + c) * AB;
                ld_acc_sc1(rr, off, v[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s += v[r];
        }
        sStat[which][c] = (float)s;
    }
    __syncthreads();
}

struct BnOnePass { BnBwdOne a[NBM]; int start[NBM + 1]; int n; unsigned* bar; int dbg; int big; };

template <int NV>
__global__ __launch_bounds__(256, 2) void bn_bwd_onepass_kernel(BnOnePass m)
{
    int ji = 0;
    while (ji + 1 < m.n && (int)blockIdx.x >= m.start[ji + 1]) ++ji;
    const BnBwdOne& a = m.a[ji];
    const int bid = blockIdx.x - m.start[ji];
    const FlatMap& fm = a.fm;
    const int t = threadIdx.x, C = a.C;
    __shared__ float sm[256][VW];
    __shared__ float sStat[2][SLAB];
    extern __shared__ __attribute__((aligned(16))) unsigned char op_smem[];
    uint4 (*sRow)[256] = reinterpret_cast<uint4 (*)[256]>(op_smem);                              // [OP_NL][256] {dy, x} rows
    unsigned char (*sGate)[256] = reinterpret_cast<unsigned char (*)[256]>(op_smem + OP_NL * 256 * 16);   // [OP_NL][256] gate bits
    const bool live = t < fm.span;
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(a.dy, fm.bytes), xr = make_rsrc(a.x, fm.bytes);
    const __amdgpu_buffer_rsrc_t mr = make_rsrc(a.mask_mode == 1 ? (const void*)a.mask : (const void*)a.x, a.mask_mode == 1 ? fm.bytes >> MSH : 0);
    const __amdgpu_buffer_rsrc_t dxr = make_rsrc(a.dx, fm.bytes), drr = make_rsrc(a.dres ? a.dres : a.dx, fm.bytes);
    int cv = 0; RowIter it; it.init(fm, bid, live ? t : 0, cv);
    const int c0 = cv * VW;
    // ---- phase 0: everything this lane owns goes into registers (independent loads, all in flight)
    // (no branches inside the unrolled loops: wave-uniform options become selects / out-of-range offsets, otherwise
    // the loops split into ~100 basic blocks and the register allocator spills)
    i32x2 gq[NV], xq[NV];
    unsigned gate[(NV + 7) / 8];                // 4 gate bits per row, 8 rows per register
#pragma unroll
    for (int w = 0; w < (NV + 7) / 8; ++w) gate[w] = 0u;
    const bool use_mask = a.relu && a.mask_mode == 1, recompute = a.relu && a.mask_mode == 2;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int off = live ? it.offset(k) : OOB;
        gq[k] = __builtin_amdgcn_raw_buffer_load_b64(gr, off, 0, 0);
        xq[k] = __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0);
        gate[k / 8] |= ((unsigned)ldmask(mr, use_mask ? off : OOB) & 15u) << (4 * (k % 8));      // (zero-sized resource otherwise: 0)
    }
    // the rows beyond the register budget go to this lane's own LDS slots, OP_CH at a time (2 * OP_CH more loads in flight)
    for (int l0 = 0; l0 < OP_NL; l0 += OP_CH) {
        i32x2 tg[OP_CH], tx[OP_CH];
        int tm[OP_CH];
#pragma unroll
        for (int u = 0; u < OP_CH; ++u) {
            const int off = live ? it.offset(NV + l0 + u) : OOB;
            tg[u] = __builtin_amdgcn_raw_buffer_load_b64(gr, off, 0, 0);
            tx[u] = __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0);
            tm[u] = ldmask(mr, use_mask ? off : OOB);
        }
#pragma unroll
        for (int u = 0; u < OP_CH; ++u) {
            sRow[l0 + u][t] = uint4{(unsigned)tg[u].x, (unsigned)tg[u].y, (unsigned)tx[u].x, (unsigned)tx[u].y};
            sGate[l0 + u][t] = (unsigned char)(tm[u] & 15);
        }
    }
    float mean[VW], invstd[VW], sc[VW], sh[VW], gam[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        mean[j] = a.saved[c0 + j]; invstd[j] = a.saved[C + c0 + j];
        gam[j] = a.gamma ? a.gamma[c0 + j] : 1.f;
        sc[j] = invstd[j] * gam[j];
        sh[j] = fmaf(-mean[j], sc[j], a.beta ? a.beta[c0 + j] : 0.f);
    }
    auto unpack = [](const i32x2 q, float* v) {
        v[0] = __uint_as_float((unsigned)q.x << 16); v[1] = __uint_as_float((unsigned)q.x & 0xffff0000u);
        v[2] = __uint_as_float((unsigned)q.y << 16); v[3] = __uint_as_float((unsigned)q.y & 0xffff0000u);
    };
    // ---- phase 1: per-channel sums of the masked gradient (rows past the end loaded zeros)
    Vec s1, s2;
#pragma unroll
    for (int j = 0; j < VW; ++j) { s1.v[j] = 0.f; s2.v[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float g[VW], x[VW];
        unpack(gq[k], g); unpack(xq[k], x);
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < VW; ++j) bits |= (fmaf(x[j], sc[j], sh[j]) > 0.f ? 1u : 0u) << j;
        gate[k / 8] |= (!a.relu ? 15u : (recompute ? bits : 0u)) << (4 * (k % 8));
        const unsigned mbk = gate[k / 8] >> (4 * (k % 8));
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float gv = (mbk >> j) & 1 ? g[j] : 0.f;
            s1.v[j] += gv; s2.v[j] = fmaf(gv * (x[j] - mean[j]), invstd[j], s2.v[j]);
        }
        __builtin_amdgcn_sched_barrier(0);        // one row's temporaries at a time (else the scheduler unpacks all rows up front)
    }
#pragma unroll 2
    for (int l = 0; l < OP_NL; ++l) {
        const uint4 rw = sRow[l][t];
        float g[VW], x[VW];
        unpack(i32x2{(int)rw.x, (int)rw.y}, g); unpack(i32x2{(int)rw.z, (int)rw.w}, x);
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < VW; ++j) bits |= (fmaf(x[j], sc[j], sh[j]) > 0.f ? 1u : 0u) << j;
        const unsigned mbk = !a.relu ? 15u : (recompute ? bits : (unsigned)sGate[l][t]);
        sGate[l][t] = (unsigned char)mbk;
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float gv = (mbk >> j) & 1 ? g[j] : 0.f;
            s1.v[j] += gv; s2.v[j] = fmaf(gv * (x[j] - mean[j]), invstd[j], s2.v[j]);
        }
    }
    // the packed registers cross the barrier as they are: without this the compiler keeps the UNPACKED floats of phase 1
    // alive for phase 2 (14 instead of 4 registers per row)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int q0 = gq[k].x, q1 = gq[k].y, q2 = xq[k].x, q3 = xq[k].y;
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        gq[k].x = q0; gq[k].y = q1; xq[k].x = q2; xq[k].y = q3;
    }
    bn_acc_t* dst = a.red + (size_t)(bid % danet_conv::bn_ncopy(C)) * 2 * C;
    block_channel_reduce(sm, s1, t, fm.CV, fm.span, dst);
    block_channel_reduce(sm, s2, t, fm.CV, fm.span, dst + C);
    if (!(m.dbg & 1)) grid_barrier(m.bar, gridDim.x, 0x30000u + gridDim.x);        // (error code: which launch gave up, by its grid)
    if (m.dbg & 2) return;
    // ---- phase 2: the complete sums (agent-scope loads), then dx / d_res from the registers
    reduce_replicas_sc1(a.red, C, fm.CV * VW, t, sStat);
    if (!live) return;
    float k0[VW], m1[VW], m2[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        const float q0 = sStat[0][c0 + j], q1 = sStat[1][c0 + j];
        k0[j] = gam[j] * invstd[j];
        m1[j] = q0 * a.inv_count; m2[j] = q1 * a.inv_count;
        if (bid == 0 && t < fm.CV && a.dparam) { a.dparam[c0 + j] = q0; a.dparam[C + c0 + j] = q1; }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int off = it.offset(k);
        float g[VW], x[VW];
        unpack(gq[k], g); unpack(xq[k], x);
        Vec d, gm;
        const unsigned mbk = gate[k / 8] >> (4 * (k % 8));
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float gv = (mbk >> j) & 1 ? g[j] : 0.f;
            gm.v[j] = gv;
            d.v[j] = k0[j] * (gv - m1[j] - (x[j] - mean[j]) * invstd[j] * m2[j]);
        }
        stv(drr, a.dres ? off : OOB, gm);
        stv(dxr, off, d);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 2
    for (int l = 0; l < OP_NL; ++l) {
        const int off = it.offset(NV + l);
        const uint4 rw = sRow[l][t];
        float g[VW], x[VW];
        unpack(i32x2{(int)rw.x, (int)rw.y}, g); unpack(i32x2{(int)rw.z, (int)rw.w}, x);
        const unsigned mbk = sGate[l][t];
        Vec d, gm;
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float gv = (mbk >> j) & 1 ? g[j] : 0.f;
            gm.v[j] = gv;
            d.v[j] = k0[j] * (gv - m1[j] - (x[j] - mean[j]) * invstd[j] * m2[j]);
        }
        stv(drr, a.dres ? off : OOB, gm);
        stv(dxr, off, d);
    }
}

#endif  // NA_F32

// ------------------------------------------------------------------------------------------
struct SumP {
    const elem_t* in[4]; int shift[4]; int nterms;
    int B, H, W, C;
};

__device__ __forceinline__ void sum_relu_body(const SumP& p, elem_t* __restrict__ y, int relu, const int lb, const int nb)
{
    const int CV = p.C / VW;
    const long nvec = (long)p.B * p.H * p.W * CV;
    for (long v = (long)lb * 256 + threadIdx.x; v < nvec; v += (long)nb * 256) {
        const int cv = (int)(v % CV);
        long pix = v / CV;
        const int w = (int)(pix % p.W); pix /= p.W;
        const int h = (int)(pix % p.H);
        const int b = (int)(pix / p.H);
        Vec s;
#pragma unroll
        for (int j = 0; j < VW; ++j) s.v[j] = 0.f;
        for (int tt = 0; tt < p.nterms; ++tt) {
            const int sh = p.shift[tt];
            const int Ht = p.H >> sh, Wt = p.W >> sh;
            const Vec a = load_bf(p.in[tt] + ((((size_t)b * Ht + (h >> sh)) * Wt + (w >> sh)) * CV + cv) * VW);
#pragma unroll
            for (int j = 0; j < VW; ++j) s.v[j] += a.v[j];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < VW; ++j) s.v[j] = fmaxf(s.v[j], 0.f);
        }
        store_bf(y + v * VW, s);
    }
}
__global__ __launch_bounds__(256) void sum_relu_kernel(SumP p, elem_t* __restrict__ y, int relu)
{
    sum_relu_body(p, y, relu, blockIdx.x, gridDim.x);
}
// The fuse sums of ONE HighResolutionModule (up to four outputs, one per branch) in one launch: they are independent of each other
// and the low-resolution ones are far too small to fill a launch of their own (4.9 - 11.6 us each at B = 32, mostly launch).
constexpr int NSM = 4;
struct SumMulti { SumP p[NSM]; elem_t* y[NSM]; int relu[NSM]; int start[NSM + 1]; int n; };
__global__ __launch_bounds__(256) void sum_relu_multi_kernel(SumMulti m)
{
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    sum_relu_body(m.p[i], m.y[i], m.relu[i], blockIdx.x - m.start[i], m.start[i + 1] - m.start[i]);
}

// d_term[b,h',w',c] = sum over the 2^sh x 2^sh window of gy * (y > 0)
__global__ __launch_bounds__(256) void sum_relu_bwd_kernel(const elem_t* __restrict__ gy, const elem_t* __restrict__ y,
                                                           int B, int H, int W, int C, int sh, int relu, elem_t* __restrict__ d)
{
    const int CV = C / VW;
    const int Ht = H >> sh, Wt = W >> sh, f = 1 << sh;
    const long nvec = (long)B * Ht * Wt * CV;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const int cv = (int)(v % CV);
        long pix = v / CV;
        const int w = (int)(pix % Wt); pix /= Wt;
        const int h = (int)(pix % Ht);
        const int b = (int)(pix / Ht);
        Vec s;
#pragma unroll
        for (int j = 0; j < VW; ++j) s.v[j] = 0.f;
        for (int dh = 0; dh < f; ++dh)
            for (int dw = 0; dw < f; ++dw) {
                const size_t off = ((((size_t)b * H + (h * f + dh)) * W + (w * f + dw)) * CV + cv) * VW;
                const Vec g = load_bf(gy + off);
                if (relu) {
                    const Vec o = load_bf(y + off);
#pragma unroll
                    for (int j = 0; j < VW; ++j) s.v[j] += o.v[j] > 0.f ? g.v[j] : 0.f;
                } else {
#pragma unroll
                    for (int j = 0; j < VW; ++j) s.v[j] += g.v[j];
                }
            }
        store_bf(d + v * VW, s);
    }
}

// All shifts of one fuse output at once: d_s[b, h >> s, w >> s, c] = window sums of gy * (y > 0) for every requested s <= smax.
// A lane owns one coarsest window (2^smax squared pixels) of one channel vector and walks it in Morton order, so every 2x2,
// 4x4, 8x8 group completes consecutively: gy and y are read ONCE (the per-shift kernel read them once per shift: 4 launches
// and 8 tensor passes for the highest-resolution output of a 4-branch module).
struct SumBwdAll { elem_t* d[4]; };
__device__ __forceinline__ void sum_relu_bwd_all_body(const elem_t* __restrict__ gy, const elem_t* __restrict__ y,
                                                      int B, int H, int W, int C, int smax, int relu, const SumBwdAll& out, const int lb, const int nb)
{
    const int CV = C / VW;
    const int F = 1 << smax, Hc = H >> smax, Wc = W >> smax;
    const long nvec = (long)B * Hc * Wc * CV;
    for (long v = (long)lb * 256 + threadIdx.x; v < nvec; v += (long)nb * 256) {
        const int cv = (int)(v % CV);
        long pix = v / CV;
        const int wc = (int)(pix % Wc); pix /= Wc;
        const int hc = (int)(pix % Hc);
        const int b = (int)(pix / Hc);
        Vec a1, a2, a3;
#pragma unroll
        for (int j = 0; j < VW; ++j) { a1.v[j] = 0.f; a2.v[j] = 0.f; a3.v[j] = 0.f; }
        for (int m = 0; m < F * F; ++m) {
            const int dw = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4), dh = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
            const int h = hc * F + dh, w = wc * F + dw;
            const size_t off = ((((size_t)b * H + h) * W + w) * CV + cv) * VW;
            Vec g = load_bf(gy + off);
            if (relu) {
                const Vec o = load_bf(y + off);
#pragma unroll
                for (int j = 0; j < VW; ++j) g.v[j] = o.v[j] > 0.f ? g.v[j] : 0.f;
            }
            if (out.d[0]) store_bf(out.d[0] + off, g);
#pragma unroll
            for (int j = 0; j < VW; ++j) a1.v[j] += g.v[j];
            if ((m & 3) == 3) {
                if (out.d[1]) store_bf(out.d[1] + ((((size_t)b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * CV + cv) * VW, a1);
#pragma unroll
                for (int j = 0; j < VW; ++j) { a2.v[j] += a1.v[j]; a1.v[j] = 0.f; }
                if ((m & 15) == 15) {
                    if (out.d[2]) store_bf(out.d[2] + ((((size_t)b * (H >> 2) + (h >> 2)) * (W >> 2) + (w >> 2)) * CV + cv) * VW, a2);
#pragma unroll
                    for (int j = 0; j < VW; ++j) { a3.v[j] += a2.v[j]; a2.v[j] = 0.f; }
                    if ((m & 63) == 63) {
                        if (out.d[3]) store_bf(out.d[3] + ((((size_t)b * (H >> 3) + (h >> 3)) * (W >> 3) + (w >> 3)) * CV + cv) * VW, a3);
#pragma unroll
                        for (int j = 0; j < VW; ++j) a3.v[j] = 0.f;
                    }
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void sum_relu_bwd_all_kernel(const elem_t* __restrict__ gy, const elem_t* __restrict__ y,
                                                               int B, int H, int W, int C, int smax, int relu, SumBwdAll out)
{
    sum_relu_bwd_all_body(gy, y, B, H, W, C, smax, relu, out, blockIdx.x, gridDim.x);
}
struct SumBwdMultiOne { const elem_t* gy; const elem_t* y; int B, H, W, C, smax, relu; SumBwdAll out; };
struct SumBwdMulti { SumBwdMultiOne a[NSM]; int start[NSM + 1]; int n; };
__global__ __launch_bounds__(256) void sum_relu_bwd_all_multi_kernel(SumBwdMulti m)
{
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const SumBwdMultiOne& a = m.a[i];
    sum_relu_bwd_all_body(a.gy, a.y, a.B, a.H, a.W, a.C, a.smax, a.relu, a.out, blockIdx.x - m.start[i], m.start[i + 1] - m.start[i]);
}

}  // namespace
// Bytes of the tensor a workgroup of the BatchNorm kernels handles at least (danet_bn_set_block_bytes; both instantiations
// of this file share the one setting, which lives in the bf16 translation unit).
#ifdef NA_F32
long danet_bn_block_bytes();
#else
static long g_bn_block_bytes = getenv("DANET_BN_BLOCK_BYTES") ? atol(getenv("DANET_BN_BLOCK_BYTES")) : 24576;
long danet_bn_block_bytes() { return g_bn_block_bytes; }
// Run-time knob (A-B timing, tests): bytes <= 0 keeps; returns the previous value.  With a value above every tensor's size each
// BatchNorm launch is ONE workgroup per tensor: its float sums then have a fixed order (the replicated atomics of larger grids
// do not), which tests that compare two executions of a chaotic deep net need.
long danet_conv::bn_block_bytes_knob(long bytes) { const long prev = g_bn_block_bytes; if (bytes > 0) g_bn_block_bytes = bytes; return prev; }
#endif
namespace {
// slab [c_begin, c_begin + Cs) of a [M, C] tensor; Cs <= 1024
inline int make_map(int64_t M, int C, int c_begin, int Cs, FlatMap* fm, int* grid) {
    if (C % VW != 0 || Cs % VW != 0 || c_begin % VW != 0 || Cs / VW > 256) return -1;
    if (M * C * ES >= (1LL << 31) - (64 << 20)) return -1;          // 32-bit byte offsets
    fm->CV = Cs / VW;
    fm->ldv = C / VW;
    fm->coff = c_begin / VW;
    fm->M = (int)M;
    fm->bytes = (int)(M * C * ES);
    fm->span = (256 / fm->CV) * fm->CV;
    const long nvec = M * fm->CV;
    long blocks = (nvec + fm->span - 1) / fm->span;
    // a block reads the replicas of its slab's statistics before it can start (~8 KB): at least ~24 KB of the tensor each
    // (tools/experiments/bn_micro.cpp: the four-branch forward 24.7 -> 12.9 us, the two-kernel backward 81 -> 40 us together
    // with bn_ncopy; eight rows per trip instead of four measured slower: 14.8 / 62 us)
    const long per_block = danet_bn_block_bytes();
    const long by_bytes = (M * Cs * ES + per_block - 1) / per_block;
    if (blocks > by_bytes) blocks = by_bytes;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    *grid = (int)blocks;
    fm->rstep = (int)((long)fm->span * blocks / fm->CV);
    return 0;
}
// slabs of a tensor wider than one slab that one multi-problem launch can take (1: not wide, or not a whole number of slabs)
inline int wide_slabs(int C) {
    static const bool off = getenv("DANET_BN_WIDE") && atoi(getenv("DANET_BN_WIDE")) == 0;       // A-B timing
    return !off && C > SLAB && C % SLAB == 0 && C / SLAB <= 12 ? C / SLAB : 1;
}

}  // namespace

extern "C" int NA_NAME(danet_bn_forward)(const void* x, const void* res, void* y, int64_t M, int C,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float* saved, float* sums_ws, int ws_is_zero, float momentum, float eps, int training, int relu,
                                void* relu_mask, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && y && M > 0 && C > 0, "bn_forward: bad arguments");
    DANET_CHECK_ARG(training ? (saved && sums_ws) : (running_mean && running_var), "bn_forward: missing buffers");
    DANET_CHECK_ARG(C % VW == 0, "bn_forward: C=%d must be a multiple of %d", C, VW);
#ifdef NA_F32
    DANET_CHECK_ARG(!training || ws_is_zero != 2, "bn_forward_f32: the fp32 statistics are shifted sums (bn_stats_kernel): no pre-accumulated sums");
#endif
    hipStream_t st = (hipStream_t)stream;
    if (training && !ws_is_zero) {
        hipError_t e = danet::zero_async(sums_ws, sizeof(bn_acc_t) * 2 * C * NCOPY, st);
        if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "bn_forward: memset: %s", hipGetErrorString(e));
    }
    const float inv = 1.0f / (float)M, unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    if (wide_slabs(C) > 1) {
        // a tensor wider than one slab is C / SLAB independent BatchNorms over the same rows: ONE launch over all of them (round 6: the
        // grouped layer4 of the limb branch, [32, 2, 2, 3072], took three 11 us launches per BatchNorm in a stretch of the step where
        // nothing else runs)
        const int ns = wide_slabs(C);
        BnFwdMulti m; m.n = ns; m.momentum = momentum; m.eps = eps; m.mode = training ? 0 : 1; m.start[0] = 0;
        for (int s = 0; s < ns; ++s) {
            const int c0 = s * SLAB;
            BnFwdOne& a = m.a[s];
            int grid;
            DANET_CHECK_ARG(make_map(M, C, c0, SLAB, &a.fm, &grid) == 0, "bn_forward: C=%d (M=%ld) unsupported: channels must be a multiple of 4 and the tensor < 2 GB", C, (long)M);
            if (training && ws_is_zero != 2) {
                hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)x, a.fm, (bn_acc_t*)sums_ws + c0, C);
                DANET_CHECK_LAUNCH("bn_stats_kernel");
            }
            a.x = (const elem_t*)x; a.res = (const elem_t*)res; a.y = (elem_t*)y; a.sums = sums_ws ? (const bn_acc_t*)sums_ws + c0 : nullptr;
            a.gamma = gamma ? gamma + c0 : nullptr; a.beta = beta ? beta + c0 : nullptr;
            a.running_mean = running_mean ? running_mean + c0 : nullptr; a.running_var = running_var ? running_var + c0 : nullptr;
            a.saved = saved ? saved + c0 : nullptr; a.mask = (unsigned char*)relu_mask; a.C = C; a.inv_count = inv; a.unbias = unbias; a.relu = relu;
            m.start[s + 1] = m.start[s] + grid;
        }
        hipLaunchKernelGGL(bn_apply_multi_kernel, dim3(m.start[ns]), dim3(256), 0, st, m);
        DANET_CHECK_LAUNCH("bn_apply_multi_kernel");
        return DANET_OK;
    }
    for (int c0 = 0; c0 < C; c0 += SLAB) {
        const int Cs = C - c0 < SLAB ? C - c0 : SLAB;
        FlatMap fm; int grid;
        DANET_CHECK_ARG(make_map(M, C, c0, Cs, &fm, &grid) == 0, "bn_forward: C=%d (M=%ld) unsupported: channels must be a multiple of 4 and the tensor < 2 GB", C, (long)M);
        // per-slab views of the per-channel buffers: [2][C] buffers are addressed as base+c0 with stride C
        if (training && ws_is_zero != 2) {          // ws_is_zero == 2: the statistics were accumulated by the conv epilogue
            hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)x, fm, (bn_acc_t*)sums_ws + c0, C);
            DANET_CHECK_LAUNCH("bn_stats_kernel");
        }
        hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)x, (const elem_t*)res, (elem_t*)y, fm,
                           sums_ws ? (const bn_acc_t*)sums_ws + c0 : nullptr, gamma ? gamma + c0 : nullptr, beta ? beta + c0 : nullptr,
                           running_mean ? running_mean + c0 : nullptr, running_var ? running_var + c0 : nullptr,
                           saved ? saved + c0 : nullptr, C, inv, unbias, momentum, eps, training ? 0 : 1, relu, (unsigned char*)relu_mask);
        DANET_CHECK_LAUNCH("bn_apply_kernel");
    }
    return DANET_OK;
}

// red_ws: danet_bn_ws_floats(C) floats of scratch; dparam [2][C] (may be NULL) receives (d beta, d gamma).
#ifndef NA_F32
extern "C" size_t danet_bn_ws_floats(int C) { return (size_t)NCOPY * 2 * C * danet_conv::BN_ACC_FLOATS; }
extern "C" int danet_bn_acc_bytes(void) { return (int)sizeof(bn_acc_t); }
#endif

extern "C" int NA_NAME(danet_bn_backward)(const void* dy, const void* x, const void* y, int64_t M, int C,
                                 const float* gamma, const float* saved, int relu,
                                 void* dx, void* dres, float* dparam, float* red_ws, int ws_is_zero,
                                 int mask_mode, const void* relu_mask, const float* beta, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(dy && x && dx && saved && red_ws && M > 0 && C > 0, "bn_backward: bad arguments");
    DANET_CHECK_ARG(!relu || (mask_mode == 0 && y) || (mask_mode == 1 && relu_mask) || (mask_mode == 2 && !dres),
                    "bn_backward: ReLU gate: mask_mode 0 needs y, 1 the forward's byte mask, 2 (recompute from x) no residual");
    DANET_CHECK_ARG(C % VW == 0, "bn_backward: C=%d must be a multiple of %d", C, VW);
    hipStream_t st = (hipStream_t)stream;
    if (!ws_is_zero) {
        hipError_t e = danet::zero_async(red_ws, sizeof(bn_acc_t) * 2 * C * NCOPY, st);
        if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "bn_backward: memset: %s", hipGetErrorString(e));
    }
    if (wide_slabs(C) > 1) {                       // as in bn_forward: the slabs of a wide tensor in one launch per phase
        const int ns = wide_slabs(C);
        BnBwdMulti m; m.n = ns; m.start[0] = 0;
        for (int s = 0; s < ns; ++s) {
            const int c0 = s * SLAB;
            BnBwdOne& a = m.a[s];
            int grid;
            DANET_CHECK_ARG(make_map(M, C, c0, SLAB, &a.fm, &grid) == 0, "bn_backward: C=%d (M=%ld) unsupported: channels must be a multiple of 4 and the tensor < 2 GB", C, (long)M);
            a.dy = (const elem_t*)dy; a.x = (const elem_t*)x; a.y = (const elem_t*)y; a.saved = saved + c0; a.gamma = gamma ? gamma + c0 : nullptr;
            a.red = (bn_acc_t*)red_ws + c0; a.dx = (elem_t*)dx; a.dres = (elem_t*)dres; a.dparam = dparam ? dparam + c0 : nullptr; a.C = C;
            a.inv_count = 1.0f / (float)M; a.relu = relu; a.have_red = ws_is_zero == 2; a.beta = beta ? beta + c0 : nullptr;
            a.mask = (const unsigned char*)relu_mask; a.mask_mode = mask_mode;
            m.start[s + 1] = m.start[s] + grid;
        }
        if (ws_is_zero != 2) {
            hipLaunchKernelGGL(bn_bwd_reduce_multi_kernel, dim3(m.start[ns]), dim3(256), 0, st, m);
            DANET_CHECK_LAUNCH("bn_bwd_reduce_multi_kernel");
        }
        hipLaunchKernelGGL(bn_bwd_apply_multi_kernel, dim3(m.start[ns]), dim3(256), 0, st, m);
        DANET_CHECK_LAUNCH("bn_bwd_apply_multi_kernel");
        return DANET_OK;
    }
    for (int c0 = 0; c0 < C; c0 += SLAB) {
        const int Cs = C - c0 < SLAB ? C - c0 : SLAB;
        FlatMap fm; int grid;
        DANET_CHECK_ARG(make_map(M, C, c0, Cs, &fm, &grid) == 0, "bn_backward: C=%d (M=%ld) unsupported: channels must be a multiple of 4 and the tensor < 2 GB", C, (long)M);
        if (ws_is_zero != 2) {                     // ws_is_zero == 2: the sums were accumulated by the consumer conv's dgrad epilogue
            hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)dy, (const elem_t*)x, (const elem_t*)y,
                               fm, saved + c0, C, relu, (bn_acc_t*)red_ws + c0, mask_mode, (const unsigned char*)relu_mask,
                               gamma ? gamma + c0 : nullptr, beta ? beta + c0 : nullptr);
            DANET_CHECK_LAUNCH("bn_bwd_reduce_kernel");
        }
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)dy, (const elem_t*)x, (const elem_t*)y,
                           fm, saved + c0, gamma ? gamma + c0 : nullptr, (const bn_acc_t*)red_ws + c0, C, 1.0f / (float)M, relu, (elem_t*)dx, (elem_t*)dres,
                           dparam ? dparam + c0 : nullptr, mask_mode, (const unsigned char*)relu_mask, beta ? beta + c0 : nullptr);
        DANET_CHECK_LAUNCH("bn_bwd_apply_kernel");
    }
    return DANET_OK;
}

// out[C] (doubles) = sum over the M rows of x [M, C] (the bias gradient of a convolution: gy.sum(dim = (0, 2, 3))); out is zeroed
// here (memset node) and accumulated with one atomic per channel and workgroup.  C % 4 == 0, C <= 1024 per launch slab.
// ncopy (round 6; 1 = the old form): out holds ncopy REPLICAS [ncopy][C], workgroup b adds into replica b % ncopy and the caller sums the
// replicas -- with one copy 256 workgroups queued on the same C addresses and could not be made more (a 17 MB head gradient took 34 us:
// 0.5 TB/s, four waves per compute unit); with 16 copies the launch runs 1 024 workgroups.
extern "C" int NA_NAME(danet_channel_sum)(const void* x, int64_t M, int C, double* out, int out_is_zero, int ncopy, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && out && M > 0 && C > 0 && C % VW == 0 && ncopy >= 1 && ncopy <= 64, "channel_sum: bad arguments (C=%d must be a multiple of %d)", C, VW);
    hipStream_t st = (hipStream_t)stream;
    if (!out_is_zero) {
        hipError_t err = danet::zero_async(out, sizeof(double) * (size_t)C * ncopy, st);
        if (err != hipSuccess) return danet::fail(DANET_ERR_HIP, "channel_sum: memset: %s", hipGetErrorString(err));
    }
    for (int c0 = 0; c0 < C; c0 += SLAB) {
        const int Cs = C - c0 < SLAB ? C - c0 : SLAB;
        FlatMap fm; int grid;
        DANET_CHECK_ARG(make_map(M, C, c0, Cs, &fm, &grid) == 0, "channel_sum: C=%d (M=%ld) unsupported", C, (long)M);
        // every workgroup ends with one atomic per channel on the SAME C addresses (no replicas here): 1024 workgroups serialised
        // there; one workgroup per compute unit reads as fast and queues a quarter of the atomics (A-B in the step: within noise)
        const int cap = ncopy >= 4 ? 1024 : 256;
        if (grid > cap) { grid = cap; fm.rstep = (int)((long)fm.span * grid / fm.CV); }
        hipLaunchKernelGGL(channel_sum_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)x, fm, out + c0, ncopy, C);
        DANET_CHECK_LAUNCH("channel_sum_kernel");
    }
    return DANET_OK;
}

extern "C" int NA_NAME(danet_sum_relu_forward)(const void* const* terms /* host array */, const int* shifts /* host */, int nterms,
                                      int B, int H, int W, int C, int relu, void* y, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(terms && shifts && y && nterms >= 1 && nterms <= 4 && C % VW == 0, "sum_relu_forward: bad arguments");
    SumP p;
    p.nterms = nterms; p.B = B; p.H = H; p.W = W; p.C = C;
    for (int i = 0; i < 4; ++i) { p.in[i] = i < nterms ? (const elem_t*)terms[i] : nullptr; p.shift[i] = i < nterms ? shifts[i] : 0; }
    for (int i = 0; i < nterms; ++i)
        DANET_CHECK_ARG(p.in[i] && p.shift[i] >= 0 && (H % (1 << p.shift[i])) == 0 && (W % (1 << p.shift[i])) == 0, "sum_relu_forward: term %d", i);
    const long nvec = (long)B * H * W * (C / VW);
    long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, (elem_t*)y, relu);
    DANET_CHECK_LAUNCH("sum_relu_kernel");
    return DANET_OK;
}

// All requested shifts in one launch: d[s] (s = 0..3, NULL = not needed) receives the [B, H >> s, W >> s, C] gradient of the
// terms that entered the sum up-sampled by 2^s.
extern "C" int NA_NAME(danet_sum_relu_backward_all)(const void* gy, const void* y, int B, int H, int W, int C, int relu,
                                           void* d0, void* d1, void* d2, void* d3, void* stream)
{
    DANET_ENTER();
    void* d[4] = {d0, d1, d2, d3};
    int smax = -1;
    for (int s_ = 0; s_ < 4; ++s_) if (d[s_]) smax = s_;
    DANET_CHECK_ARG(gy && (!relu || y) && C % VW == 0 && smax >= 0 && H % (1 << smax) == 0 && W % (1 << smax) == 0,
                    "sum_relu_backward_all: bad arguments");
    SumBwdAll out;
    for (int s_ = 0; s_ < 4; ++s_) out.d[s_] = (elem_t*)d[s_];
    const long nvec = (long)B * (H >> smax) * (W >> smax) * (C / VW);
    long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_relu_bwd_all_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)gy,
                       (const elem_t*)y, B, H, W, C, smax, relu, out);
    DANET_CHECK_LAUNCH("sum_relu_bwd_all_kernel");
    return DANET_OK;
}

// Multi-problem forms (host job arrays, n <= 4): the fuse sums of one HighResolutionModule / their gradients in ONE launch each.
//  forward job  { const void* terms[4]; int shifts[4]; int nterms, B, H, W, C, relu; void* y; }
//  backward job { const void* gy; const void* y; int B, H, W, C, relu; void* d[4]; }      d[s] = gradient of the terms with shift s (NULL: none)
struct SumFwdJob { const void* terms[4]; int shifts[4]; int nterms, B, H, W, C, relu; void* y; };
struct SumBwdJob { const void* gy; const void* y; int B, H, W, C, relu; void* d[4]; };

extern "C" int NA_NAME(danet_sum_relu_forward_multi)(const void* jobs_, int n, void* stream)
{
    DANET_ENTER();
    const SumFwdJob* jobs = (const SumFwdJob*)jobs_;
    DANET_CHECK_ARG(jobs && n >= 1 && n <= NSM, "sum_relu_forward_multi: 1..%d jobs", NSM);
    SumMulti m; m.n = n; m.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        const SumFwdJob& j = jobs[i];
        DANET_CHECK_ARG(j.y && j.nterms >= 1 && j.nterms <= 4 && j.C % VW == 0 && j.B > 0, "sum_relu_forward_multi: job %d: bad arguments", i);
        SumP& p = m.p[i];
        p.nterms = j.nterms; p.B = j.B; p.H = j.H; p.W = j.W; p.C = j.C;
        for (int t = 0; t < 4; ++t) { p.in[t] = t < j.nterms ? (const elem_t*)j.terms[t] : nullptr; p.shift[t] = t < j.nterms ? j.shifts[t] : 0; }
        for (int t = 0; t < j.nterms; ++t)
            DANET_CHECK_ARG(p.in[t] && p.shift[t] >= 0 && (j.H % (1 << p.shift[t])) == 0 && (j.W % (1 << p.shift[t])) == 0, "sum_relu_forward_multi: job %d term %d", i, t);
        m.y[i] = (elem_t*)j.y; m.relu[i] = j.relu;
        const long nvec = (long)j.B * j.H * j.W * (j.C / VW);
        long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
        m.start[i + 1] = m.start[i] + (int)blocks;
    }
    hipLaunchKernelGGL(sum_relu_multi_kernel, dim3((unsigned)m.start[n]), dim3(256), 0, (hipStream_t)stream, m);
    DANET_CHECK_LAUNCH("sum_relu_multi_kernel");
    return DANET_OK;
}

extern "C" int NA_NAME(danet_sum_relu_backward_all_multi)(const void* jobs_, int n, void* stream)
{
    DANET_ENTER();
    const SumBwdJob* jobs = (const SumBwdJob*)jobs_;
    DANET_CHECK_ARG(jobs && n >= 1 && n <= NSM, "sum_relu_backward_all_multi: 1..%d jobs", NSM);
    SumBwdMulti m; m.n = n; m.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        const SumBwdJob& j = jobs[i];
        int smax = -1;
        for (int s_ = 0; s_ < 4; ++s_) if (j.d[s_]) smax = s_;
        DANET_CHECK_ARG(j.gy && (!j.relu || j.y) && j.C % VW == 0 && smax >= 0 && j.H % (1 << smax) == 0 && j.W % (1 << smax) == 0,
                        "sum_relu_backward_all_multi: job %d: bad arguments", i);
        SumBwdMultiOne& a = m.a[i];
        a.gy = (const elem_t*)j.gy; a.y = (const elem_t*)j.y; a.B = j.B; a.H = j.H; a.W = j.W; a.C = j.C; a.smax = smax; a.relu = j.relu;
        for (int s_ = 0; s_ < 4; ++s_) a.out.d[s_] = (elem_t*)j.d[s_];
        const long nvec = (long)j.B * (j.H >> smax) * (j.W >> smax) * (j.C / VW);
        long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
        m.start[i + 1] = m.start[i] + (int)blocks;
    }
    hipLaunchKernelGGL(sum_relu_bwd_all_multi_kernel, dim3((unsigned)m.start[n]), dim3(256), 0, (hipStream_t)stream, m);
    DANET_CHECK_LAUNCH("sum_relu_bwd_all_multi_kernel");
    return DANET_OK;
}

extern "C" int NA_NAME(danet_sum_relu_backward)(const void* gy, const void* y, int B, int H, int W, int C, int shift, int relu,
                                       void* d_term, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(gy && d_term && (!relu || y) && C % VW == 0 && shift >= 0, "sum_relu_backward: bad arguments");
    const long nvec = (long)B * (H >> shift) * (W >> shift) * (C / VW);
    long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)gy,
                       (const elem_t*)y, B, H, W, C, shift, relu, (elem_t*)d_term);
    DANET_CHECK_LAUNCH("sum_relu_bwd_kernel");
    return DANET_OK;
}


// ---------------------------------------------------------------------------------------------
// Multi-tensor BatchNorm (training mode): jobs on the host, up to 4 per call, C <= 1024 each.
//  forward job:  { x, res, y, gamma, beta, running_mean, running_var, saved, sums; int64 M; int C, sums_state, relu }
//      sums_state: 1 = sums is zeroed scratch (statistics pass needed), 2 = statistics already accumulated by the conv epilogue
//      mask: NULL or M*C/4 bytes that receive the ReLU gate of y (bit j of byte i = channel 4i+j positive)
//  backward job: { dy, x, y, gamma, saved, dx, dres, dparam, red, beta, mask; int64 M; int C, red_state, relu, mask_mode }
//      (red_state as above; mask_mode: see ldmask above)
struct BnFwdJob { const void* x; const void* res; void* y; const float* gamma; const float* beta; float* running_mean; float* running_var;
                  float* saved; float* sums; void* mask; int64_t M; int C, sums_state, relu; };
struct BnBwdJob { const void* dy; const void* x; const void* y; const float* gamma; const float* saved; void* dx; void* dres; float* dparam;
                  float* red; const float* beta; const void* mask; int64_t M; int C, red_state, relu, mask_mode; };

extern "C" int NA_NAME(danet_bn_forward_multi)(const void* jobs_, int n, float momentum, float eps, void* stream)
{
    DANET_ENTER();
    const BnFwdJob* jobs = (const BnFwdJob*)jobs_;
    DANET_CHECK_ARG(jobs && n >= 1 && n <= NBM, "bn_forward_multi: 1..%d jobs", NBM);
    hipStream_t st = (hipStream_t)stream;
    BnFwdMulti m; m.n = n; m.momentum = momentum; m.eps = eps; m.mode = 0; m.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        const BnFwdJob& j = jobs[i];
        DANET_CHECK_ARG(j.x && j.y && j.saved && j.sums && j.M > 0 && j.C > 0 && j.C <= SLAB && (j.sums_state == 1 || j.sums_state == 2),
                        "bn_forward_multi: job %d: bad arguments (C <= %d, zeroed or pre-accumulated sums required)", i, SLAB);
#ifdef NA_F32
        DANET_CHECK_ARG(j.sums_state == 1, "bn_forward_multi_f32: job %d: the fp32 statistics are shifted sums (bn_stats_kernel): no pre-accumulated sums", i);
#endif
        BnFwdOne& a = m.a[i];
        int grid;
        DANET_CHECK_ARG(make_map(j.M, j.C, 0, j.C, &a.fm, &grid) == 0, "bn_forward_multi: job %d: C=%d unsupported", i, j.C);
        if (j.sums_state == 1) {
            hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(256), 0, st, (const elem_t*)j.x, a.fm, (bn_acc_t*)j.sums, j.C);
            DANET_CHECK_LAUNCH("bn_stats_kernel");
        }
        a.x = (const elem_t*)j.x; a.res = (const elem_t*)j.res; a.y = (elem_t*)j.y; a.sums = (const bn_acc_t*)j.sums; a.gamma = j.gamma; a.beta = j.beta;
        a.running_mean = j.running_mean; a.running_var = j.running_var; a.saved = j.saved; a.C = j.C; a.mask = (unsigned char*)j.mask;
        a.inv_count = 1.0f / (float)j.M; a.unbias = j.M > 1 ? (float)j.M / (float)(j.M - 1) : 1.f; a.relu = j.relu;
        m.start[i + 1] = m.start[i] + grid;
    }
    hipLaunchKernelGGL(bn_apply_multi_kernel, dim3(m.start[n]), dim3(256), 0, st, m);
    DANET_CHECK_LAUNCH("bn_apply_multi_kernel");
    return DANET_OK;
}

extern "C" int NA_NAME(danet_bn_backward_multi)(const void* jobs_, int n, void* stream)
{
    DANET_ENTER();
    const BnBwdJob* jobs = (const BnBwdJob*)jobs_;
    DANET_CHECK_ARG(jobs && n >= 1 && n <= NBM, "bn_backward_multi: 1..%d jobs", NBM);
    hipStream_t st = (hipStream_t)stream;
    BnBwdMulti m; m.n = n; m.start[0] = 0;
    bool need_reduce = false;
    for (int i = 0; i < n; ++i) {
        const BnBwdJob& j = jobs[i];
        DANET_CHECK_ARG(j.dy && j.x && j.dx && j.saved && j.red && j.M > 0 && j.C > 0 && j.C <= SLAB &&
                        (j.red_state == 1 || j.red_state == 2), "bn_backward_multi: job %d: bad arguments", i);
        DANET_CHECK_ARG(!j.relu || (j.mask_mode == 0 && j.y) || (j.mask_mode == 1 && j.mask) || (j.mask_mode == 2 && !j.dres),
                        "bn_backward_multi: job %d: ReLU gate (mask_mode %d) lacks its input", i, j.mask_mode);
        BnBwdOne& a = m.a[i];
        int grid;
        DANET_CHECK_ARG(make_map(j.M, j.C, 0, j.C, &a.fm, &grid) == 0, "bn_backward_multi: job %d: C=%d unsupported", i, j.C);
        a.dy = (const elem_t*)j.dy; a.x = (const elem_t*)j.x; a.y = (const elem_t*)j.y; a.saved = j.saved; a.gamma = j.gamma; a.red = (bn_acc_t*)j.red;
        a.dx = (elem_t*)j.dx; a.dres = (elem_t*)j.dres; a.dparam = j.dparam; a.C = j.C; a.inv_count = 1.0f / (float)j.M; a.relu = j.relu;
        a.have_red = j.red_state == 2; a.beta = j.beta; a.mask = (const unsigned char*)j.mask; a.mask_mode = j.mask_mode;
        need_reduce = need_reduce || !a.have_red;
        m.start[i + 1] = m.start[i] + grid;
    }
    if (need_reduce) {
        hipLaunchKernelGGL(bn_bwd_reduce_multi_kernel, dim3(m.start[n]), dim3(256), 0, st, m);
        DANET_CHECK_LAUNCH("bn_bwd_reduce_multi_kernel");
    }
    hipLaunchKernelGGL(bn_bwd_apply_multi_kernel, dim3(m.start[n]), dim3(256), 0, st, m);
    DANET_CHECK_LAUNCH("bn_bwd_apply_multi_kernel");
    return DANET_OK;
}


#ifndef NA_F32
// ---------------------------------------------------------------------------------------------
// One-pass form of danet_bn_backward_multi (bn_bwd_onepass_kernel).  `bar`: danet_bn_backward_onepass_bar_words() uints of device memory, zeroed ONCE by
// the caller and then owned by these launches (barrier state + error flag), shared by all launches -- which must not
// overlap (one stream).  danet_bn_backward_onepass_ok says whether a job set qualifies: every job needs its own reduction
// (red_state 1: zeroed scratch), a ReLU gate that does not need y (mask_mode 1 or 2, or no ReLU), C <= 1024, and the
// every job must fit the register + LDS budget (rows per lane <= 26 with <= 512 workgroups); jobs are packed into as
// few launches as that allows.
// Packs the jobs, in order, into launches of at most OP_MAX_BLOCKS workgroups; returns the number of launches (0: the
// set does not qualify).
// Workgroups that are certainly resident together: two per CU of the current device (256 CUs on an MI355X in SPX mode; a
// partitioned device exposes fewer, and a barrier over more workgroups than fit would only end by its spin bound).
static int onepass_max_blocks() {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 1;
        cached = 2 * cus < OP_MAX_BLOCKS ? 2 * cus : OP_MAX_BLOCKS;
    }
    return cached;
}

// cap (> 0): the caller's co-residency budget -- fewer workgroups than the device could hold when other kernels may occupy
// compute units while these launches run (a data-parallel trainer reserves the communication library's channels, see
// danet_hip.h); <= 0: the whole device.
static int onepass_plan_rows(const BnBwdJob* jobs, int n, BnOnePass* ms, int cap, int rows_per_lane);
// Rows per lane: 25 (OP_ROWS, bn_bwd_onepass_kernel<OP_NV>) unless the 29-row instantiation packs the set into FEWER launches: on the whole
// device the four-branch level is 489 workgroups at 25 rows -- one launch --; beside a data-parallel trainer's 24 communication channels
// the budget is 464 workgroups, which 25 rows would split in two and 29 rows fit in one (422 workgroups; round 6: -0.26 ms/step on the
// N > 1 path, the single-process step untouched).
static int onepass_plan(const BnBwdJob* jobs, int n, BnOnePass* ms /* [NBM] */, int cap) {
    BnOnePass alt[NBM];
    const int nl_small = onepass_plan_rows(jobs, n, alt, cap, OP_ROWS);
    const int nl_big = onepass_plan_rows(jobs, n, ms, cap, OP_ROWS_BIG);
    if (nl_small > 0 && (nl_big == 0 || nl_small <= nl_big)) {
        for (int l = 0; l < nl_small; ++l) { ms[l] = alt[l]; ms[l].big = 0; }
        return nl_small;
    }
    for (int l = 0; l < nl_big; ++l) ms[l].big = 1;
    return nl_big;
}
static int onepass_plan_rows(const BnBwdJob* jobs, int n, BnOnePass* ms /* [NBM] */, int cap, int rows_per_lane) {
    if (!jobs || n < 1 || n > NBM || getenv("DANET_NO_BN_ONEPASS")) return 0;
    int max_blocks = onepass_max_blocks();
    if (cap > 0 && cap < max_blocks) max_blocks = cap;
    int nl = 0;
    long total = 0;
    BnOnePass* m = nullptr;
    for (int i = 0; i < n; ++i) {
        const BnBwdJob& j = jobs[i];
        if (!(j.dy && j.x && j.dx && j.saved && j.red && j.M > 0 && j.C > 0 && (j.C <= SLAB || wide_slabs(j.C) > 1) && j.red_state == 1)) return 0;
        if (j.relu && !((j.mask_mode == 1 && j.mask) || (j.mask_mode == 2 && !j.dres))) return 0;
      // a wide tensor: one entry per channel slab (same rows, per-channel buffers offset, the full width as their stride)
      for (int s = 0, ns = wide_slabs(j.C); s < ns; ++s) {
        const int c0 = s * SLAB, Cs = j.C < SLAB ? j.C : SLAB;
        BnBwdOne a;
        int grid;
        if (make_map(j.M, j.C, c0, Cs, &a.fm, &grid) != 0) return 0;
        // rows per lane <= rows_per_lane: blocks >= rows / (rows per block step * rows_per_lane)
        const long rows_per_block = a.fm.span / a.fm.CV;
        long blocks = (j.M + rows_per_block * rows_per_lane - 1) / (rows_per_block * rows_per_lane);
        if (blocks < 1) blocks = 1;
        if (blocks > max_blocks) return 0;
        a.fm.rstep = (int)(rows_per_block * blocks);
        a.dy = (const elem_t*)j.dy; a.x = (const elem_t*)j.x; a.y = (const elem_t*)j.y; a.saved = j.saved + c0; a.gamma = j.gamma ? j.gamma + c0 : nullptr;
        a.red = (bn_acc_t*)j.red + c0;
        a.dx = (elem_t*)j.dx; a.dres = (elem_t*)j.dres; a.dparam = j.dparam ? j.dparam + c0 : nullptr; a.C = j.C; a.inv_count = 1.0f / (float)j.M; a.relu = j.relu;
        a.have_red = 0; a.beta = j.beta ? j.beta + c0 : nullptr; a.mask = (const unsigned char*)j.mask; a.mask_mode = j.mask_mode;
        if (!m || m->n == NBM || m->start[m->n] + blocks > max_blocks) {
            if (nl == NBM) return 0;
            m = &ms[nl++]; m->n = 0; m->start[0] = 0;
        }
        m->a[m->n] = a;
        m->start[m->n + 1] = m->start[m->n] + (int)blocks;
        ++m->n;
        total += blocks;
      }
    }
    // small sets are launch-bound either way: in isolation the barrier (~9 us; 17 + 0.055 us per workgroup in all) costs
    // more than a second ~6 us launch, but inside the captured step a threshold did not pay (33.38 / 33.39 / 33.52 / 33.70
    // ms per step for 0 / 100 / 200 / 300 workgroups), so every qualifying set takes the one-pass kernel
    static const long min_blocks = getenv("DANET_BN_ONEPASS_MIN") ? atol(getenv("DANET_BN_ONEPASS_MIN")) : 0;
    return total >= min_blocks ? nl : 0;
}

extern "C" int danet_bn_backward_onepass_bar_words(void) { return OP_BAR_WORDS; }

extern "C" int danet_bn_backward_onepass_ok(const void* jobs, int n, int max_blocks)
{
    BnOnePass ms[NBM];
    return onepass_plan((const BnBwdJob*)jobs, n, ms, max_blocks) > 0 ? 1 : 0;
}

extern "C" int danet_bn_backward_onepass(const void* jobs, int n, void* bar, int max_blocks, void* stream)
{
    DANET_ENTER();
    BnOnePass ms[NBM];
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_onepass_kernel<OP_NV>), hipFuncAttributeMaxDynamicSharedMemorySize, OP_NL * 256 * 17);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_onepass_kernel<OP_NV_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, OP_NL * 256 * 17);
        attr_set = true;
    }
    const int nl = onepass_plan((const BnBwdJob*)jobs, n, ms, max_blocks);
    DANET_CHECK_ARG(bar && nl > 0, "bn_backward_onepass: the job set does not qualify (see danet_bn_backward_onepass_ok)");
    for (int l = 0; l < nl; ++l) {
        ms[l].bar = (unsigned*)bar;
        ms[l].dbg = getenv("DANET_BN_ONEPASS_DBG") ? atoi(getenv("DANET_BN_ONEPASS_DBG")) : 0;
        if (ms[l].big) hipLaunchKernelGGL(bn_bwd_onepass_kernel<OP_NV_BIG>, dim3(ms[l].start[ms[l].n]), dim3(256), OP_NL * 256 * 17, (hipStream_t)stream, ms[l]);
        else hipLaunchKernelGGL(bn_bwd_onepass_kernel<OP_NV>, dim3(ms[l].start[ms[l].n]), dim3(256), OP_NL * 256 * 17, (hipStream_t)stream, ms[l]);
        DANET_CHECK_LAUNCH("bn_bwd_onepass_kernel");
    }
    return DANET_OK;
}
#endif  // NA_F32
