// Weight gradient of a convolution on the gfx950 matrix cores.
//
//   dW[cout][tap][cin] = sum_m dY[m][cout] * X[m @ tap][cin]         m = (b, oh, ow)
//
// The reduction axis is the PIXEL index, but both tensors are NHWC (channels contiguous), while
// an MFMA fragment wants 8 consecutive reduction elements per lane.  Each block therefore stages
// 32-pixel chunks of dY and of the tap-shifted X through LDS TRANSPOSED ([channel][pixel]), and
// reads fragments back with ds_read_b128.  Accumulators for all taps of the block's
// (cout-block, cin-block) stay in registers across the block's whole pixel range; the partial
// results are added atomically into a packed fp32 buffer dWp[G][taps][Cout_g][Cin_g] (cin
// contiguous -> 64-byte atomic segments), which conv_wgrad_unpack turns into the torch layout.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

constexpr int CHUNK = 32;              // pixels per MFMA k-step
constexpr int LDP = CHUNK + 8;         // padded pixel row (80 B: conflict-free ds_read_b128)

// The packed accumulator dWp holds bn_acc_t values (doubles; conv_common.h): every workgroup adds its fp32 partial products with one
// device-scope atomic per element, and in double precision those additions are exact (partials of an element within 2^(29 - log2
// msplit) of each other), hence independent of the order in which the pixel chunks' workgroups arrive -- the weight gradients of
// two executions of the same step agree bit for bit.
typedef bn_acc_t wg_acc_t;
constexpr int WG_ACC_FLOATS = BN_ACC_FLOATS;
struct WgradP {
    const bf16_t* x; const bf16_t* dy; wg_acc_t* dwp; const float* zero;
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, dil, groups;
    int Cin_g, Cout_g;
    int ntapgroups, msplit;
    long M;
};

// raw[i] = 8 channels of pixel i (i = 0..3).  Writes row j (channel j) = {pix0, pix1, pix2, pix3} as one 8-byte
// store to dst + j*LDP (dst points at [channel 0][pixel 0] of the 4-pixel group).
__device__ inline void store_transposed(bf16_t* dst, const uint4* raw) {
    const unsigned w[4][4] = {{raw[0].x, raw[0].y, raw[0].z, raw[0].w}, {raw[1].x, raw[1].y, raw[1].z, raw[1].w},
                              {raw[2].x, raw[2].y, raw[2].z, raw[2].w}, {raw[3].x, raw[3].y, raw[3].z, raw[3].w}};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {           // channel pair (2jj, 2jj+1)
        uint2 lo, hi;
        lo.x = __builtin_amdgcn_perm(w[1][jj], w[0][jj], 0x05040100u);   // ch 2jj   : pix0 | pix1 << 16
        lo.y = __builtin_amdgcn_perm(w[3][jj], w[2][jj], 0x05040100u);   //           pix2 | pix3 << 16
        hi.x = __builtin_amdgcn_perm(w[1][jj], w[0][jj], 0x07060302u);   // ch 2jj+1
        hi.y = __builtin_amdgcn_perm(w[3][jj], w[2][jj], 0x07060302u);
        *reinterpret_cast<uint2*>(dst + (2 * jj) * LDP) = lo;
        *reinterpret_cast<uint2*>(dst + (2 * jj + 1) * LDP) = hi;
    }
}

__device__ inline unsigned udiv24(unsigned n, unsigned d, float rcp) {      // n < 2^24 (checked by the host)
    unsigned q = (unsigned)((float)n * rcp);
    const int r = (int)(n - q * d);
    if (r < 0) --q; else if (r >= (int)d) ++q;
    return q;
}
__device__ inline int sTap_dh(const WgradP& p, int tap) { const int tt = min(tap, p.R * p.S - 1); return (tt / p.S) * p.dil; }
__device__ inline int sTap_dw(const WgradP& p, int tap) { const int tt = min(tap, p.R * p.S - 1); return (tt - (tt / p.S) * p.S) * p.dil; }

// Channel counts must be multiples of 8 (the host pads; see conv.py): every staged run is one aligned
// 16-byte load.  All loads are unconditional (clamped address + select) -- predicated loads compiled to
// ~400 branches per chunk and made the kernel instruction-bound.
template <int CT, int NI, int TG>
__device__ __forceinline__ void wgrad_body(const WgradP& p, const int bx, const int by_in, const int bz)
{
    constexpr int BCO = CT * 16, BCI = NI * 16;
    constexpr int ROWS = BCO + TG * BCI;                       // LDS rows: dY^T channels, then X^T per tap
    __shared__ __attribute__((aligned(16))) bf16_t sT[ROWS][LDP];
    __shared__ int sTap[TG][3];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // grid.x = msplit, grid.y = cout-blocks * cin-blocks * tapgroups, grid.z = groups
    const int nci = (p.Cin_g + BCI - 1) / BCI;
    int by = by_in;
    const int tg = by % p.ntapgroups; by /= p.ntapgroups;
    const int cib = by % nci, cob = by / nci;
    const int g = bz;
    const int taps = p.R * p.S;
    const int tap0 = tg * TG;
    const int ntap = min(TG, taps - tap0);
    const int co0 = cob * BCO, ci0 = cib * BCI;
    if (t < TG) {
        const int tap = min(tap0 + t, taps - 1);
        const int r = tap / p.S;
        sTap[t][0] = r * p.dil; sTap[t][1] = (tap - r * p.S) * p.dil;
        sTap[t][2] = (sTap[t][0] * p.W + sTap[t][1]) * p.Cin;
    }

    // accumulator tiles of this wave: tile = wave + 4*q over (tap, ct, ni); LDS fragment offsets are
    // chunk-independent and precomputed
    constexpr int NTILES = TG * CT * NI;
    constexpr int MAXQ = (NTILES + 3) / 4;
    f32x4 acc[MAXQ];
    int aoff[MAXQ], boff[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tile = min(wave + 4 * q, NTILES - 1);
        const int tl = tile / (CT * NI), rem = tile - tl * (CT * NI);
        const int ct = rem / NI, ni = rem - ct * NI;
        aoff[q] = (ct * 16 + li) * LDP + lg * 8;
        boff[q] = (BCO + tl * BCI + ni * 16 + li) * LDP + lg * 8;
    }
    const int ntiles = ntap * CT * NI;

    const long nchunks = (p.M + CHUNK - 1) / CHUNK;
    const long per = (nchunks + p.msplit - 1) / p.msplit;
    const long c_begin = (long)bx * per, c_end = min(nchunks, c_begin + per);
    const int ohw = p.OH * p.OW;

    // Items of one chunk: one item = 4 consecutive pixels x 8 channels: four 16-byte loads, a 4x8 transpose
    // in registers (v_perm_b32), eight 8-byte LDS stores.  X^T items (NIT_B of them, row block BCO/8 + itb/8)
    // are spread over NITEM_B rounds of the 256 lanes; the dY^T items (NIT_A <= 64) take one extra round on
    // wave 0.  A lane keeps the same pixel quad (t & 7) for all its items: the (b, oh, ow) decode and the
    // 32-bit element offsets of its four pixels are computed once per chunk, an item then costs two adds
    // and two unsigned compares per load.  Out-of-image / out-of-range runs read a 16-byte zero block behind
    // the workspace instead of branching.  The loads of chunk i+1 are issued before the MFMAs of chunk i and
    // committed to LDS after them (register double buffering).
    constexpr int NIT_A = (BCO / 8) * (CHUNK / 4), NIT_B = TG * (BCI / 8) * (CHUNK / 4);
    constexpr int NITEM_B = (NIT_B + 255) / 256;
    static_assert(NIT_A <= 256, "dY items must fit one round");
    const int pq = t & (CHUNK / 4 - 1);
    const int nit_b = ntap * (BCI / 8) * (CHUNK / 4);
    uint4 raw[NITEM_B + 1][4];
    // 32-bit byte offsets into buffer resources; out-of-image / out-of-range runs get an out-of-range offset and
    // load zeros (no address selects, no branches)
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy + (size_t)g * p.Cout_g), 0, (int)(p.M * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x + (size_t)g * p.Cin_g), 0, (int)((long)p.B * p.H * p.W * p.Cin * 2), 0x00020000);
    const float rc_ohw = 1.0f / (float)ohw, rc_ow = 1.0f / (float)p.OW;
    // per item: channel offset (bytes) and tap geometry do not depend on the chunk
    int it_dh[NITEM_B], it_dw[NITEM_B], it_off[NITEM_B];
#pragma unroll
    for (int u = 0; u < NITEM_B; ++u) {
        const int itb = t + u * 256;
        const int rb = itb / (CHUNK / 4);
        const int c8 = rb % (BCI / 8), tl = min(rb / (BCI / 8), TG - 1);
        const int cB = ci0 + c8 * 8;
        const bool chan_ok = itb < nit_b && cB < p.Cin_g;
        it_dh[u] = chan_ok ? sTap_dh(p, tap0 + tl) : -100000;                  // (-100000: never inside the image)
        it_dw[u] = sTap_dw(p, tap0 + tl);
        it_off[u] = (it_dh[u] * p.W + it_dw[u]) * p.Cin * 2 + cB * 2;
    }
    const int a_off = (co0 + (t / (CHUNK / 4)) * 8) * 2;
    const bool a_ok = t < NIT_A && co0 + (t / (CHUNK / 4)) * 8 < p.Cout_g;

    auto fetch = [&](long chunk) {
        const int mbase = (int)(chunk * CHUNK);
        int offA[4], offB[4], pih[4], piw[4];
        // decode the first pixel of the quad, step the other three
        int m = mbase + pq * 4;
        const int mc = m < (int)p.M ? m : (int)p.M - 1;
        int b = (int)udiv24((unsigned)mc, (unsigned)ohw, rc_ohw);
        const int rem = mc - b * ohw;
        int oh = (int)udiv24((unsigned)rem, (unsigned)p.OW, rc_ow), ow = rem - oh * p.OW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool pv = m + i < (int)p.M;
            pih[i] = pv ? oh * p.stride - p.pad : -100000;
            piw[i] = ow * p.stride - p.pad;
            offA[i] = pv ? (m + i) * p.Cout * 2 : OOB;
            offB[i] = ((b * p.H + oh * p.stride - p.pad) * p.W + piw[i]) * p.Cin * 2;      // byte offset of tap (0,0)
            if (++ow == p.OW) { ow = 0; if (++oh == p.OH) { oh = 0; ++b; } }
        }
#pragma unroll
        for (int u = 0; u < NITEM_B; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (unsigned)(pih[i] + it_dh[u]) < (unsigned)p.H && (unsigned)(piw[i] + it_dw[u]) < (unsigned)p.W;
                raw[u][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? offB[i] + it_off[u] : OOB, 0, 0));
            }
        }
        if (t < NIT_A) {                                   // dY^T items: wave 0 only
#pragma unroll
            for (int i = 0; i < 4; ++i)
                raw[NITEM_B][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yr, (a_ok && offA[i] != OOB) ? offA[i] + a_off : OOB, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NITEM_B; ++u) {
            const int itb = t + u * 256;
            if (itb < nit_b) store_transposed(&sT[BCO + (itb / (CHUNK / 4)) * 8][pq * 4], raw[u]);
        }
        if (t < NIT_A) store_transposed(&sT[(t / (CHUNK / 4)) * 8][pq * 4], raw[NITEM_B]);
    };

    __syncthreads();                       // tap table visible
    if (c_begin < c_end) fetch(c_begin);
    const bf16_t* const lds = &sT[0][0];
    for (long ch = c_begin; ch < c_end; ++ch) {
        __syncthreads();                   // every wave finished reading the previous chunk
        commit();
        __syncthreads();
        if (ch + 1 < c_end) fetch(ch + 1);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + aoff[q]);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(lds + boff[q]);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
        }
    }
    // D[i = cout][j = cin]: lane holds couts ct*16 + lg*4 + {0..3} for cin ni*16 + li
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int tile = wave + 4 * q;
        if (tile >= ntiles) continue;
        const int tl = tile / (CT * NI), rem = tile - tl * (CT * NI);
        const int ct = rem / NI, ni = rem - ct * NI;
        const int cin = ci0 + ni * 16 + li;
        if (cin >= p.Cin_g) continue;
        const int tap = tap0 + tl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = co0 + ct * 16 + lg * 4 + r;
            if (cout < p.Cout_g)
                __hip_atomic_fetch_add((__attribute__((address_space(1))) wg_acc_t*)(p.dwp + (((size_t)g * taps + tap) * p.Cout_g + cout) * p.Cin_g + cin),
                                       (wg_acc_t)acc[q][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int CT, int NI, int TG>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p)
{
    wgrad_body<CT, NI, TG>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent problems of one kernel instance in one launch (see conv_wgrad3x3.hip: weight gradients are
// queued during the backward pass and flushed in batches).
constexpr long PW_WGRAD_BLOCKS = 768;       // workgroups of a pointwise weight-gradient launch (three 45 KB workgroups per compute unit)
constexpr int NPM = 16;
struct WgradMulti { WgradP p[NPM]; int start[NPM + 1]; int nyb[NPM]; int n; };

template <int CT, int NI, int TG>
__global__ __launch_bounds__(256) void conv_wgrad_multi_kernel(WgradMulti mp)
{
    int i = 0;
    while (i + 1 < mp.n && (int)blockIdx.x >= mp.start[i + 1]) ++i;
    const int l = blockIdx.x - mp.start[i];
    const WgradP& p = mp.p[i];
    const int bx = l % p.msplit, rest = l / p.msplit;
    wgrad_body<CT, NI, TG>(p, bx, rest % mp.nyb[i], rest / mp.nyb[i]);
}

struct UnpackMulti { const wg_acc_t* dwp[NPM]; float* dw[NPM]; int G[NPM], Cout_g[NPM], Cin_g[NPM], taps[NPM]; long start[NPM + 1]; int n; float beta; };

__global__ __launch_bounds__(256) void wgrad_unpack_multi_kernel(UnpackMulti up)
{
    const long gidx = (long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= up.start[up.n]) return;
    int i = 0;
    while (i + 1 < up.n && gidx >= up.start[i + 1]) ++i;
    const long idx = gidx - up.start[i];
    const int taps = up.taps[i], Cin_g = up.Cin_g[i], Cout_g = up.Cout_g[i];
    const int tap = (int)(idx % taps);
    long rest = idx / taps;
    const int cin = (int)(rest % Cin_g); rest /= Cin_g;
    const int cout = (int)(rest % Cout_g);
    const int g = (int)(rest / Cout_g);
    const float v = (float)up.dwp[i][(((size_t)g * taps + tap) * Cout_g + cout) * Cin_g + cin];
    float* dw = up.dw[i];
    dw[idx] = up.beta != 0.f ? dw[idx] * up.beta + v : v;
}

// dWp[G][taps][Cout_g][Cin_g] -> dW[Cout][Cin_g][R][S]  (beta = 0: overwrite, 1: accumulate)
__global__ void wgrad_unpack_kernel(const wg_acc_t* __restrict__ dwp, float* __restrict__ dw,
                                    int G, int Cout_g, int Cin_g, int taps, float beta)
{
    const long total = (long)G * Cout_g * Cin_g * taps;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int tap = (int)(idx % taps);
    long rest = idx / taps;
    const int cin = (int)(rest % Cin_g); rest /= Cin_g;
    const int cout = (int)(rest % Cout_g);
    const int g = (int)(rest / Cout_g);
    const float v = (float)dwp[(((size_t)g * taps + tap) * Cout_g + cout) * Cin_g + cin];
    dw[idx] = beta != 0.f ? dw[idx] * beta + v : v;
}

template <int CT, int NI, int TG>
void launch_wgrad(const WgradP& p, hipStream_t st) {
    const int nco = (p.Cout_g + CT * 16 - 1) / (CT * 16), nci = (p.Cin_g + NI * 16 - 1) / (NI * 16);
    const dim3 grid((unsigned)p.msplit, (unsigned)(nco * nci * p.ntapgroups), (unsigned)p.groups);
    hipLaunchKernelGGL((conv_wgrad_kernel<CT, NI, TG>), grid, dim3(256), 0, st, p);
}

inline int tiles_for(int c) { return c <= 16 ? 1 : (c <= 32 ? 2 : ((c % 48 == 0 || c <= 48) ? 3 : 4)); }

}  // namespace

// Template instance conv_wgrad_kernel<CT, NI, TG>: returns CT*100 + NI*10 + (TG == 1 ? 1 : 9).
// 1x1 convolutions (one tap) keep only CT*NI/4 accumulator tiles per wave, so they take 64x64 blocks.
extern "C" int danet_conv_wgrad_kernel_id(int Cin, int Cout, int groups, int taps) {
    int ct = tiles_for(Cout / groups), ni = tiles_for(Cin / groups);
    if (taps > 1 && ct * ni > 9) { if (ct == 4) ct = 2; if (ni == 4 && ct * ni > 9) ni = 2; }
    return ct * 100 + ni * 10 + (taps == 1 ? 1 : 9);
}

extern "C" size_t danet_conv_wgrad_ws_floats(int Cout, int Cin_g, int R, int S) {
    return (size_t)Cout * Cin_g * R * S * WG_ACC_FLOATS + 16;            // accumulators (wg_acc_t) + a zero block the kernel reads for out-of-range runs
}

// Workspace of danet_conv_wgrad for a given problem: the packed accumulator of danet_conv_wgrad_ws_floats, or -- 1x1 / stride-1 layers
// on csrc/conv_pw_wgrad.hip -- that kernel's partial sums, whichever is larger.
extern "C" size_t danet_conv_wgrad_ws_floats_for(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups) {
    size_t need = groups > 0 ? danet_conv_wgrad_ws_floats(Cout, Cin / groups, R, S) : 0;
    const WgJob job{nullptr, nullptr, nullptr, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups};
    const int one = 0;
    if (conv_pw_wgrad_ok(job)) { const size_t n2 = conv_pw_wgrad_ws_floats(&job, &one, 1, PW_WGRAD_BLOCKS); if (n2 > need) need = n2; }
    return need;
}

// dW[Cout][Cin_g][R][S] (fp32, torch layout) = beta * dW + conv_wgrad(x, dy).
// ws: danet_conv_wgrad_ws_floats() floats of scratch (zeroed and filled here).
extern "C" int danet_conv_wgrad(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                                int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                int R, int S, int stride, int pad, int dil, int groups, float beta, int ws_is_zero, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && dy && dw && ws, "conv_wgrad: null pointer");
    DANET_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 &&
                    groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv_wgrad: bad sizes");
    DANET_CHECK_ARG((Cin / groups) % 8 == 0 && (Cout / groups) % 8 == 0 && (long)B * OH * OW * Cout < 2147483647L &&
                    (long)B * H * W * Cin < 1073741823L && (long)B * OH * OW * Cout < 1073741823L && (long)B * OH * OW < (1L << 24),
                    "conv_wgrad: channels per group must be multiples of 8 (Cin_g=%d, Cout_g=%d); the host pads them",
                    Cin / groups, Cout / groups);
    {
        const WgJob job{x, dy, dw, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups};
        const int one = 0;
        if (conv_pw_wgrad_ok(job) && conv_pw_wgrad_ws_floats(&job, &one, 1, PW_WGRAD_BLOCKS) <= ws_floats) {
            // (a workspace sized by danet_conv_wgrad_ws_floats_for: the pointwise kernel's partial sums; need not be zeroed)
            if (conv_pw_wgrad_launch(&job, &one, 1, ws, beta, PW_WGRAD_BLOCKS, stream) != 0) return danet::fail(DANET_ERR_HIP, "conv_wgrad: pointwise launch failed");
            return DANET_OK;
        }
    }
    WgradP p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dwp = (wg_acc_t*)ws;
    p.zero = ws + (danet_conv_wgrad_ws_floats(Cout, Cin / groups, R, S) - 16);
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.dil = dil; p.groups = groups;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups;
    p.M = (long)B * OH * OW;
    const size_t need = danet_conv_wgrad_ws_floats(Cout, p.Cin_g, R, S);
    if (ws_floats < need) return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad: workspace %zu < %zu floats", ws_floats, need);
    hipStream_t st = (hipStream_t)stream;
    if (!ws_is_zero) {
        hipError_t e = danet::zero_async(ws, need * sizeof(float), st);
        if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "conv_wgrad: memset: %s", hipGetErrorString(e));
    }
    const int taps = R * S;
    const int kid = danet_conv_wgrad_kernel_id(Cin, Cout, groups, taps);               // bounds registers
    const int ct = kid / 100, ni = (kid / 10) % 10, tgs = kid % 10;
    p.ntapgroups = (taps + tgs - 1) / tgs;
    const int nco = (p.Cout_g + ct * 16 - 1) / (ct * 16), nci = (p.Cin_g + ni * 16 - 1) / (ni * 16);
    const long other = (long)nco * nci * p.ntapgroups * groups;
    const long nchunks = (p.M + CHUNK - 1) / CHUNK;
    long target = 384;
    if (const char* e = getenv("DANET_WGRAD_BLOCKS")) target = atol(e);       // tuning knob
    long msplit = (target + other - 1) / other;
    if (msplit > nchunks / 2) msplit = nchunks / 2;
    if (msplit < 1) msplit = 1;
    p.msplit = (int)msplit;
#define WG_CASE(a, b) if (ct == a && ni == b && tgs == 9) launch_wgrad<a, b, 9>(p, st); else
#define WG_CASE1(a, b) if (ct == a && ni == b && tgs == 1) launch_wgrad<a, b, 1>(p, st); else
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3)
    WG_CASE(4, 1) WG_CASE(4, 2)
    WG_CASE1(1, 1) WG_CASE1(1, 2) WG_CASE1(1, 3) WG_CASE1(1, 4)
    WG_CASE1(2, 1) WG_CASE1(2, 2) WG_CASE1(2, 3) WG_CASE1(2, 4)
    WG_CASE1(3, 1) WG_CASE1(3, 2) WG_CASE1(3, 3) WG_CASE1(3, 4)
    WG_CASE1(4, 1) WG_CASE1(4, 2) WG_CASE1(4, 3) WG_CASE1(4, 4)
    return danet::fail(DANET_ERR_ARG, "conv_wgrad: no kernel for tiles %dx%d", ct, ni);
#undef WG_CASE
#undef WG_CASE1
    DANET_CHECK_LAUNCH("conv_wgrad_kernel");
    const long total = (long)Cout * p.Cin_g * taps;
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, st, (const wg_acc_t*)ws, dw, groups, p.Cout_g,
                       p.Cin_g, taps, beta);
    DANET_CHECK_LAUNCH("wgrad_unpack_kernel");
    return DANET_OK;
}


// ---------------------------------------------------------------------------------------------
// Batched form: n independent problems, grouped by kernel instance, up to 16 per launch.  ws: zeroed by the
// caller (danet_conv_wgrad_multi_ws_floats floats; every problem's packed accumulator lives there).
// (WgJob: conv_common.h)

static bool wg_job_ok(const WgJob& j) {
    return j.x && j.dy && j.dw && j.B > 0 && j.H > 0 && j.W > 0 && j.Cin > 0 && j.OH > 0 && j.OW > 0 && j.Cout > 0 && j.R > 0 && j.S > 0 &&
           j.stride > 0 && j.groups > 0 && j.Cin % j.groups == 0 && j.Cout % j.groups == 0 && (j.Cin / j.groups) % 8 == 0 &&
           (j.Cout / j.groups) % 8 == 0 && (long)j.B * j.H * j.W * j.Cin < 1073741823L && (long)j.B * j.OH * j.OW * j.Cout < 1073741823L &&
           (long)j.B * j.OH * j.OW < (1L << 24);
}

static int wg_multi(const WgJob* jobs, int n, float* ws, size_t ws_floats, float beta, hipStream_t st, size_t* need_out, size_t* zero_from_out = nullptr)
{
    if (n > 4096) return danet::fail(DANET_ERR_ARG, "conv_wgrad_multi: too many jobs (%d)", n);
    bool done[4096];
    for (int i = 0; i < n; ++i) done[i] = false;
    size_t used = 0;
    {
        // 1x1 / stride-1 problems: csrc/conv_pw_wgrad.hip (its partial sums need no zeroed workspace)
        thread_local static int idx[4096];
        int cnt = 0;
        for (int i = 0; i < n; ++i) if (conv_pw_wgrad_ok(jobs[i])) { idx[cnt++] = i; done[i] = true; }
        if (cnt > 0) {
            const size_t need = conv_pw_wgrad_ws_floats(jobs, idx, cnt, PW_WGRAD_BLOCKS);
            if (ws) {
                if (need > ws_floats) return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad_multi: workspace too small");
                if (conv_pw_wgrad_launch(jobs, idx, cnt, ws, beta, PW_WGRAD_BLOCKS, st) != 0) return danet::fail(DANET_ERR_HIP, "conv_wgrad_multi: pointwise launch failed");
            }
            used += need;
        }
    }
    if (zero_from_out) *zero_from_out = used;         // the pointwise kernel's partial sums come first and need no zeroing; the packed accumulators follow
    long target = 1024;     // swept on MI355X (512 / 768 / 1024 / 1536 / 2048: 33.04 / 32.82 / 32.67 / 32.81 / 32.89 ms per step once
                            // the 7x7 stem has its own kernel and one process flushes all problems at once)
    if (const char* e = getenv("DANET_WGRAD_MULTI_BLOCKS")) target = atol(e);
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        const int kid = danet_conv_wgrad_kernel_id(jobs[i].Cin, jobs[i].Cout, jobs[i].groups, jobs[i].R * jobs[i].S);
        int idx[NPM], cnt = 0;
        for (int k = i; k < n && cnt < NPM; ++k)
            if (!done[k] && danet_conv_wgrad_kernel_id(jobs[k].Cin, jobs[k].Cout, jobs[k].groups, jobs[k].R * jobs[k].S) == kid) { idx[cnt++] = k; done[k] = true; }
        const int ct = kid / 100, ni = (kid / 10) % 10, tgs = kid % 10;
        double tot = 0;
        auto cw = [](const WgJob& j) { const double p = (double)j.Cout * (j.Cin / j.groups); return p < 2304.0 ? 2304.0 : p; };      // (floor: see conv_wgrad3x3.hip wg3_weight)
        for (int k = 0; k < cnt; ++k) { const WgJob& j = jobs[idx[k]]; tot += (double)j.B * j.OH * j.OW * cw(j) * j.R * j.S; }
        WgradMulti mp; UnpackMulti up;
        mp.n = cnt; up.n = cnt; up.beta = beta; mp.start[0] = 0; up.start[0] = 0;
        for (int k = 0; k < cnt; ++k) {
            const WgJob& j = jobs[idx[k]];
            WgradP& p = mp.p[k];
            const int taps = j.R * j.S;
            p.x = (const bf16_t*)j.x; p.dy = (const bf16_t*)j.dy; p.dwp = ws ? (wg_acc_t*)(ws + used) : nullptr; p.zero = nullptr;
            p.B = j.B; p.H = j.H; p.W = j.W; p.Cin = j.Cin; p.OH = j.OH; p.OW = j.OW; p.Cout = j.Cout;
            p.R = j.R; p.S = j.S; p.stride = j.stride; p.pad = j.pad; p.dil = j.dil; p.groups = j.groups;
            p.Cin_g = j.Cin / j.groups; p.Cout_g = j.Cout / j.groups;
            p.M = (long)j.B * j.OH * j.OW;
            p.ntapgroups = (taps + tgs - 1) / tgs;
            const int nco = (p.Cout_g + ct * 16 - 1) / (ct * 16), nci = (p.Cin_g + ni * 16 - 1) / (ni * 16);
            const long other = (long)nco * nci * p.ntapgroups * j.groups;
            const long nchunks = (p.M + CHUNK - 1) / CHUNK;
            const double w = (double)p.M * cw(j) * taps;
            long ms = (long)(target * (w / tot) / other + 0.5);
            if (ms > nchunks / 2) ms = nchunks / 2;
            if (ms < 1) ms = 1;
            p.msplit = (int)ms;
            mp.nyb[k] = nco * nci * p.ntapgroups;
            mp.start[k + 1] = mp.start[k] + (int)(ms * other);
            const long total = (long)j.Cout * p.Cin_g * taps;
            up.dwp[k] = p.dwp; up.dw[k] = j.dw; up.G[k] = j.groups; up.Cout_g[k] = p.Cout_g; up.Cin_g[k] = p.Cin_g; up.taps[k] = taps;
            up.start[k + 1] = up.start[k] + total;
            used += (size_t)(total * WG_ACC_FLOATS + 15) / 16 * 16;
        }
        if (!ws) continue;
        if (used > ws_floats) return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad_multi: workspace too small");
        const dim3 grid((unsigned)mp.start[cnt]);
#define WGM(a, b) if (ct == a && ni == b && tgs == 9) hipLaunchKernelGGL((conv_wgrad_multi_kernel<a, b, 9>), grid, dim3(256), 0, st, mp); else
#define WGM1(a, b) if (ct == a && ni == b && tgs == 1) hipLaunchKernelGGL((conv_wgrad_multi_kernel<a, b, 1>), grid, dim3(256), 0, st, mp); else
        WGM(1, 1) WGM(1, 2) WGM(1, 3) WGM(1, 4) WGM(2, 1) WGM(2, 2) WGM(2, 3) WGM(2, 4) WGM(3, 1) WGM(3, 2) WGM(3, 3) WGM(4, 1) WGM(4, 2)
        WGM1(1, 1) WGM1(1, 2) WGM1(1, 3) WGM1(1, 4) WGM1(2, 1) WGM1(2, 2) WGM1(2, 3) WGM1(2, 4)
        WGM1(3, 1) WGM1(3, 2) WGM1(3, 3) WGM1(3, 4) WGM1(4, 1) WGM1(4, 2) WGM1(4, 3) WGM1(4, 4)
        return danet::fail(DANET_ERR_ARG, "conv_wgrad_multi: no kernel for tiles %dx%d", ct, ni);
#undef WGM
#undef WGM1
        DANET_CHECK_LAUNCH("conv_wgrad_multi_kernel");
        hipLaunchKernelGGL(wgrad_unpack_multi_kernel, dim3((unsigned)danet::cdiv(up.start[cnt], 256)), dim3(256), 0, st, up);
        DANET_CHECK_LAUNCH("wgrad_unpack_multi_kernel");
    }
    if (need_out) *need_out = used;
    return DANET_OK;
}

extern "C" size_t danet_conv_wgrad_multi_ws_floats(const void* jobs, int n)
{
    size_t need = 0;
    if (!jobs || n <= 0) return 0;
    wg_multi((const WgJob*)jobs, n, nullptr, 0, 0.f, nullptr, &need);
    return need;
}

// First float of the workspace that must be zero when danet_conv_wgrad_multi is called (everything from there to
// danet_conv_wgrad_multi_ws_floats): the leading part holds partial sums that are written before they are read.
extern "C" size_t danet_conv_wgrad_multi_ws_zero_from(const void* jobs, int n)
{
    size_t need = 0, from = 0;
    if (!jobs || n <= 0) return 0;
    wg_multi((const WgJob*)jobs, n, nullptr, 0, 0.f, nullptr, &need, &from);
    return from;
}

extern "C" int danet_conv_wgrad_multi(const void* jobs, int n, float* ws, size_t ws_floats, float beta, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(jobs && n > 0 && ws, "conv_wgrad_multi: bad arguments");
    const WgJob* jb = (const WgJob*)jobs;
    for (int i = 0; i < n; ++i) DANET_CHECK_ARG(wg_job_ok(jb[i]), "conv_wgrad_multi: job %d has unsupported sizes", i);
    return wg_multi(jb, n, ws, ws_floats, beta, (hipStream_t)stream, nullptr);
}
