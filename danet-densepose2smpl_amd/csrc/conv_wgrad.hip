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
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

constexpr int CHUNK = 32;              // pixels per MFMA k-step
constexpr int LDP = CHUNK + 8;         // padded pixel row (80 B: conflict-free ds_read_b128)
constexpr int MAX_TG = 9;              // taps per block

struct WgradP {
    const bf16_t* x; const bf16_t* dy; float* dwp;
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, dil, groups;
    int Cin_g, Cout_g;
    int ntapgroups, msplit;
    long M;
};

template <int CT, int NI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p)
{
    constexpr int BCO = CT * 16, BCI = NI * 16;
    __shared__ __attribute__((aligned(16))) bf16_t sA[BCO][LDP];
    __shared__ __attribute__((aligned(16))) bf16_t sB[MAX_TG][BCI][LDP];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // grid.x = msplit, grid.y = cout-blocks * cin-blocks * tapgroups, grid.z = groups
    const int nco = (p.Cout_g + BCO - 1) / BCO, nci = (p.Cin_g + BCI - 1) / BCI;
    int by = blockIdx.y;
    const int tg = by % p.ntapgroups; by /= p.ntapgroups;
    const int cib = by % nci, cob = by / nci;
    const int g = blockIdx.z;
    const int taps = p.R * p.S;
    const int tap0 = tg * MAX_TG;
    const int ntap = min(MAX_TG, taps - tap0);
    const int co0 = cob * BCO, ci0 = cib * BCI;

    // this wave's accumulator tiles: indices tile = wave + 4*q over (tap, ct, ni)
    constexpr int MAXQ = (MAX_TG * CT * NI + 3) / 4;
    f32x4 acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = ntap * CT * NI;

    const long nchunks = (p.M + CHUNK - 1) / CHUNK;
    const long per = (nchunks + p.msplit - 1) / p.msplit;
    const long c_begin = (long)blockIdx.x * per, c_end = min(nchunks, c_begin + per);
    const int ohw = p.OH * p.OW;

    for (long ch = c_begin; ch < c_end; ++ch) {
        const long mbase = ch * CHUNK;
        __syncthreads();
        // stage dY^T: items = (8-channel run, pixel), pixel fastest
        for (int it = t; it < (BCO / 8) * CHUNK; it += 256) {
            const int pix = it % CHUNK, c8 = it / CHUNK;
            const long m = mbase + pix;
            const int c = co0 + c8 * 8;
            uint4 raw = {0u, 0u, 0u, 0u};
            if (m < p.M && c < p.Cout_g) {
                const bf16_t* src = p.dy + (size_t)m * p.Cout + (size_t)g * p.Cout_g + c;
                if (c + 8 <= p.Cout_g && (p.Cout % 8 == 0) && (p.Cout_g % 8 == 0)) raw = *reinterpret_cast<const uint4*>(src);
                else {
                    unsigned short v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = c + j < p.Cout_g ? src[j] : (unsigned short)0;
                    raw.x = v[0] | ((unsigned)v[1] << 16); raw.y = v[2] | ((unsigned)v[3] << 16);
                    raw.z = v[4] | ((unsigned)v[5] << 16); raw.w = v[6] | ((unsigned)v[7] << 16);
                }
            }
            const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) sA[c8 * 8 + j][pix] = (bf16_t)((w4[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        }
        // stage X^T for every tap of the group
        for (int it = t; it < ntap * (BCI / 8) * CHUNK; it += 256) {
            const int pix = it % CHUNK;
            const int rest = it / CHUNK;
            const int c8 = rest % (BCI / 8), tl = rest / (BCI / 8);
            const int tap = tap0 + tl;
            const int r = tap / p.S, s = tap - r * p.S;
            const long m = mbase + pix;
            const int c = ci0 + c8 * 8;
            uint4 raw = {0u, 0u, 0u, 0u};
            if (m < p.M && c < p.Cin_g) {
                const int b = (int)(m / ohw), rem = (int)(m - (long)b * ohw);
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih = oh * p.stride - p.pad + r * p.dil, iw = ow * p.stride - p.pad + s * p.dil;
                if (ih >= 0 && iw >= 0 && ih < p.H && iw < p.W) {
                    const bf16_t* src = p.x + (((size_t)b * p.H + ih) * p.W + iw) * p.Cin + (size_t)g * p.Cin_g + c;
                    if (c + 8 <= p.Cin_g && (p.Cin % 8 == 0) && (p.Cin_g % 8 == 0)) raw = *reinterpret_cast<const uint4*>(src);
                    else {
                        unsigned short v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = c + j < p.Cin_g ? src[j] : (unsigned short)0;
                        raw.x = v[0] | ((unsigned)v[1] << 16); raw.y = v[2] | ((unsigned)v[3] << 16);
                        raw.z = v[4] | ((unsigned)v[5] << 16); raw.w = v[6] | ((unsigned)v[7] << 16);
                    }
                }
            }
            const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) sB[tl][c8 * 8 + j][pix] = (bf16_t)((w4[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int tile = wave + 4 * q;
            if (tile < ntiles) {
                const int tl = tile / (CT * NI), rem = tile - tl * (CT * NI);
                const int ct = rem / NI, ni = rem - ct * NI;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sA[ct * 16 + li][lg * 8]);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sB[tl][ni * 16 + li][lg * 8]);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
            }
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
                atomicAdd(p.dwp + (((size_t)g * taps + tap) * p.Cout_g + cout) * p.Cin_g + cin, acc[q][r]);
        }
    }
}

// dWp[G][taps][Cout_g][Cin_g] -> dW[Cout][Cin_g][R][S]  (beta = 0: overwrite, 1: accumulate)
__global__ void wgrad_unpack_kernel(const float* __restrict__ dwp, float* __restrict__ dw,
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
    const float v = dwp[(((size_t)g * taps + tap) * Cout_g + cout) * Cin_g + cin];
    dw[idx] = beta != 0.f ? dw[idx] * beta + v : v;
}

template <int CT, int NI>
void launch_wgrad(const WgradP& p, hipStream_t st) {
    const int nco = (p.Cout_g + CT * 16 - 1) / (CT * 16), nci = (p.Cin_g + NI * 16 - 1) / (NI * 16);
    const dim3 grid((unsigned)p.msplit, (unsigned)(nco * nci * p.ntapgroups), (unsigned)p.groups);
    hipLaunchKernelGGL((conv_wgrad_kernel<CT, NI>), grid, dim3(256), 0, st, p);
}

inline int tiles_for(int c) { return c <= 16 ? 1 : (c <= 32 ? 2 : ((c % 48 == 0 || c <= 48) ? 3 : 4)); }

}  // namespace

// Template instance of conv_wgrad_kernel<CT, NI>: CT*10 + NI.
extern "C" int danet_conv_wgrad_kernel_id(int Cin, int Cout, int groups) {
    int ct = tiles_for(Cout / groups), ni = tiles_for(Cin / groups);
    if (ct * ni > 9) { if (ct == 4) ct = 2; if (ni == 4 && ct * ni > 9) ni = 2; }
    return ct * 10 + ni;
}

extern "C" size_t danet_conv_wgrad_ws_floats(int Cout, int Cin_g, int R, int S) {
    return (size_t)Cout * Cin_g * R * S;
}

// dW[Cout][Cin_g][R][S] (fp32, torch layout) = beta * dW + conv_wgrad(x, dy).
// ws: danet_conv_wgrad_ws_floats() floats of scratch (zeroed and filled here).
extern "C" int danet_conv_wgrad(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                                int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                int R, int S, int stride, int pad, int dil, int groups, float beta, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && dy && dw && ws, "conv_wgrad: null pointer");
    DANET_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 &&
                    groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv_wgrad: bad sizes");
    WgradP p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dwp = ws;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.dil = dil; p.groups = groups;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups;
    p.M = (long)B * OH * OW;
    const size_t need = danet_conv_wgrad_ws_floats(Cout, p.Cin_g, R, S);
    if (ws_floats < need) return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad: workspace %zu < %zu floats", ws_floats, need);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws, 0, need * sizeof(float), st);
    if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "conv_wgrad: memset: %s", hipGetErrorString(e));
    const int taps = R * S;
    p.ntapgroups = (taps + MAX_TG - 1) / MAX_TG;
    const int kid = danet_conv_wgrad_kernel_id(Cin, Cout, groups);                      // bounds registers
    const int ct = kid / 10, ni = kid % 10;
    const int nco = (p.Cout_g + ct * 16 - 1) / (ct * 16), nci = (p.Cin_g + ni * 16 - 1) / (ni * 16);
    const long other = (long)nco * nci * p.ntapgroups * groups;
    const long nchunks = (p.M + CHUNK - 1) / CHUNK;
    long msplit = (512 + other - 1) / other;
    if (msplit > nchunks / 2) msplit = nchunks / 2;
    if (msplit < 1) msplit = 1;
    p.msplit = (int)msplit;
#define WG_CASE(a, b) if (ct == a && ni == b) launch_wgrad<a, b>(p, st); else
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3)
    WG_CASE(4, 1) WG_CASE(4, 2)
    return danet::fail(DANET_ERR_ARG, "conv_wgrad: no kernel for tiles %dx%d", ct, ni);
#undef WG_CASE
    DANET_CHECK_LAUNCH("conv_wgrad_kernel");
    const long total = (long)Cout * p.Cin_g * taps;
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, st, ws, dw, groups, p.Cout_g,
                       p.Cin_g, taps, beta);
    DANET_CHECK_LAUNCH("wgrad_unpack_kernel");
    return DANET_OK;
}
