// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) with BOTH MFMA operands served from LDS.
//
// These layers are 89.6 % of the backbone's MACs (every BasicBlock / Bottleneck conv of
// /root/reference/models/module/hr_module.py and res_module.py).  The generic implicit-GEMM kernel
// (conv_igemm.hip) gathers every B fragment from global memory -- each input pixel travels through the
// texture path nine times -- and every wave of a block re-reads the same weights, which makes it
// operand-bandwidth bound at ~10 % of the MFMA rate.  Here a block
//   * copies the (TH+2) x (TW+2) halo of its pixel tile into LDS once per channel chunk (plain coalesced
//     16-byte loads; a tap shift is then just an LDS address offset),
//   * streams the weights through LDS in slices of SK k-steps (loaded once per block, not once per wave),
//   * prefetches the next slice / chunk into registers while the MFMAs of the current one run.
// K order: (channel chunk, tap, channel within chunk), each chunk zero-padded to a multiple of 32 --
// the weights are packed accordingly (conv_igemm.hip: pack modes with a chunk size).
// The data gradient is the same computation with the taps mirrored and the dgrad-packed weights.
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

struct C3P {
    const bf16_t* x; const bf16_t* w; bf16_t* y; float* stats;
    int B, H, W, Cin, Cout, Cout_pad;
    int CK, CKpad, nchunk, nks_chunk, SK, nsl;          // channels per chunk (+8 pad in LDS), k-steps per chunk, slice
    int lTW, lTH, lTB, HH, HW;                          // log2 tile sizes; halo size
    int nty, ntx, flip;
    long nks_total;
};

struct Plan { int MT, NT, lTW, lTH, lTB, CK, SK; };

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// MT pixel tiles per wave x 4 waves = 64*MT pixels per block, arranged as 2^lTB images x 2^lTH x 2^lTW pixels.
Plan make_plan(int B, int H, int W, int Cin, int Cout) {
    Plan pl;
    pl.NT = danet_conv_nt(Cout);
    const int cpad = (Cout + 16 * pl.NT - 1) / (16 * pl.NT) * (16 * pl.NT);
    const int tw = 1 << (ilog2(W) < 4 ? ilog2(W) : 4);
    pl.MT = 1;
    for (int mt = 4; mt >= 1; mt >>= 1) {
        const int P = 64 * mt;
        int th = P / tw; const int hp = 1 << ilog2(H); if (th > hp) th = hp;
        const int tb = P / (tw * th);
        const long blocks = (long)((B + tb - 1) / tb) * ((H + th - 1) / th) * ((W + tw - 1) / tw) * (cpad / (16 * pl.NT));
        pl.MT = mt; pl.lTW = ilog2(tw); pl.lTH = ilog2(th); pl.lTB = ilog2(tb);
        if (blocks >= 256) break;
    }
    const int npix = (1 << pl.lTB) * ((1 << pl.lTH) + 2) * ((1 << pl.lTW) + 2);
    // channel chunk: halo tile <= 40 KB
    static const int cands[] = {96, 64, 48, 32, 16};
    pl.CK = 16;
    if (Cin <= 96 && (size_t)npix * (Cin + 8) * 2 <= 40 * 1024) pl.CK = Cin;
    else
        for (int c : cands)
            if (Cin % c == 0 && (size_t)npix * (c + 8) * 2 <= 40 * 1024) { pl.CK = c; break; }
    const int nks = (9 * pl.CK + 31) / 32;
    pl.SK = 1;
    for (int s = 9; s >= 1; --s) if (nks % s == 0) { pl.SK = s; break; }
    return pl;
}

constexpr int NHV = 10;      // halo prefetch registers (uint4) per thread: 2560 pieces = 40 KB

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv3x3_lds_kernel(C3P p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWV = (NT * 9 * 64 + 255) / 256;           // weight-slice prefetch registers (SK <= 9)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int TW = 1 << p.lTW, TH = 1 << p.lTH;
    const int npix_halo = (p.HH * p.HW) << p.lTB;
    // LDS: [tap table: nks_chunk*4 ints][weight slice: NT*SK KB][halo]
    int* const sTab = reinterpret_cast<int*>(smem);
    const unsigned tab_bytes = (unsigned)((p.nks_chunk * 4 * 4 + 15) / 16 * 16);
    unsigned char* const sW = smem + tab_bytes;
    unsigned char* const sH = sW + (size_t)NT * p.SK * 1024;

    // tile of this block
    int bx = blockIdx.x;
    const int txi = bx % p.ntx; bx /= p.ntx;
    const int tyi = bx % p.nty; const int bti = bx / p.nty;
    const int b0 = bti << p.lTB, oy0 = tyi << p.lTH, ox0 = txi << p.lTW;
    const int n0 = blockIdx.y * (16 * NT);

    for (int e = t; e < p.nks_chunk * 4; e += 256) {
        const int k = e * 8;
        int tap = k / p.CK, cl = k - tap * p.CK;
        if (tap >= 9) { tap = 0; cl = 0; }                    // zero-weight padding: any finite operand will do
        const int r = tap / 3, s = tap - r * 3;
        const int dr = p.flip ? 1 - r : r - 1, ds = p.flip ? 1 - s : s - 1;
        sTab[e] = ((dr * p.HW + ds) * p.CKpad + cl) * 2;
    }

    // pixels of this wave
    unsigned pixbase[MT];
    int ob[MT], oy[MT], ox[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 16 + li;
        const int tx = q & (TW - 1), ty = (q >> p.lTW) & (TH - 1), tb = q >> (p.lTW + p.lTH);
        pixbase[mt] = (unsigned)((((tb * p.HH + ty + 1) * p.HW + tx + 1) * p.CKpad) * 2);
        ob[mt] = b0 + tb; oy[mt] = oy0 + ty; ox[mt] = ox0 + tx;
    }

    // halo staging plan (same pixels for every chunk): global element offset (or -1) and LDS byte offset
    const int c8n = p.CK / 8;
    const int nph = npix_halo * c8n;
    long hsrc[NHV];
    unsigned hdst[NHV];
#pragma unroll
    for (int u = 0; u < NHV; ++u) {
        const int idx = t + u * 256;
        hsrc[u] = -1; hdst[u] = 0;
        if (idx < nph) {
            const int c8 = idx % c8n, hp = idx / c8n;
            const int hx = hp % p.HW; const int r2 = hp / p.HW;
            const int hy = r2 % p.HH, tb = r2 / p.HH;
            const int b = b0 + tb, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            hdst[u] = (unsigned)((hp * p.CKpad + c8 * 8) * 2);
            if (b < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                hsrc[u] = (((long)b * p.H + iy) * p.W + ix) * p.Cin + c8 * 8;
        }
    }
    uint4 hreg[NHV], wreg[NWV];
    auto fetch_halo = [&](int c) {
#pragma unroll
        for (int u = 0; u < NHV; ++u) {
            uint4 v = {0u, 0u, 0u, 0u};
            if (hsrc[u] >= 0) v = *reinterpret_cast<const uint4*>(p.x + hsrc[u] + (long)c * p.CK);
            hreg[u] = v;
        }
    };
    auto commit_halo = [&]() {
#pragma unroll
        for (int u = 0; u < NHV; ++u)
            if (t + u * 256 < nph) *reinterpret_cast<uint4*>(sH + hdst[u]) = hreg[u];
    };
    // weight slice `it`: NT runs of SK KB (fragment-major packing keeps a row tile's k-steps contiguous)
    const int wpieces = NT * p.SK * 64;
    const bf16_t* const wrow = p.w + (size_t)(n0 / 16) * p.nks_total * 512;
    auto fetch_w = [&](long it) {
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int idx = t + u * 256;
            if (idx < wpieces) {
                const int nt = idx / (p.SK * 64), r = idx - nt * (p.SK * 64);
                wreg[u] = *reinterpret_cast<const uint4*>(wrow + ((size_t)nt * p.nks_total + it * p.SK) * 512 + (size_t)r * 8);
            }
        }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int u = 0; u < NWV; ++u)
            if (t + u * 256 < wpieces) *reinterpret_cast<uint4*>(sW + (size_t)(t + u * 256) * 16) = wreg[u];
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch_halo(0);
    fetch_w(0);
    commit_halo();
    commit_w();
    __syncthreads();

    const long total = (long)p.nchunk * p.nsl;
    int c = 0, s = 0;
    for (long it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        const bool chunk_ends = s == p.nsl - 1;
        if (more) {
            fetch_w(it + 1);
            if (chunk_ends) fetch_halo(c + 1);
        }
        const int* tab = sTab + s * p.SK * 4 + lg;
        const unsigned char* wl = sW + lane * 16;
        for (int ks = 0; ks < p.SK; ++ks) {
            const unsigned e = (unsigned)tab[ks * 4];
            bf16x8 a[NT], bq[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                a[nt] = *reinterpret_cast<const bf16x8*>(wl + (size_t)(nt * p.SK + ks) * 1024);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                bq[mt] = *reinterpret_cast<const bf16x8*>(sH + (unsigned)(pixbase[mt] + e));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], bq[mt], acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            __syncthreads();                       // everyone is done with this slice (and halo, if the chunk ends)
            commit_w();
            if (chunk_ends) commit_halo();
            __syncthreads();
        }
        if (chunk_ends) { s = 0; ++c; } else ++s;
    }

    // fused BatchNorm statistics of the bf16 output (see conv_igemm.hip)
    if (p.stats) {
        __syncthreads();
        float* sStat = reinterpret_cast<float*>(sW);                 // [4 waves][2][NT*16], the weight slice is dead
        float s1[NT][4], s2[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bool ok = ob[mt] < p.B && oy[mt] < p.H && ox[mt] < p.W;
                    const float v = ok ? bf2f(f2bf(acc[mt][nt][r])) : 0.f;
                    a += v; b += v * v;
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                s1[nt][r] = a; s2[nt][r] = b;
            }
        if (li == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sStat[(wave * 2 + 0) * (NT * 16) + nt * 16 + lg * 4 + r] = s1[nt][r];
                    sStat[(wave * 2 + 1) * (NT * 16) + nt * 16 + lg * 4 + r] = s2[nt][r];
                }
        }
        __syncthreads();
        if (t < 2 * NT * 16) {
            const int which = t / (NT * 16), ch = t - which * (NT * 16);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += sStat[(w * 2 + which) * (NT * 16) + ch];
            const int cl = n0 + ch;
            if (cl < p.Cout)
                atomicAdd(p.stats + ((size_t)(blockIdx.x % BN_NCOPY) * 2 + which) * p.Cout + cl, v);
        }
    }

    // epilogue: lane holds couts n0 + nt*16 + lg*4 + {0..3} of its MT pixels
    const bool vec_ok = p.Cout % 4 == 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!(ob[mt] < p.B && oy[mt] < p.H && ox[mt] < p.W)) continue;
        bf16_t* yp = p.y + (((size_t)ob[mt] * p.H + oy[mt]) * p.W + ox[mt]) * p.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cl = n0 + nt * 16 + lg * 4;
            if (cl >= p.Cout) continue;
            if (vec_ok) {
                uint2 pk;
                pk.x = f2bf_pk(acc[mt][nt][0], acc[mt][nt][1]);
                pk.y = f2bf_pk(acc[mt][nt][2], acc[mt][nt][3]);
                *reinterpret_cast<uint2*>(yp + cl) = pk;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (cl + r < p.Cout) yp[cl + r] = f2bf(acc[mt][nt][r]);
            }
        }
    }
}

template <int MT, int NT>
void launch3(const C3P& p, size_t lds, dim3 grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_lds_kernel<MT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_lds_kernel<MT, NT>), grid, dim3(256), lds, st, p);
}


}  // namespace

// Is the LDS 3x3 kernel applicable?  (square 3x3, stride 1, pad 1, no dilation, one group, 16-channel granules)
extern "C" int danet_conv3x3_ok(int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups)
{
    return (R == 3 && S == 3 && stride == 1 && pad == 1 && dil == 1 && groups == 1 && Cin % 16 == 0 && Cin >= 16 &&
            Cout % 4 == 0 && H > 0 && W > 0) ? 1 : 0;
}

// Channel chunk the weights must be packed with (danet_conv_pack_weights modes 2 / 3) for this problem size.
extern "C" int danet_conv3x3_chunk(int B, int H, int W, int Cin, int Cout)
{
    return make_plan(B, H, W, Cin, Cout).CK;
}

// MT*10 + NT of the instance that runs (profiling attribution).
extern "C" int danet_conv3x3_kernel_id(int B, int H, int W, int Cin, int Cout)
{
    const Plan pl = make_plan(B, H, W, Cin, Cout);
    return pl.MT * 10 + pl.NT;
}

// y[B,H,W,Cout] = conv3x3(x[B,H,W,Cin]) with weights packed by danet_conv_pack_weights(mode 2, chunk) --
// or, with flip = 1 and mode-3 weights, the data gradient (x = dY with Cin = the layer's Cout channels).
extern "C" int danet_conv3x3_forward(const void* x, const void* wp, void* y, int B, int H, int W, int Cin, int Cout,
                                     int flip, float* bn_sums, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && wp && y, "conv3x3_forward: null pointer");
    DANET_CHECK_ARG(B > 0 && danet_conv3x3_ok(H, W, Cin, Cout, 3, 3, 1, 1, 1, 1), "conv3x3_forward: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
    const Plan pl = make_plan(B, H, W, Cin, Cout);
    C3P p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wp; p.y = (bf16_t*)y; p.stats = bn_sums;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.Cout_pad = (Cout + 16 * pl.NT - 1) / (16 * pl.NT) * (16 * pl.NT);
    p.CK = pl.CK; p.CKpad = pl.CK + 8; p.nchunk = Cin / pl.CK;
    p.nks_chunk = (9 * pl.CK + 31) / 32; p.SK = pl.SK; p.nsl = p.nks_chunk / pl.SK;
    p.nks_total = (long)p.nchunk * p.nks_chunk;
    p.lTW = pl.lTW; p.lTH = pl.lTH; p.lTB = pl.lTB;
    p.HH = (1 << pl.lTH) + 2; p.HW = (1 << pl.lTW) + 2;
    p.nty = (H + (1 << pl.lTH) - 1) >> pl.lTH; p.ntx = (W + (1 << pl.lTW) - 1) >> pl.lTW;
    p.flip = flip;
    const int nbt = (B + (1 << pl.lTB) - 1) >> pl.lTB;
    const size_t npix = (size_t)(p.HH * p.HW) << pl.lTB;
    DANET_CHECK_ARG(npix * (pl.CK / 8) <= (size_t)NHV * 256, "conv3x3_forward: halo tile too large");
    const size_t lds = (size_t)((p.nks_chunk * 16 + 15) / 16 * 16) + (size_t)pl.NT * pl.SK * 1024 + npix * p.CKpad * 2;
    DANET_CHECK_ARG(lds <= 96 * 1024, "conv3x3_forward: LDS budget exceeded (%zu bytes)", lds);
    const dim3 grid((unsigned)(nbt * p.nty * p.ntx), (unsigned)(p.Cout_pad / (16 * pl.NT)));
    hipStream_t st = (hipStream_t)stream;
#define C3_CASE(M_, N_) if (pl.MT == M_ && pl.NT == N_) launch3<M_, N_>(p, lds, grid, st); else
    C3_CASE(1, 1) C3_CASE(2, 1) C3_CASE(4, 1) C3_CASE(1, 2) C3_CASE(2, 2) C3_CASE(4, 2)
    C3_CASE(1, 3) C3_CASE(2, 3) C3_CASE(4, 3) C3_CASE(1, 4) C3_CASE(2, 4) C3_CASE(4, 4)
    return danet::fail(DANET_ERR_ARG, "conv3x3_forward: no kernel for tiles %dx%d", pl.MT, pl.NT);
#undef C3_CASE
    DANET_CHECK_LAUNCH("conv3x3_lds_kernel");
    return DANET_OK;
}
