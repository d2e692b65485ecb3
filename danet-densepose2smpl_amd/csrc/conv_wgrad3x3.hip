// Weight gradient of 3x3 / stride-1 / pad-1 convolutions (89.6 % of the backbone's MACs) using the
// gfx950 LDS transpose read.
//
//   dW[cout][tap][cin] = sum_pixels dY[pixel][cout] * X[pixel + tap][cin]
//
// The reduction (MFMA K) axis is the pixel index, but NHWC keeps channels contiguous.  Instead of
// transposing while staging (conv_wgrad.hip: one gather per tap, i.e. X is read 9 times), this kernel
//   * copies a 4x8-pixel tile of dY and the 6x10-pixel halo tile of X into LDS AS THEY ARE
//     (plain coalesced 16-byte copies; every input pixel is loaded ~1.9x instead of 9x), and
//   * reads MFMA fragments with ds_read_b64_tr_b16, which transposes on the fly: within a 16-lane
//     group, lane j receives R[j][e] = S[4e + (j>>2)][j&3] where S[i][0..3] are the four bf16 at
//     lane i's address (measured on MI355X, tools/experiments/tr_read.hip).  With lane i pointing
//     at (pixel q0 + (i>>2), channels c0 + 4(i&3) ..+3), lane j ends up with channel c0 + j of the
//     four pixels q0..q0+3 -- one half of an MFMA fragment (8 pixels of one channel row).
// Any pixel <-> k assignment is valid as long as dY and X use the same one, so the tap shift is
// just an address offset into the halo tile.
// Partial sums of the blocks that share a dW tile go to a [split] workspace with plain coalesced
// stores and are reduced (deterministically) by a second kernel that also writes the torch layout.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

constexpr int TH = 4, TW = 8;                 // output pixels per chunk (= one MFMA k-step of 32)
// halo tile of a chunk at stride ST (1 or 2): input rows ST*oy - 1 .. ST*(oy + TH - 1) + 1
constexpr int halo_h(int st) { return st * (TH - 1) + 3; }
constexpr int halo_w(int st) { return st * (TW - 1) + 3; }

struct Wg3P {
    const bf16_t* x; const bf16_t* dy; float* part;
    float* direct;                                      // msplit == 1 and beta == 0: the block IS the gradient -- written straight into dW
                                                        // (torch layout [Cout][Cin_g][3][3]), no partial block, no reduction pass
    int B, H, W, Cin, Cout, groups, Cin_g, Cout_g;      // H, W: OUTPUT size (= input size / stride)
    int IH, IW;                                         // input size
    int tiles_h, tiles_w, msplit;
    long nchunks;                              // B * tiles_h * tiles_w
    long x_bytes, dy_bytes;                    // extents from the group / channel-block offset on (buffer resources)
};

typedef __attribute__((ext_vector_type(2))) unsigned v2u;

__device__ inline v2u tr_read(unsigned lds_byte_addr) {
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr) : "memory");
    return r;
}

// One workgroup's share of one problem: (bx of msplit pixel ranges, by = cout-block x cin-block, bz = group).
// PAIR (round 6): 4 x 4 OUTPUT maps (the regressor tails: limb_net layer3 over the 768 part crops, /root/reference/models/module/res_module.py:393-464)
// do not hold a 4 x 8 chunk, so they ran on the generic gather kernel at 3 % of the peak (0.56 ms per step).  Here a chunk is TWO
// images side by side: the caller describes the tensors as [B / 2, 4, 8] (chunk c = images 2c, 2c + 1 = 32 consecutive pixels in
// memory), pixel (ty, tx) of the chunk is pixel (ty, tx & 3) of image tx >> 2, and each image gets a halo tile of its own (6 x 6,
// every border cell outside its image = zero): the staged strip is 6 x 12 and the second half of a fragment starts 6 columns on
// (stride 2: 8 x 8 inputs, 9 x 18 strip).
// SM = 2 (round 6, same idea): 2 x 2 output maps (body_net layer4, the grouped limb layer4: `LimbResLayers`, res_module.py:500-535) -- a
// chunk is EIGHT images, two rows of four, each behind its own 4 x 4 (stride 2: 5 x 5) halo; these layers' 2.4 M-element weight gradients
// were 71 M atomic adds on the generic kernel (287 us + 150 us of unpacking per step).
template <int CT, int NI, int ST, int SM = 0>
__device__ __forceinline__ void wgrad3x3_body(const Wg3P& p, const int bx, const int by, const int bz)
{
    constexpr bool PAIR = SM != 0;                               // small-map mode: SM = output map side (4: two images per chunk, 2: eight)
    constexpr int OS = SM ? SM : 4;                              // output rows / columns of one image
    constexpr int PHW = ST * (OS - 1) + 3;                       // halo columns (and rows) of ONE image
    constexpr int IMH = SM ? TH / OS : 1, IMW = SM ? TW / OS : 1; // images per chunk, vertically / horizontally
    constexpr int HH = SM ? IMH * PHW : halo_h(ST), HW = SM ? IMW * PHW : halo_w(ST);
    constexpr int HALF = SM ? (4 / OS) * PHW : ST * 4;           // staged columns between the two 4-pixel halves of a fragment
    constexpr int BCO = CT * 16, BCI = NI * 16;
    constexpr int PXY = BCO * 2, PXX = BCI * 2;                  // bytes per staged pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [2 buffers][dY tile 32 px | X halo 60 px]
    constexpr int YB = TH * TW * PXY, XB = HH * HW * PXX, BUF = YB + XB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nci = (p.Cin_g + BCI - 1) / BCI;
    const int cib = by % nci, cob = by / nci;
    const int g = bz;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const bf16_t* const dyg = p.dy + (size_t)g * p.Cout_g + co0;
    const bf16_t* const xg = p.x + (size_t)g * p.Cin_g + ci0;

    // (tap, ni) pairs round-robin over the 4 waves; each pair carries CT accumulator tiles
    constexpr int NPAIR = 9 * NI;
    constexpr int MAXP = (NPAIR + 3) / 4;
    f32x4 acc[MAXP][CT];
    unsigned boff[MAXP];                     // byte offset of the pair's B fragment (h = 0) in the X tile
#pragma unroll
    for (int pi = 0; pi < MAXP; ++pi) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[pi][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int pair = min(wave + 4 * pi, NPAIR - 1);
        const int tap = pair / NI, ni = pair - tap * NI;
        const int r = tap / 3, s = tap - r * 3;
        // lane i of 16-lane group lg supplies output pixel (ty = lg, tx = 4h + (i>>2)) = halo pixel (ST*ty + r, ST*tx + s),
        // channels ni*16 + 4(i&3)
        // (small-map mode: pixel (ty, tx) lies in image (ty / OS, tx / OS), whose halo starts PHW cells further per image)
        const int hrow = SM ? (lg / OS) * PHW + ST * (lg % OS) + r : ST * lg + r;
        const int hcol = SM ? ((li >> 2) / OS) * PHW + ST * ((li >> 2) % OS) + s : ST * (li >> 2) + s;
        boff[pi] = (unsigned)((hrow * HW + hcol) * PXX + (ni * 16 + 4 * (li & 3)) * 2);
    }
    const unsigned aoff = (unsigned)((lg * TW + (li >> 2)) * PXY + (4 * (li & 3)) * 2);   // dY fragment, h = 0, ct = 0

    const int nchunks = (int)p.nchunks;
    const int per = (nchunks + p.msplit - 1) / p.msplit;
    const int c_begin = bx * per, c_end = min(nchunks, c_begin + per);

    // staging: 16-byte pieces; dY tile = 32 px * (BCO/8) pieces, X halo = 60 px * (BCI/8) pieces.  What a thread
    // copies does not depend on the chunk: its piece's offset relative to the tile origin and its halo
    // coordinates are fixed, so a chunk costs one add and a bounds test per piece (32-bit buffer offsets;
    // out-of-image halo pixels get an out-of-range offset and load zeros).
    constexpr int NPY = TH * TW * (BCO / 8), NPX = HH * HW * (BCI / 8);
    constexpr int NRY = (NPY + 255) / 256, NRX = (NPX + 255) / 256;
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dyg), 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xg), 0, (int)p.x_bytes, 0x00020000);
    int yrel[NRY], xrel[NRX], xhy[NRX], xhx[NRX];
#pragma unroll
    for (int u = 0; u < NRY; ++u) {
        const int pc = t + u * 256;
        const int c8 = pc % (BCO / 8), q = pc / (BCO / 8);
        const int ty = q / TW, tx = q % TW;
        const int ypix = SM ? ((ty / OS) * IMW + tx / OS) * (OS * OS) + (ty % OS) * OS + tx % OS : ty * p.W + tx;
        yrel[u] = (pc < NPY && co0 + c8 * 8 < p.Cout_g) ? (ypix * p.Cout + c8 * 8) * 2 : OOB;
    }
#pragma unroll
    for (int u = 0; u < NRX; ++u) {
        const int pc = t + u * 256;
        const int c8 = pc % (BCI / 8), q = pc / (BCI / 8);
        const bool live = pc < NPX && ci0 + c8 * 8 < p.Cin_g;
        if (PAIR) {
            // halo cell (hq, hx): image (hq / PHW, hx / PHW), its input pixel (hq % PHW - 1, hx % PHW - 1) -- inside the (OS ST)^2 image or a
            // zero; static per piece
            const int hq = q / HW, img = (hq / PHW) * IMW + (q % HW) / PHW, hy = hq % PHW - 1, hc = (q % HW) % PHW - 1;
            const bool inside = live && (unsigned)hy < (unsigned)(OS * ST) && (unsigned)hc < (unsigned)(OS * ST);
            xrel[u] = ((img * (OS * OS * ST * ST) + hy * (OS * ST) + hc) * p.Cin + c8 * 8) * 2;
            xhy[u] = inside ? 1 : -100000;                                // (row test of fetch(): 0 <= -1 + 1 < IH)
            xhx[u] = 1;
        } else {
            xrel[u] = (((q / HW - 1) * p.IW + q % HW - 1) * p.Cin + c8 * 8) * 2;
            xhy[u] = live ? q / HW : -100000;                             // (-100000: never inside the image)
            xhx[u] = q % HW;
        }
    }
    // tile position of the current fetch (block-uniform), advanced chunk by chunk
    int f_tw = c_begin % p.tiles_w, f_th = (c_begin / p.tiles_w) % p.tiles_h, f_b = c_begin / (p.tiles_w * p.tiles_h);
    uint4 ystage[NRY], xstage[NRX];

    auto fetch = [&]() {
        const int oh0 = f_th * TH, ow0 = f_tw * TW;
        const int pix0 = (f_b * p.H + oh0) * p.W + ow0;
        const int by = pix0 * p.Cout * 2, bx = ((f_b * p.IH + ST * oh0) * p.IW + ST * ow0) * p.Cin * 2;
#pragma unroll
        for (int u = 0; u < NRY; ++u)
            ystage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yr, yrel[u] != OOB ? by + yrel[u] : OOB, 0, 0));
#pragma unroll
        for (int u = 0; u < NRX; ++u) {
            const bool ok = PAIR ? xhy[u] > 0
                                 : ((unsigned)(ST * oh0 - 1 + xhy[u]) < (unsigned)p.IH && (unsigned)(ST * ow0 - 1 + xhx[u]) < (unsigned)p.IW);
            xstage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? bx + xrel[u] : OOB, 0, 0));
        }
        if (++f_tw == p.tiles_w) { f_tw = 0; if (++f_th == p.tiles_h) { f_th = 0; ++f_b; } }
    };
    auto commit = [&](int buf) {
        unsigned char* base = smem + buf * BUF;                            // tiles are stored piece-linear: dY pieces, then X pieces
#pragma unroll
        for (int u = 0; u < NRY; ++u)
            if (t + u * 256 < NPY) *reinterpret_cast<uint4*>(base + (size_t)(t + u * 256) * 16) = ystage[u];
#pragma unroll
        for (int u = 0; u < NRX; ++u)
            if (t + u * 256 < NPX) *reinterpret_cast<uint4*>(base + (size_t)(NPY + t + u * 256) * 16) = xstage[u];
    };

    if (c_begin < c_end) fetch();
    int buf = 0;
    for (int ch = c_begin; ch < c_end; ++ch) {
        commit(buf);
        __syncthreads();                       // tile `buf` complete; previous reads of `buf^1` also done
        if (ch + 1 < c_end) fetch();
        const unsigned ybase = (unsigned)(buf * BUF), xbase = ybase + YB;
        // all transpose reads of the chunk are issued back to back, then ONE wait (the compiler does not
        // count LDS operations issued from inline asm) and a scheduling barrier so that no MFMA is hoisted
        // above the wait
        v2u alo[CT], ahi[CT], blo[MAXP], bhi[MAXP];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            alo[ct] = tr_read(ybase + aoff + ct * 32);
            ahi[ct] = tr_read(ybase + aoff + ct * 32 + 4 * PXY);
        }
#pragma unroll
        for (int pi = 0; pi < MAXP; ++pi) {
            blo[pi] = tr_read(xbase + boff[pi]);
            bhi[pi] = tr_read(xbase + boff[pi] + HALF * PXX);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 a[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const uint4 raw = {alo[ct].x, alo[ct].y, ahi[ct].x, ahi[ct].y};
            a[ct] = __builtin_bit_cast(bf16x8, raw);
        }
#pragma unroll
        for (int pi = 0; pi < MAXP; ++pi) {
            const uint4 raw = {blo[pi].x, blo[pi].y, bhi[pi].x, bhi[pi].y};
            const bf16x8 bq = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[pi][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ct], bq, acc[pi][ct], 0, 0, 0);
        }
        buf ^= 1;
    }
    // partial dW of this block: part[blockIdx.x][g][tap][cout][cin] (32-bit index arithmetic, one add per store)
    const int gsz = 9 * p.Cout_g * p.Cin_g;
    if (p.direct) {
        // the deep layers of a multi-problem launch (384 / 192 channels: 64 / 16 channel blocks per layer already fill their share of
        // the launch) take no pixel split: their only block used to travel through a partial copy and the reduction kernel for nothing
        float* const dw = p.direct + (size_t)g * gsz;
#pragma unroll
        for (int pi = 0; pi < MAXP; ++pi) {
            const int pair = wave + 4 * pi;
            const int tap = pair / NI, ni = pair - tap * NI;
            const int cin = ci0 + ni * 16 + li;
            if (pair >= NPAIR || cin >= p.Cin_g) continue;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cout = co0 + ct * 16 + lg * 4 + r;
                    if (cout < p.Cout_g) dw[((size_t)cout * p.Cin_g + cin) * 9 + tap] = acc[pi][ct][r];
                }
        }
        return;
    }
    float* dst = p.part + ((size_t)bx * p.groups + g) * gsz;
#pragma unroll
    for (int pi = 0; pi < MAXP; ++pi) {
        const int pair = wave + 4 * pi;
        const int tap = pair / NI, ni = pair - tap * NI;
        const int cin = ci0 + ni * 16 + li;
        if (pair >= NPAIR || cin >= p.Cin_g) continue;
        float* row = dst + ((tap * p.Cout_g + co0 + lg * 4) * p.Cin_g + cin);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co0 + ct * 16 + lg * 4 + r < p.Cout_g) row[(ct * 16 + r) * p.Cin_g] = acc[pi][ct][r];
    }
}

template <int CT, int NI, int ST>
__global__ __launch_bounds__(256) void conv_wgrad3x3_kernel(Wg3P p)
{
    wgrad3x3_body<CT, NI, ST>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent problems in one launch (weight gradients are only needed by the optimizer, so a trainer
// queues them during the backward pass and flushes the queue with a few of these): workgroup id -> problem by
// the prefix table, then (bx, by, bz) within the problem.  Fewer, longer workgroups per problem amortise the
// per-workgroup prologue / partial-sum traffic that dominates a single 17 us launch.
constexpr int NPM = 20;
struct Wg3Multi { Wg3P p[NPM]; int start[NPM + 1]; int nyb[NPM]; int n; int xcd; };

template <int CT, int NI, int ST, int SM = 0>
__global__ __launch_bounds__(256) void conv_wgrad3x3_multi_kernel(Wg3Multi mp)
{
    // XCD-contiguous order (round 6 experiment, DANET_WGRAD3_XCD=1; measured slower -- the 256 MB infinity cache already absorbs the
    // cross-XCD re-reads, as conv_fast.hip found for its halos): hardware workgroup b runs on XCD b % 8 (observed, grid_barrier.h).  With the plain order a
    // problem's workgroups are dealt round-robin over all eight XCDs, so each of the eight L2s fetches the problem's x and dy slices
    // for itself: a 384-channel layer's 3.2 MB of operands cross the fabric as ~26 MB.  Here XCD x takes the x-th eighth of the
    // virtual ids, i.e. a problem's (consecutive) workgroups share one or two L2s.
    int vb = (int)blockIdx.x;
    if (mp.xcd) {
        const int G = (int)gridDim.x, q = G >> 3, r = G & 7;
        const int x = vb & 7, slot = vb >> 3;
        vb = x * q + (x < r ? x : r) + slot;
    }
    int i = 0;
    while (i + 1 < mp.n && vb >= mp.start[i + 1]) ++i;
    const int l = vb - mp.start[i];
    const Wg3P& p = mp.p[i];
    const int bx = l % p.msplit, rest = l / p.msplit;
    wgrad3x3_body<CT, NI, ST, SM>(p, bx, rest % mp.nyb[i], rest / mp.nyb[i]);
}

struct Red3Multi { const float* part[NPM]; float* dw[NPM]; int G[NPM], Cout_g[NPM], Cin_g[NPM], msplit[NPM]; long start[NPM + 1]; int n; float beta; };

__global__ __launch_bounds__(256) void wgrad3x3_reduce_multi_kernel(Red3Multi rp)
{
    const long gidx = (long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= rp.start[rp.n]) return;
    int i = 0;
    while (i + 1 < rp.n && gidx >= rp.start[i + 1]) ++i;
    const long idx = gidx - rp.start[i], total = rp.start[i + 1] - rp.start[i];
    const float* part = rp.part[i];
    float s = 0.f;
    int sp = 0;
    for (; sp + 4 <= rp.msplit[i]; sp += 4) {
        const float v0 = part[(size_t)sp * total + idx], v1 = part[(size_t)(sp + 1) * total + idx];
        const float v2 = part[(size_t)(sp + 2) * total + idx], v3 = part[(size_t)(sp + 3) * total + idx];
        s += (v0 + v1) + (v2 + v3);
    }
    for (; sp < rp.msplit[i]; ++sp) s += part[(size_t)sp * total + idx];
    const int Cin_g = rp.Cin_g[i], Cout_g = rp.Cout_g[i];
    const int cin = (int)(idx % Cin_g);
    long rest = idx / Cin_g;
    const int cout = (int)(rest % Cout_g); rest /= Cout_g;
    const int tap = (int)(rest % 9), g = (int)(rest / 9);
    const size_t o = (((size_t)(g * Cout_g + cout)) * Cin_g + cin) * 9 + tap;
    float* dw = rp.dw[i];
    dw[o] = rp.beta != 0.f ? dw[o] * rp.beta + s : s;
}

// dW[Cout][Cin_g][3][3] = beta*dW + sum_s part[s][g][tap][cout][cin]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void wgrad3x3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                               int G, int Cout_g, int Cin_g, int msplit, float beta)
{
    const long total = (long)G * Cout_g * Cin_g * 9;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // index in part layout: [g][tap][cout][cin]
    if (idx >= total) return;
    float s = 0.f;
    int sp = 0;
    for (; sp + 4 <= msplit; sp += 4) {
        const float v0 = part[(size_t)sp * total + idx], v1 = part[(size_t)(sp + 1) * total + idx];
        const float v2 = part[(size_t)(sp + 2) * total + idx], v3 = part[(size_t)(sp + 3) * total + idx];
        s += (v0 + v1) + (v2 + v3);
    }
    for (; sp < msplit; ++sp) s += part[(size_t)sp * total + idx];
    const int cin = (int)(idx % Cin_g);
    long rest = idx / Cin_g;
    const int cout = (int)(rest % Cout_g); rest /= Cout_g;
    const int tap = (int)(rest % 9), g = (int)(rest / 9);
    const size_t o = (((size_t)(g * Cout_g + cout)) * Cin_g + cin) * 9 + tap;
    dw[o] = beta != 0.f ? dw[o] * beta + s : s;
}

template <int CT, int NI, int ST>
int launch3(const Wg3P& p, hipStream_t st) {
    const int nco = (p.Cout_g + CT * 16 - 1) / (CT * 16), nci = (p.Cin_g + NI * 16 - 1) / (NI * 16);
    const size_t lds = 2 * (size_t)(TH * TW * CT * 32 + halo_h(ST) * halo_w(ST) * NI * 32);
    hipLaunchKernelGGL((conv_wgrad3x3_kernel<CT, NI, ST>), dim3(p.msplit, nco * nci, p.groups), dim3(256), lds, st, p);
    return 0;
}

inline int tiles3(int c) { return c <= 16 ? 1 : (c <= 32 ? 2 : ((c % 48 == 0 || c <= 48) ? 3 : 4)); }

}  // namespace

// Applicability ((H, W) = INPUT size): 3x3, stride 1 or 2, pad 1, dilation 1, OW % 8 == 0, OH % 4 == 0, channels per
// group % 8 == 0.
extern "C" int danet_conv_wgrad3x3_ok(int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups) {
    static const bool no_s2 = getenv("DANET_NO_WGRAD3_S2") != nullptr;         // A/B knob
    if (!(R == 3 && S == 3 && (stride == 1 || (stride == 2 && !no_s2)) && pad == 1 && dil == 1)) return 0;
    if (H % stride != 0 || W % stride != 0) return 0;
    return (H / stride) % TH == 0 && (W / stride) % TW == 0 && (Cin / groups) % 8 == 0 && (Cout / groups) % 8 == 0;
}

// Pair mode (wgrad3x3_body<..., PAIR>): 4 x 4 maps, two images per chunk -- through danet_conv_wgrad3x3_multi only.
extern "C" int danet_conv_wgrad3x3_pair_ok(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups) {
    static const bool off = getenv("DANET_NO_WGRAD3_PAIR") != nullptr;         // A/B knob
    if (off || !(R == 3 && S == 3 && (stride == 1 || stride == 2) && pad == 1 && dil == 1 && H == W && B > 0 && groups > 0)) return 0;
    if (!((H == 4 * stride && B % 2 == 0) || (H == 2 * stride && B % 8 == 0))) return 0;          // 4 x 4 outputs: image pairs; 2 x 2: octets
    if (Cin % groups != 0 || Cout % groups != 0) return 0;
    const int ci = Cin / groups, co = Cout / groups;
    if (ci % 8 != 0 || co % 8 != 0) return 0;
    // instantiated for 32- and 48-wide blocks (tiles3: 32 < c and c % 48 != 0 -> 2 tiles; c % 48 == 0 or c <= 48 -> 3)
    return 1;
}
// small-map mode of a problem: the output map side (4 or 2) when it runs with several images per chunk, 0 otherwise
static inline int wg3_small(int B, int H, int W, int stride) {
    if (!(stride == 1 || stride == 2) || H != W) return 0;
    if (H == 4 * stride && B % 2 == 0) return 4;
    if (H == 2 * stride && B % 8 == 0) return 2;
    return 0;
}
static inline bool wg3_is_pair(int B, int H, int W, int stride) { return wg3_small(B, H, W, stride) != 0; }

// (H, W): OUTPUT size
static void plan3(int B, int H, int W, int Cin, int Cout, int groups, int* ct, int* ni, int* msplit) {
    const int Cout_g = Cout / groups, Cin_g = Cin / groups;
    *ct = tiles3(Cout_g); *ni = tiles3(Cin_g);
    if (*ct == 4) *ct = 2;                      // accumulators: ceil(9*NI/4)*CT tiles per wave
    if (*ni == 4) *ni = 2;
    // (round 6, measured and dropped: 96 x 48 blocks for the 96 / 192 / 384-channel layers -- x staged half as often, 42 instead of 21
    //  MFMAs per wave between two barriers, but 265 registers = ONE workgroup per compute unit instead of three: the 12-problem flush of
    //  tools/wgrad3_bench.py took 229 us against 107, the step 26.58 ms against 26.20)
    const long other = (long)((Cout_g + *ct * 16 - 1) / (*ct * 16)) * ((Cin_g + *ni * 16 - 1) / (*ni * 16)) * groups;
    const long nchunks = (long)B * (H / TH) * (W / TW);
    long target = 256;
    if (const char* e = getenv("DANET_WGRAD3_BLOCKS")) target = atol(e);
    long ms = (target + other - 1) / other;
    if (ms > nchunks / 4) ms = nchunks / 4;
    if (ms < 1) ms = 1;
    *msplit = (int)ms;
}

// CT*10 + NI of the conv_wgrad3x3_kernel<CT, NI> instance that runs (profiling attribution).
extern "C" int danet_conv_wgrad3x3_kernel_id(int B, int H, int W, int Cin, int Cout, int groups, int stride) {
    int ct, ni, ms;
    plan3(B, H / stride, W / stride, Cin, Cout, groups, &ct, &ni, &ms);
    return ct * 10 + ni;
}

extern "C" size_t danet_conv_wgrad3x3_ws_floats(int B, int H, int W, int Cin, int Cout, int groups, int stride) {
    int ct, ni, ms;
    if (stride < 1) return 0;
    plan3(B, H / stride, W / stride, Cin, Cout, groups, &ct, &ni, &ms);
    return (size_t)ms * Cout * (Cin / groups) * 9;
}

// x [B,H,W,Cin], dy [B,H/stride,W/stride,Cout]
extern "C" int danet_conv_wgrad3x3(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                                   int B, int H, int W, int Cin, int Cout, int groups, int stride, float beta, int phase, void* stream)
{
    // phase: 0 = both kernels; 1 = the MFMA kernel only (partials into ws); 2 = the reduction only (profiling brackets)
    DANET_ENTER();
    DANET_CHECK_ARG(x && dy && dw && ws && B > 0 && phase >= 0 && phase <= 2, "conv_wgrad3x3: bad arguments");
    DANET_CHECK_ARG(danet_conv_wgrad3x3_ok(H, W, Cin, Cout, 3, 3, stride, 1, 1, groups), "conv_wgrad3x3: unsupported shape");
    Wg3P p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.part = ws; p.direct = nullptr;
    p.IH = H; p.IW = W; H /= stride; W /= stride;                      // from here on (H, W) = output size
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.groups = groups;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups;
    p.tiles_h = H / TH; p.tiles_w = W / TW;
    p.nchunks = (long)B * p.tiles_h * p.tiles_w;
    p.x_bytes = (long)B * p.IH * p.IW * Cin * 2; p.dy_bytes = (long)B * H * W * Cout * 2;
    DANET_CHECK_ARG(p.x_bytes < (1L << 31) && p.dy_bytes < (1L << 31), "conv_wgrad3x3: tensors of 2 GB or more are not supported");
    int ct, ni;
    plan3(B, H, W, Cin, Cout, groups, &ct, &ni, &p.msplit);
    if (ws_floats < (size_t)p.msplit * Cout * p.Cin_g * 9)
        return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad3x3: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (phase != 2) {
#define W3(a, b) if (ct == a && ni == b) { if (stride == 1) launch3<a, b, 1>(p, st); else launch3<a, b, 2>(p, st); } else
        W3(1, 1) W3(1, 2) W3(1, 3) W3(2, 1) W3(2, 2) W3(2, 3) W3(3, 1) W3(3, 2) W3(3, 3)
        return danet::fail(DANET_ERR_ARG, "conv_wgrad3x3: no kernel for tiles %dx%d", ct, ni);
#undef W3
        DANET_CHECK_LAUNCH("conv_wgrad3x3_kernel");
    }
    if (phase == 1) return DANET_OK;
    const long total = (long)Cout * p.Cin_g * 9;
    hipLaunchKernelGGL(wgrad3x3_reduce_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, st, ws, dw, groups, p.Cout_g,
                       p.Cin_g, p.msplit, beta);
    DANET_CHECK_LAUNCH("wgrad3x3_reduce_kernel");
    return DANET_OK;
}


// ---------------------------------------------------------------------------------------------
// Batched form: n independent 3x3 weight-gradient problems (jobs[i] = {x, dy, dw, B, H, W, Cin, Cout, groups, stride};
// (H, W) = input size), grouped internally by kernel instance and launched NPM at a time.  ws must hold
// danet_conv_wgrad3x3_multi_ws_floats(jobs, n) floats.
struct Wg3Job { const void* x; const void* dy; float* dw; int B, H, W, Cin, Cout, groups, stride; };

static long multi_target_blocks() {
    if (const char* e = getenv("DANET_WGRAD3_MULTI_BLOCKS")) return atol(e);
    return 768;          // swept on MI355X (384 / 512 / 768 / 1024): three workgroups per CU
}

// A problem's share of a multi-problem launch ~ pixel chunks x channel products, with a floor on the channel product: a layer of 12 or
// 16 channels (the channel-padded head layers, queued since round 5) is bound by reading its pixels, not by its few MFMAs, and with its
// true product it got one or two workgroups for 4 096 chunks -- a serial tail the whole launch waited for.
static inline double wg3_weight(int Cout, int Cin_g) { const double p = (double)Cout * Cin_g; return p < 2304.0 ? 2304.0 : p; }
// msplit of every job when jobs [first, last) of one instance share a launch
static void plan_multi(const Wg3Job* jobs, const int* idx, int cnt, int ct, int ni, int* msplit) {
    auto chunks_of = [](const Wg3Job& j) -> long {
        const int sm = wg3_small(j.B, j.H, j.W, j.stride);
        return sm ? (long)(j.B / (sm == 4 ? 2 : 8)) : (long)j.B * (j.H / j.stride / TH) * (j.W / j.stride / TW);
    };
    double tot = 0;
    for (int k = 0; k < cnt; ++k) {
        const Wg3Job& j = jobs[idx[k]];
        tot += (double)chunks_of(j) * wg3_weight(j.Cout, j.Cin / j.groups);
    }
    const long target = multi_target_blocks();
    for (int k = 0; k < cnt; ++k) {
        const Wg3Job& j = jobs[idx[k]];
        const int Cout_g = j.Cout / j.groups, Cin_g = j.Cin / j.groups;
        const long other = (long)((Cout_g + ct * 16 - 1) / (ct * 16)) * ((Cin_g + ni * 16 - 1) / (ni * 16)) * j.groups;
        const long nchunks = chunks_of(j);
        const double w = (double)nchunks * wg3_weight(j.Cout, Cin_g);
        long ms = (long)(target * (w / tot) / other + 0.5);
        if (ms > nchunks / 4) ms = nchunks / 4;
        if (ms < 1) ms = 1;
        // (forcing ms 2 -> 1 for the 192-channel layers so that they go direct as well measured +0.4 ms/step: half as many workgroups for them)
        msplit[k] = (int)ms;
    }
}

static int multi_foreach_launch(const Wg3Job* jobs, int n, float* ws, size_t ws_floats, float beta, hipStream_t st, size_t* need_out)
{
    size_t used = 0;
    bool done[4096];
    if (n > 4096) return danet::fail(DANET_ERR_ARG, "conv_wgrad3x3_multi: too many jobs (%d)", n);
    for (int i = 0; i < n; ++i) done[i] = false;
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        int ct, ni, dummy;
        const int stride = jobs[i].stride;
        const int sm = wg3_small(jobs[i].B, jobs[i].H, jobs[i].W, stride);
        const bool pair = sm != 0;
        plan3(jobs[i].B, jobs[i].H / stride, jobs[i].W / stride, jobs[i].Cin, jobs[i].Cout, jobs[i].groups, &ct, &ni, &dummy);
        static const int npm_env = getenv("DANET_WGRAD3_NPM") ? atoi(getenv("DANET_WGRAD3_NPM")) : 0;      // problems per launch (A-B knob)
        const int npm = npm_env > 0 && npm_env < NPM ? npm_env : NPM;
        int idx[NPM], cnt = 0;
        for (int k = i; k < n && cnt < npm; ++k) {
            if (done[k] || jobs[k].stride != stride || wg3_small(jobs[k].B, jobs[k].H, jobs[k].W, stride) != sm) continue;
            int c2, n2;
            plan3(jobs[k].B, jobs[k].H / stride, jobs[k].W / stride, jobs[k].Cin, jobs[k].Cout, jobs[k].groups, &c2, &n2, &dummy);
            if (c2 == ct && n2 == ni) { idx[cnt++] = k; done[k] = true; }
        }
        int msplit[NPM];
        plan_multi(jobs, idx, cnt, ct, ni, msplit);
        Wg3Multi mp; Red3Multi rp;
        static const bool xcd_order = getenv("DANET_WGRAD3_XCD") && atoi(getenv("DANET_WGRAD3_XCD")) != 0;      // A-B knob, OFF: measured 117.7 vs 110.1 us per 12-problem flush, step 26.26 vs 26.21 ms
        mp.n = cnt; rp.n = cnt; rp.beta = beta; mp.xcd = xcd_order ? 1 : 0;
        mp.start[0] = 0; rp.start[0] = 0;
        for (int k = 0; k < cnt; ++k) {
            const Wg3Job& j = jobs[idx[k]];
            Wg3P& p = mp.p[k];
            static const bool no_direct = getenv("DANET_WGRAD3_NO_DIRECT") != nullptr;          // A-B timing knob
            const bool direct = msplit[k] == 1 && beta == 0.f && !no_direct;
            p.x = (const bf16_t*)j.x; p.dy = (const bf16_t*)j.dy; p.part = (ws && !direct) ? ws + used : nullptr;
            p.direct = direct ? j.dw : nullptr;
            p.IH = j.H; p.IW = j.W;
            p.B = j.B; p.H = j.H / stride; p.W = j.W / stride; p.Cin = j.Cin; p.Cout = j.Cout; p.groups = j.groups;
            if (pair) { p.B = j.B / (sm == 4 ? 2 : 8); p.H = 4; p.W = 8; p.IH = 4 * stride; p.IW = 8 * stride; }      // 2 (8) images = one 4 x 8 chunk (32 consecutive output pixels)
            p.Cin_g = j.Cin / j.groups; p.Cout_g = j.Cout / j.groups;
            p.tiles_h = p.H / TH; p.tiles_w = p.W / TW;
            p.nchunks = (long)p.B * p.tiles_h * p.tiles_w;
            p.x_bytes = (long)j.B * j.H * j.W * j.Cin * 2; p.dy_bytes = (long)j.B * (j.H / stride) * (j.W / stride) * j.Cout * 2;
            p.msplit = msplit[k];
            const int nyb = ((p.Cout_g + ct * 16 - 1) / (ct * 16)) * ((p.Cin_g + ni * 16 - 1) / (ni * 16));
            mp.nyb[k] = nyb;
            mp.start[k + 1] = mp.start[k] + msplit[k] * nyb * j.groups;
            const long total = (long)j.Cout * p.Cin_g * 9;
            rp.part[k] = p.part; rp.dw[k] = j.dw; rp.G[k] = j.groups; rp.Cout_g[k] = p.Cout_g; rp.Cin_g[k] = p.Cin_g; rp.msplit[k] = msplit[k];
            rp.start[k + 1] = rp.start[k] + (direct ? 0 : total);          // (nothing to reduce for a direct job)
            if (!direct || !ws) used += (size_t)msplit[k] * total;          // (the sizing pass does not know beta: room for the partial form)
            if (ws && !(p.x_bytes < (1L << 31) && p.dy_bytes < (1L << 31)))
                return danet::fail(DANET_ERR_ARG, "conv_wgrad3x3_multi: tensors of 2 GB or more are not supported");
        }
        if (!ws) continue;                                   // sizing pass
        if (used > ws_floats) return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad3x3_multi: workspace too small");
        const int phw = stride * ((sm ? sm : 4) - 1) + 3;                                   // halo side of one image in small-map mode
        const size_t lds = 2 * (size_t)(TH * TW * ct * 32 + (sm ? (TH / sm) * phw * (TW / sm) * phw : halo_h(stride) * halo_w(stride)) * ni * 32);
        const dim3 grid((unsigned)mp.start[cnt]);
        if (pair) {
#define W3P(a, b) if (ct == a && ni == b) { \
            if (stride == 1 && sm == 4) hipLaunchKernelGGL((conv_wgrad3x3_multi_kernel<a, b, 1, 4>), grid, dim3(256), lds, st, mp); \
            else if (stride == 2 && sm == 4) hipLaunchKernelGGL((conv_wgrad3x3_multi_kernel<a, b, 2, 4>), grid, dim3(256), lds, st, mp); \
            else if (stride == 1) hipLaunchKernelGGL((conv_wgrad3x3_multi_kernel<a, b, 1, 2>), grid, dim3(256), lds, st, mp); \
            else hipLaunchKernelGGL((conv_wgrad3x3_multi_kernel<a, b, 2, 2>), grid, dim3(256), lds, st, mp); } else
            W3P(2, 2) W3P(2, 3) W3P(3, 2) W3P(3, 3) W3P(1, 1) W3P(1, 2) W3P(2, 1) W3P(1, 3) W3P(3, 1)
            return danet::fail(DANET_ERR_ARG, "conv_wgrad3x3_multi: no pair-mode kernel for tiles %dx%d", ct, ni);
#undef W3P
        } else
#define W3M(a, b) if (ct == a && ni == b) { \
            if (stride == 1) hipLaunchKernelGGL((conv_wgrad3x3_multi_kernel<a, b, 1>), grid, dim3(256), lds, st, mp); \
            else hipLaunchKernelGGL((conv_wgrad3x3_multi_kernel<a, b, 2>), grid, dim3(256), lds, st, mp); } else
        W3M(1, 1) W3M(1, 2) W3M(1, 3) W3M(2, 1) W3M(2, 2) W3M(2, 3) W3M(3, 1) W3M(3, 2) W3M(3, 3)
        return danet::fail(DANET_ERR_ARG, "conv_wgrad3x3_multi: no kernel for tiles %dx%d", ct, ni);
#undef W3M
        DANET_CHECK_LAUNCH("conv_wgrad3x3_multi_kernel");
        if (rp.start[cnt] > 0) {
            hipLaunchKernelGGL(wgrad3x3_reduce_multi_kernel, dim3((unsigned)danet::cdiv(rp.start[cnt], 256)), dim3(256), 0, st, rp);
            DANET_CHECK_LAUNCH("wgrad3x3_reduce_multi_kernel");
        }
    }
    if (need_out) *need_out = used;
    return DANET_OK;
}

extern "C" size_t danet_conv_wgrad3x3_multi_ws_floats(const void* jobs, int n)
{
    size_t need = 0;
    if (!jobs || n <= 0) return 0;
    multi_foreach_launch((const Wg3Job*)jobs, n, nullptr, 0, 0.f, nullptr, &need);
    return need;
}

extern "C" int danet_conv_wgrad3x3_multi(const void* jobs, int n, float* ws, size_t ws_floats, float beta, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(jobs && n > 0 && ws, "conv_wgrad3x3_multi: bad arguments");
    const Wg3Job* jb = (const Wg3Job*)jobs;
    for (int i = 0; i < n; ++i)
        DANET_CHECK_ARG(jb[i].x && jb[i].dy && jb[i].dw && jb[i].B > 0 &&
                        (danet_conv_wgrad3x3_ok(jb[i].H, jb[i].W, jb[i].Cin, jb[i].Cout, 3, 3, jb[i].stride, 1, 1, jb[i].groups) ||
                         danet_conv_wgrad3x3_pair_ok(jb[i].B, jb[i].H, jb[i].W, jb[i].Cin, jb[i].Cout, 3, 3, jb[i].stride, 1, 1, jb[i].groups)),
                        "conv_wgrad3x3_multi: job %d is not a supported 3x3 problem", i);
    return multi_foreach_launch(jb, n, ws, ws_floats, beta, (hipStream_t)stream, nullptr);
}
