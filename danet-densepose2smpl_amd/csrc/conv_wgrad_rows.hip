// Weight gradient of large-kernel / strided convolutions (the 7x7 stride-2 stems of the SMPL regressor's ResNets,
// /root/reference/models/module/res_module.py:407: 315 GFLOP per pass at 768 part crops) with the gfx950 LDS
// transpose read, one filter ROW per workgroup.
//
//   dW[cout][r][s][cin] = sum_pixels dY[b, oy, ox, cout] * X[b, ST*oy + r - pad, ST*ox + s - pad, cin]
//
// The generic kernel (conv_wgrad.hip) gathers X once per tap: 49 passes over a 403 MB tensor, bound by the
// texture-address path.  Here a workgroup owns filter row r and a range of 4x8-pixel output chunks.  Per chunk it
// copies the dY tile (32 pixels) and the 4 input rows ST*(oy0 + 0..3) + r - pad, ST*7 + S columns wide, into LDS
// exactly as they are (coalesced 16-byte pieces), and every (s, 16-cin) pair reads its MFMA B fragment from that
// one staged strip with ds_read_b64_tr_b16 at a tap- and stride-dependent address (the transposing read puts the
// pixel index on the MFMA K axis; see conv_wgrad3x3.hip for the lane mapping).  X is read R/ST * (ST*7+S)/(ST*8)
// = 4.6 times instead of 49, dY R = 7 times.  LDS pixel strides are padded by 16 bytes so that the stride-ST pixel
// step of a 16-lane read group does not fold onto the same banks.
// Partial sums of the workgroups that share a filter row go to a [split] workspace (plain stores) and are reduced
// in a fixed order by a second kernel that writes the torch layout.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

constexpr int TH = 4, TW = 8;                 // output pixels per chunk (= one MFMA k-step of 32)

struct WgRP {
    const bf16_t* x; const bf16_t* dy; float* part;
    int B, H, W, Cin, OH, OW, Cout, groups, Cin_g, Cout_g;
    int R, pad;
    int tiles_h, tiles_w, msplit, nchunks;
    long x_bytes, dy_bytes;
};

typedef __attribute__((ext_vector_type(2))) unsigned v2u;

__device__ inline v2u tr_read(unsigned lds_byte_addr) {
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr) : "memory");
    return r;
}

// grid: (msplit pixel ranges, R * cout-blocks * cin-blocks, groups)
template <int CT, int NI, int S, int ST>
__global__ __launch_bounds__(256) void conv_wgrad_rows_kernel(WgRP p)
{
    constexpr int BCO = CT * 16, BCI = NI * 16;
    constexpr int PXY = BCO * 2 + 16, PXX = BCI * 2 + 16;        // bytes per staged pixel (+16: bank spread)
    constexpr int HWc = ST * (TW - 1) + S;                       // staged input columns
    constexpr int YB = TH * TW * PXY, XB = TH * HWc * PXX, BUF = YB + XB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nci = (p.Cin_g + BCI - 1) / BCI, nco = (p.Cout_g + BCO - 1) / BCO;
    const int by = blockIdx.y;
    const int cib = by % nci, cob = (by / nci) % nco, r = by / (nci * nco);
    const int g = blockIdx.z, bx = blockIdx.x;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const bf16_t* const dyg = p.dy + (size_t)g * p.Cout_g + co0;
    const bf16_t* const xg = p.x + (size_t)g * p.Cin_g + ci0;

    // (s, ni) pairs round-robin over the 4 waves; each pair carries CT accumulator tiles
    constexpr int NPAIR = S * NI;
    constexpr int MAXP = (NPAIR + 3) / 4;
    f32x4 acc[MAXP][CT];
    unsigned boff[MAXP];
#pragma unroll
    for (int pi = 0; pi < MAXP; ++pi) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[pi][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int pair = min(wave + 4 * pi, NPAIR - 1);
        const int s = pair / NI, ni = pair - s * NI;
        // lane i of 16-lane group lg supplies output pixel (ty = lg, tx = 4h + (i>>2)) = staged pixel (row lg, column ST*tx + s)
        boff[pi] = (unsigned)((lg * HWc + ST * (li >> 2) + s) * PXX + (ni * 16 + 4 * (li & 3)) * 2);
    }
    const unsigned aoff = (unsigned)((lg * TW + (li >> 2)) * PXY + (4 * (li & 3)) * 2);

    const int per = (p.nchunks + p.msplit - 1) / p.msplit;
    const int c_begin = bx * per, c_end = min(p.nchunks, c_begin + per);

    constexpr int NPY = TH * TW * (BCO / 8), NPX = TH * HWc * (BCI / 8);
    constexpr int NRY = (NPY + 255) / 256, NRX = (NPX + 255) / 256;
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dyg), 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xg), 0, (int)p.x_bytes, 0x00020000);
    int yrel[NRY], ylds[NRY], xrel[NRX], xlds[NRX], xhy[NRX], xhx[NRX];
#pragma unroll
    for (int u = 0; u < NRY; ++u) {
        const int pc = t + u * 256;
        const int c8 = pc % (BCO / 8), q = pc / (BCO / 8);
        yrel[u] = (pc < NPY && co0 + c8 * 8 < p.Cout_g) ? (((q / TW) * p.OW + q % TW) * p.Cout + c8 * 8) * 2 : OOB;
        ylds[u] = pc < NPY ? q * PXY + c8 * 16 : -1;
    }
#pragma unroll
    for (int u = 0; u < NRX; ++u) {
        const int pc = t + u * 256;
        const int c8 = pc % (BCI / 8), q = pc / (BCI / 8);
        const int hy = q / HWc, hx = q % HWc;
        xrel[u] = ((ST * hy * p.W + hx) * p.Cin + c8 * 8) * 2;
        const bool live = pc < NPX && ci0 + c8 * 8 < p.Cin_g;
        xhy[u] = live ? ST * hy + r - p.pad : -100000000;                  // (never inside the image)
        xhx[u] = hx - p.pad;
        xlds[u] = pc < NPX ? YB + q * PXX + c8 * 16 : -1;
    }
    int f_tw = c_begin % p.tiles_w, f_th = (c_begin / p.tiles_w) % p.tiles_h, f_b = c_begin / (p.tiles_w * p.tiles_h);
    uint4 ystage[NRY], xstage[NRX];

    auto fetch = [&]() {
        const int oh0 = f_th * TH, ow0 = f_tw * TW;
        const int byo = ((f_b * p.OH + oh0) * p.OW + ow0) * p.Cout * 2;
        // byte offset of input pixel (b, ST*oh0 + r - pad, ST*ow0 - pad): may be negative, used only where the pixel is inside
        const int bxo = ((f_b * p.H + ST * oh0 + r - p.pad) * p.W + ST * ow0 - p.pad) * p.Cin * 2;
#pragma unroll
        for (int u = 0; u < NRY; ++u)
            ystage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yr, yrel[u] != OOB ? byo + yrel[u] : OOB, 0, 0));
#pragma unroll
        for (int u = 0; u < NRX; ++u) {
            const bool ok = (unsigned)(ST * oh0 + xhy[u]) < (unsigned)p.H && (unsigned)(ST * ow0 + xhx[u]) < (unsigned)p.W;
            xstage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? bxo + xrel[u] : OOB, 0, 0));
        }
        if (++f_tw == p.tiles_w) { f_tw = 0; if (++f_th == p.tiles_h) { f_th = 0; ++f_b; } }
    };
    auto commit = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int u = 0; u < NRY; ++u)
            if (ylds[u] >= 0) *reinterpret_cast<uint4*>(base + ylds[u]) = ystage[u];
#pragma unroll
        for (int u = 0; u < NRX; ++u)
            if (xlds[u] >= 0) *reinterpret_cast<uint4*>(base + xlds[u]) = xstage[u];
    };

    if (c_begin < c_end) fetch();
    int buf = 0;
    for (int ch = c_begin; ch < c_end; ++ch) {
        commit(buf);
        __syncthreads();                       // tile `buf` complete; previous reads of `buf^1` also done
        if (ch + 1 < c_end) fetch();
        const unsigned ybase = (unsigned)(buf * BUF), xbase = ybase;       // (the X strip starts YB bytes into the buffer)
        v2u alo[CT], ahi[CT], blo[MAXP], bhi[MAXP];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            alo[ct] = tr_read(ybase + aoff + ct * 32);
            ahi[ct] = tr_read(ybase + aoff + ct * 32 + 4 * PXY);
        }
#pragma unroll
        for (int pi = 0; pi < MAXP; ++pi) {
            blo[pi] = tr_read(xbase + YB + boff[pi]);
            bhi[pi] = tr_read(xbase + YB + boff[pi] + ST * 4 * PXX);
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
    // partial dW of this block: part[bx][g][tap][cout][cin]
    const int taps = p.R * S;
    const int gsz = taps * p.Cout_g * p.Cin_g;
    float* dst = p.part + ((size_t)bx * p.groups + g) * gsz;
#pragma unroll
    for (int pi = 0; pi < MAXP; ++pi) {
        const int pair = wave + 4 * pi;
        const int s = pair / NI, ni = pair - s * NI;
        const int cin = ci0 + ni * 16 + li;
        if (pair >= NPAIR || cin >= p.Cin_g) continue;
        float* row = dst + (((r * S + s) * p.Cout_g + co0 + lg * 4) * p.Cin_g + cin);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (co0 + ct * 16 + lg * 4 + rr < p.Cout_g) row[(ct * 16 + rr) * p.Cin_g] = acc[pi][ct][rr];
    }
}

// dW[Cout][Cin_g][R][S] = beta*dW + sum_s part[s][g][tap][cout][cin]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void wgrad_rows_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 int G, int Cout_g, int Cin_g, int taps, int msplit, float beta)
{
    const long total = (long)G * Cout_g * Cin_g * taps;
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
    const int tap = (int)(rest % taps), g = (int)(rest / taps);
    const size_t o = (((size_t)(g * Cout_g + cout)) * Cin_g + cin) * taps + tap;
    dw[o] = beta != 0.f ? dw[o] * beta + s : s;
}

int plan_rows(int B, int OH, int OW, int Cin, int Cout, int R, int groups) {
    const int Cout_g = Cout / groups, Cin_g = Cin / groups;
    const long other = (long)R * ((Cout_g + 63) / 64) * ((Cin_g + 63) / 64) * groups;
    const long nchunks = (long)B * (OH / TH) * (OW / TW);
    long target = 1024;
    if (const char* e = getenv("DANET_WGRAD_ROWS_BLOCKS")) target = atol(e);
    long ms = (target + other - 1) / other;
    if (ms > nchunks / 4) ms = nchunks / 4;
    if (ms < 1) ms = 1;
    return (int)ms;
}

}  // namespace

// Applicability: 7x7, stride 2, pad 3, dilation 1, OH % 4 == 0, OW % 8 == 0, channels per group % 8 == 0 and enough
// work to pay for the partial-sum pass (the small B = 32 stems stay on the generic kernel).
extern "C" int danet_conv_wgrad_rows_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups) {
    if (getenv("DANET_NO_WGRAD_ROWS")) return 0;
    if (!(R == 7 && S == 7 && stride == 2 && pad == 3 && dil == 1 && groups > 0 && Cin % groups == 0 && Cout % groups == 0)) return 0;
    if (OH != (H + 2 * pad - R) / stride + 1 || OW != (W + 2 * pad - S) / stride + 1) return 0;
    if (OH % TH != 0 || OW % TW != 0 || (Cin / groups) % 8 != 0 || (Cout / groups) % 8 != 0) return 0;
    if ((long)B * H * W * Cin * 2 >= (1L << 31) || (long)B * OH * OW * Cout * 2 >= (1L << 31)) return 0;
    return (long)B * OH * OW >= 65536;
}

extern "C" size_t danet_conv_wgrad_rows_ws_floats(int B, int OH, int OW, int Cin, int Cout, int R, int S, int groups) {
    return (size_t)plan_rows(B, OH, OW, Cin, Cout, R, groups) * Cout * (Cin / groups) * R * S;
}

// dW[Cout][Cin/groups][R][S] (fp32, torch layout) = beta * dW + conv_wgrad(x, dy);  x [B,H,W,Cin], dy [B,OH,OW,Cout] bf16 NHWC.
// ws: danet_conv_wgrad_rows_ws_floats() floats of scratch (every element is written before it is read).
extern "C" int danet_conv_wgrad_rows(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats,
                                     int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                     int R, int S, int stride, int pad, int groups, float beta, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && dy && dw && ws && B > 0, "conv_wgrad_rows: bad arguments");
    DANET_CHECK_ARG(danet_conv_wgrad_rows_ok(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, 1, groups) || getenv("DANET_NO_WGRAD_ROWS"),
                    "conv_wgrad_rows: unsupported shape");
    WgRP p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.part = ws;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.groups = groups;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups; p.R = R; p.pad = pad;
    p.tiles_h = OH / TH; p.tiles_w = OW / TW;
    p.nchunks = B * p.tiles_h * p.tiles_w;
    p.x_bytes = (long)B * H * W * Cin * 2; p.dy_bytes = (long)B * OH * OW * Cout * 2;
    p.msplit = plan_rows(B, OH, OW, Cin, Cout, R, groups);
    if (ws_floats < danet_conv_wgrad_rows_ws_floats(B, OH, OW, Cin, Cout, R, S, groups))
        return danet::fail(DANET_ERR_WORKSPACE, "conv_wgrad_rows: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    constexpr int CT = 4, NI = 4, SS = 7, ST = 2;
    const int nco = (p.Cout_g + CT * 16 - 1) / (CT * 16), nci = (p.Cin_g + NI * 16 - 1) / (NI * 16);
    const size_t lds = 2 * (size_t)(TH * TW * (CT * 32 + 16) + TH * (ST * (TW - 1) + SS) * (NI * 32 + 16));
    hipLaunchKernelGGL((conv_wgrad_rows_kernel<CT, NI, SS, ST>), dim3(p.msplit, R * nco * nci, groups), dim3(256), lds, st, p);
    DANET_CHECK_LAUNCH("conv_wgrad_rows_kernel");
    const long total = (long)Cout * p.Cin_g * R * S;
    hipLaunchKernelGGL(wgrad_rows_reduce_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, st, ws, dw, groups, p.Cout_g,
                       p.Cin_g, R * S, p.msplit, beta);
    DANET_CHECK_LAUNCH("wgrad_rows_reduce_kernel");
    return DANET_OK;
}
