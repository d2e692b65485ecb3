// fp32 convolutions on the matrix cores: the reference's own arithmetic type (BASELINE config C4; nn.Conv2d in fp32 all
// over /root/reference/models/module/hr_module.py:188-378, res_module.py:27-97) as a performance path.
//
// gfx950 has no reduced-precision fast path for fp32 inputs (no xf32); v_mfma_f32_16x16x4_f32 computes an exact fp32
// fmaf chain at 64 FLOP / clk / SIMD = 157 TFLOP/s, 1 / 16 of the bf16 rate.  A layer therefore has 16x more matrix
// time per byte than in bf16 and the simple structure is enough: the implicit-GEMM gather of conv_fast.hip (tap table in
// LDS, buffer resources with out-of-range offsets for padding, weight fragments packed lane-major) feeds the MFMAs from
// global memory through a two-deep register ring -- with a 64 x 64 register tile a wave loads 2 KB per 16 MFMAs of 32
// cycles each.
//   forward / data gradient (conv_f32m_kernel<MT, NT>): lane (li = lane & 15, lg = lane >> 4) loads 16 bytes = the four
//     K values 4 lg .. 4 lg + 3 of a 16-wide k-step for its pixel (B operand) and its output-channel row (A operand); the
//     i-th of four MFMAs takes element i of both, so the instruction's K index lg stands for K value 4 lg + i.
//   weight gradient (conv_f32m_wgrad_kernel<BC, NTAP>): K = pixels.  A lane loads BC consecutive output channels of dY
//     and BC consecutive input channels of X for pixel p0 + lg; MFMA (i, j) takes element i of dY and j of X and owns
//     rows {BC li + i} x columns {BC li + j}: a (16 BC) x (16 BC) block of dW per tap, NTAP taps of a filter row (or the
//     whole 3x3) accumulated side by side so that dY is loaded once for all of them.
// Activations fp32 NHWC; weights repacked (danet_conv_f32m_pack_weights) to [group][row / 16][k / 16][lane][4].
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int OOB = 0x7fffffff;

__device__ inline unsigned udiv24(unsigned n, unsigned d, float rcp) {      // n < 2^24
    unsigned q = (unsigned)((float)n * rcp);
    const int r = (int)(n - q * d);
    if (r < 0) --q; else if (r >= (int)d) ++q;
    return q;
}

struct ConvF {
    const float* x; const float* w; const float* bias; float* y;
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, dil, groups, transposed;
    int Cin_g, Cout_g, Cout_pad, K, Kp;       // Kp = roundup(K, 16)
    int relu, parity;
    long M;
    long x_bytes, y_bytes;
};

// fp32 W[Cout][Cin_g][R][S] -> [G][rows_pad / 16][Kp / 16][64 lanes][4]: lane (lg * 16 + li) of fragment (row block rb,
// k-step ks) holds row rb * 16 + li, K values ks * 16 + lg * 4 .. + 3.
// mode 0 (forward): rows = cout of the group, k = (r * S + s) * Cin_gp + cin; mode 1 (data gradient): rows = cin, k = (r * S + s) * Cout_gp + cout.
// Cin_gp / Cout_gp: the channel counts the kernel runs with (the caller zero-pads activations to multiples of 4); channels
// beyond the real Cin_g / Cout_g of w pack as zeros.
__global__ void pack_f32m_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout_g, int Cin_g, int Cout_gp, int Cin_gp, int R, int S, int G,
                                 int rows_pad, int Kp, int mode, long total)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long frag = idx >> 8;
    const int nks = Kp / 16;
    const int ks = (int)(frag % nks); frag /= nks;
    const int rbs = rows_pad / 16;
    const int rb = (int)(frag % rbs), g = (int)(frag / rbs);
    const int row = rb * 16 + (lane & 15), k = ks * 16 + (lane >> 4) * 4 + j;
    const int inner = mode == 0 ? Cin_gp : Cout_gp;
    float v = 0.f;
    if (k < R * S * inner) {
        const int tap = k / inner, ch = k - tap * inner;
        const int cout = mode == 0 ? row : ch, cin = mode == 0 ? ch : row;
        if (cout < Cout_g && cin < Cin_g) v = w[(((size_t)(g * Cout_g + cout) * Cin_g + cin) * (R * S)) + tap];
    }
    wp[idx] = v;
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_f32m_kernel(ConvF p)
{
    extern __shared__ __attribute__((aligned(16))) i32x2 sTab[];     // per 4-channel k group: {byte delta, r | s<<5 | wks<<10}
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;

    // parity classes of the strided transposed gather: an output pixel only sees taps with r = oy + pad (mod stride)
    const int nclass = p.parity ? p.stride * p.stride : 1;
    const int g = bz / nclass, cls = bz - g * nclass;
    const int py = p.parity ? cls / p.stride : 0, px = p.parity ? cls - py * p.stride : 0;
    const int step = p.parity ? p.stride : 1;
    const int OHc = (p.OH - py + step - 1) / step, OWc = (p.OW - px + step - 1) / step;
    const int Mc = p.parity ? p.B * OHc * OWc : (int)p.M;
    if (bx * (64 * MT) >= Mc) return;

    int nr = p.R, ns = p.S, r0 = 0, s0 = 0, rstep = 1, dh, orig_h = 0, orig_w = 0;
    if (p.parity) {
        r0 = (py + p.pad) % p.stride; s0 = (px + p.pad) % p.stride; rstep = p.stride;
        nr = r0 < p.R ? (p.R - r0 + p.stride - 1) / p.stride : 0;
        ns = s0 < p.S ? (p.S - s0 + p.stride - 1) / p.stride : 0;
        dh = -1;
        orig_h = (py + p.pad - r0) / p.stride; orig_w = (px + p.pad - s0) / p.stride;
    } else if (p.transposed) {
        dh = -p.dil; orig_h = p.pad; orig_w = p.pad;
    } else {
        dh = p.dil; orig_h = -p.pad; orig_w = -p.pad;
    }
    const int nks = p.parity ? nr * ns * p.Cin_g / 16 : p.Kp / 16;
    const int Kreal = p.parity ? nks * 16 : p.K;
    {
        const float rc_c = 1.0f / (float)p.Cin_g, rc_s = 1.0f / (float)(ns > 0 ? ns : 1);
        for (int e = t; e < nks * 4; e += 256) {
            const int k = e * 4;
            i32x2 v = {0, 31 | ((e >> 2) << 10)};                       // row bit 31 is never set: invalid
            if (k < Kreal) {
                const int ctap = (int)udiv24((unsigned)k, (unsigned)p.Cin_g, rc_c), cin = k - ctap * p.Cin_g;
                const int ri = (int)udiv24((unsigned)ctap, (unsigned)ns, rc_s), si = ctap - ri * ns;
                const int wk = p.parity ? (((r0 + ri * rstep) * p.S + s0 + si * rstep) * p.Cin_g + cin) >> 4 : e >> 2;
                v.x = ((ri * dh * p.W + si * dh) * p.Cin + cin) * 4;
                v.y = ri | (si << 5) | (wk << 10);
            }
            sTab[e] = v;
        }
    }
    const int m0 = bx * (64 * MT) + wave * (16 * MT);
    int pixoff[MT], outoff[MT];
    unsigned rowmask[MT], colmask[MT];
    {
        const int ohw = OHc * OWc;
        const float rc_ohw = 1.0f / (float)ohw, rc_ow = 1.0f / (float)OWc;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mreal = m0 + mt * 16 + li;
            const int m = mreal < Mc ? mreal : Mc - 1;
            const int b = (int)udiv24((unsigned)m, (unsigned)ohw, rc_ohw), rem = m - b * ohw;
            const int oh = (int)udiv24((unsigned)rem, (unsigned)OWc, rc_ow), ow = rem - oh * OWc;
            const int ph = (p.transposed ? oh : oh * p.stride) + orig_h;
            const int pw = (p.transposed ? ow : ow * p.stride) + orig_w;
            pixoff[mt] = (((b * p.H + ph) * p.W + pw) * p.Cin + g * p.Cin_g) * 4;
            unsigned rm = 0, cm = 0;
            for (int i = 0; i < nr; ++i) rm |= ((unsigned)(ph + i * dh) < (unsigned)p.H ? 1u : 0u) << i;
            for (int i = 0; i < ns; ++i) cm |= ((unsigned)(pw + i * dh) < (unsigned)p.W ? 1u : 0u) << i;
            rowmask[mt] = rm; colmask[mt] = cm;
            const int opix = p.parity ? (b * p.OH + oh * step + py) * p.OW + ow * step + px : m;
            outoff[mt] = mreal < Mc ? (opix * p.Cout + g * p.Cout_g) * 4 : OOB;
        }
    }
    __syncthreads();

    const int n0 = by * (16 * NT);
    const int nks_w = p.Kp / 16;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const float* wblk = p.w + ((size_t)g * (p.Cout_pad / 16) + n0 / 16) * (size_t)nks_w * 256;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wblk), 0, NT * nks_w * 1024, 0x00020000);
    const int wlane = lane * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_step = [&](int ks, f32x4* a, f32x4* bq) {
        const i32x2 e = sTab[ks * 4 + lg];
        const int wks = p.parity ? __builtin_amdgcn_readfirstlane(e.y >> 10) : ks;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            a[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, (nt * nks_w + wks) * 1024, 0));
        const int rb = e.y & 31, sb = (e.y >> 5) & 31;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned ok = (rowmask[mt] >> rb) & (colmask[mt] >> sb) & 1u;
            bq[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? pixoff[mt] + e.x : OOB, 0, 0));
        }
    };
    auto mma_step = [&](const f32x4* a, const f32x4* bq) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt][i], bq[mt][i], acc[mt][nt], 0, 0, 0);
    };
    constexpr int D = 2;
    f32x4 A[D][NT], Bq[D][MT];
    const int last = nks - 1;
    if (nks > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_step(min(d, last), A[d], Bq[d]);
    }
    const int nfull = nks / D;
    for (int r = 0; r < nfull; ++r) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            mma_step(A[d], Bq[d]);
            load_step(min((r + 1) * D + d, last), A[d], Bq[d]);
        }
    }
    const int rem = nks - nfull * D;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < rem) mma_step(A[d], Bq[d]);

    // epilogue: lane holds couts n0 + nt*16 + lg*4 + {0..3} of its MT pixels (Cout_g % 4 == 0, checked by the host)
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cl = n0 + nt * 16 + lg * 4;
        const bool cok = cl < p.Cout_g;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.Cout * 4, 0x00020000);
            bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(br, cok ? (g * p.Cout_g + cl) * 4 : OOB, 0, 0));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = acc[mt][nt] + bv;
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            const bool ok = cok && outoff[mt] != OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yr, ok ? outoff[mt] + cl * 4 : OOB, 0, 0);
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
struct WgF {
    const float* x; const float* dy; float* dw; float* ws;
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, dil, groups, Cin_g, Cout_g;
    int Cin_gr, Cout_gr;     // channel counts of dW itself (<= Cin_g, Cout_g: x / dy may carry zero-padded channels)
    int chunk;               // output pixels per workgroup (a multiple of 4)
    int nchunk;
    int ntg;                 // tap groups
    int ncb, nib;            // cout / cin blocks of 16 * BC channels per group
    int nblk;                // groups * ncb * nib * ntg register blocks; a workgroup's four waves take four consecutive ones
    int bc, ntap;
    long M;
    long x_bytes, y_bytes;
};

// One wave: a (16 BC) x (16 BC) block of dW for NTAP taps (tap group tg: taps tg * NTAP ...), summed over the workgroup's
// pixel chunk (the four waves of a workgroup work on four register blocks over the SAME pixels, so dY / X lines are shared
// in L1).  The partial sums go to the workspace as they lie in the registers, ws[chunk][block][tile][r][lane] (256-byte
// stores); conv_f32m_wgrad_reduce_kernel sums the chunks in a fixed order -- deterministic, no atomics.
template <int BC, int NTAP>
__global__ __launch_bounds__(256) void conv_f32m_wgrad_kernel(WgF p)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int blk = blockIdx.y * 4 + wave;
    if (blk >= p.nblk) return;
    int id = blk;
    const int tg = id % p.ntg; id /= p.ntg;
    const int ib = id % p.nib; id /= p.nib;
    const int cb = id % p.ncb, g = id / p.ncb;
    const int RS = p.R * p.S;
    const int tap0 = tg * NTAP;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.y_bytes, 0x00020000);
    const int co = cb * 16 * BC + li * BC, ci = ib * 16 * BC + li * BC;          // this lane's first cout / cin within the group
    const bool cok = co < p.Cout_g, iok = ci < p.Cin_g;                           // (channel counts are multiples of BC)
    const int dyc = (g * p.Cout_g + co) * 4, xc = (g * p.Cin_g + ci) * 4;

    f32x4 acc[NTAP][BC][BC];
#pragma unroll
    for (int tp = 0; tp < NTAP; ++tp)
#pragma unroll
        for (int i = 0; i < BC; ++i)
#pragma unroll
            for (int j = 0; j < BC; ++j) acc[tp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tr[NTAP], ts[NTAP];
#pragma unroll
    for (int tp = 0; tp < NTAP; ++tp) {
        const int tap = tap0 + tp < RS ? tap0 + tp : RS - 1;
        tr[tp] = (tap / p.S) * p.dil - p.pad; ts[tp] = (tap % p.S) * p.dil - p.pad;
    }
    const long m_begin = (long)blockIdx.x * p.chunk;
    const long m_end = min((long)p.M, m_begin + p.chunk);
    const int ohw = p.OH * p.OW;
    const float rc_ohw = 1.0f / (float)ohw, rc_ow = 1.0f / (float)p.OW;

    // (components through __int_as_float: __builtin_bit_cast applied to a vector ELEMENT expression, bit_cast(float, q[k]),
    //  read element 0 for every k with this compiler)
    struct fvec { float v[BC]; };
    auto ldc = [&](__amdgpu_buffer_rsrc_t r, int off) -> fvec {
        fvec o;
        if constexpr (BC == 4) {
            const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) o.v[k] = __int_as_float(q[k]);
        } else if constexpr (BC == 3) {
            const i32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
            const int q2 = (int)__builtin_amdgcn_raw_buffer_load_b32(r, off, 8, 0);
            o.v[0] = __int_as_float(q[0]); o.v[1] = __int_as_float(q[1]); o.v[2] = __int_as_float(q2);
        } else if constexpr (BC == 2) {
            const i32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
            o.v[0] = __int_as_float(q[0]); o.v[1] = __int_as_float(q[1]);
        } else {
            o.v[0] = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
        }
        return o;
    };
    auto load_step = [&](long mq, fvec& gv, fvec* xv) {
        const long m = mq + lg;
        const bool mok = m < m_end;
        const int mm = mok ? (int)m : 0;
        const int b = (int)udiv24((unsigned)mm, (unsigned)ohw, rc_ohw), rem = mm - b * ohw;
        const int oh = (int)udiv24((unsigned)rem, (unsigned)p.OW, rc_ow), ow = rem - oh * p.OW;
        gv = ldc(gr, (mok && cok) ? mm * p.Cout * 4 + dyc : OOB);
        const int ih0 = oh * p.stride, iw0 = ow * p.stride;
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
            const int ih = ih0 + tr[tp], iw = iw0 + ts[tp];
            const bool ok = mok && iok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && tap0 + tp < RS;
            xv[tp] = ldc(xr, ok ? ((b * p.H + ih) * p.W + iw) * p.Cin * 4 + xc : OOB);
        }
    };
    auto mma_step = [&](const fvec& gv, const fvec* xv) {
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp)
#pragma unroll
            for (int i = 0; i < BC; ++i)
#pragma unroll
                for (int j = 0; j < BC; ++j)
                    acc[tp][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.v[i], xv[tp].v[j], acc[tp][i][j], 0, 0, 0);
    };
    // two-deep register ring over the 4-pixel steps
    fvec g0, g1, x0[NTAP], x1[NTAP];
    load_step(m_begin, g0, x0);
    load_step(m_begin + 4, g1, x1);
    for (long mq = m_begin; mq < m_end; mq += 8) {
        mma_step(g0, x0);
        load_step(mq + 8, g0, x0);
        mma_step(g1, x1);                      // (steps past m_end load zeros: their MFMAs add nothing)
        load_step(mq + 12, g1, x1);
    }
    // D[row = lg * 4 + r][col = li]: row stands for cout cb*16*BC + (lg*4 + r) * BC + i, col for cin ib*16*BC + li * BC + j
    float* dst = p.ws + ((size_t)blockIdx.x * p.nblk + blk) * (size_t)(NTAP * BC * BC * 256) + lane;
#pragma unroll
    for (int tp = 0; tp < NTAP; ++tp)
#pragma unroll
        for (int i = 0; i < BC; ++i)
#pragma unroll
            for (int j = 0; j < BC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(((tp * BC + i) * BC + j) * 4 + r) * 64] = acc[tp][i][j][r];
}

// dW[g][cout][cin][tap] = sum over chunks of the partial blocks, in a fixed order.  A workgroup owns 32 consecutive
// workspace elements (coalesced 128-byte reads) x 8 chunk groups: thread (cg, e) sums the chunks cg, cg + 8, ... of its
// element, the eight partial sums meet in LDS, and the thread with cg == 0 writes the element to its place in dW.
__global__ __launch_bounds__(256) void conv_f32m_wgrad_reduce_kernel(WgF p)
{
    __shared__ float sPart[8][32];
    const int t = threadIdx.x, el = t & 31, cg = t >> 5;
    const int BC = p.bc, NTAP = p.ntap;
    const size_t bsz = (size_t)NTAP * BC * BC * 256;
    const size_t per_chunk = (size_t)p.nblk * bsz;
    const size_t e = (size_t)blockIdx.x * 32 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < per_chunk) {
        const float* src = p.ws + e;
        int c = cg;
        for (; c + 8 < p.nchunk; c += 16) { s0 += src[(size_t)c * per_chunk]; s1 += src[(size_t)(c + 8) * per_chunk]; }
        if (c < p.nchunk) s0 += src[(size_t)c * per_chunk];
    }
    sPart[cg][el] = s0 + s1;
    __syncthreads();
    if (cg != 0 || e >= per_chunk) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += sPart[k][el];
    // workspace element -> (block, tile (tp, i, j), r, lane) -> (g, cout, cin, tap)
    const int blk = (int)(e / bsz);
    int q = (int)(e - (size_t)blk * bsz);
    const int lane = q & 63; q >>= 6;
    const int r = q & 3; q >>= 2;
    const int j = q % BC; q /= BC;
    const int i = q % BC, tp = q / BC;
    int id = blk;
    const int tg = id % p.ntg; id /= p.ntg;
    const int ib = id % p.nib; id /= p.nib;
    const int cb = id % p.ncb, g = id / p.ncb;
    const int RS = p.R * p.S;
    const int tap = tg * NTAP + tp;
    const int cout = cb * 16 * BC + ((lane >> 4) * 4 + r) * BC + i, cin = ib * 16 * BC + (lane & 15) * BC + j;
    if (tap < RS && cout < p.Cout_gr && cin < p.Cin_gr)
        p.dw[((size_t)(g * p.Cout_gr + cout) * p.Cin_gr + cin) * RS + tap] = s;
}

int f32m_mt(long M, int nt, int groups) {      // pixel tiles per wave: enough workgroups for the chip, then the larger register tile
    for (int mt : {4, 2, 1}) {
        const long blocks = (M + 64 * mt - 1) / (64 * mt) * nt * groups;
        if (blocks >= 512 || mt == 1) return mt;
    }
    return 1;
}

bool fill(ConvF& p, int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups, int transposed, int relu)
{
    if (!(B > 0 && H > 0 && W > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0 && dil > 0 &&
          groups > 0 && Cin % groups == 0 && Cout % groups == 0)) return false;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.dil = dil; p.groups = groups; p.transposed = transposed;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups;
    p.K = R * S * p.Cin_g; p.Kp = (p.K + 15) / 16 * 16;
    const int nt = danet_conv_nt(p.Cout_g);
    p.Cout_pad = (p.Cout_g + 16 * nt - 1) / (16 * nt) * (16 * nt);
    p.relu = relu;
    p.M = (long)B * OH * OW;
    p.parity = (transposed && stride > 1 && dil == 1) ? 1 : 0;
    p.x_bytes = (long)B * H * W * Cin * 4;
    p.y_bytes = p.M * Cout * 4;
    return true;
}

bool supported(const ConvF& p) {
    if (p.Cin_g % 4 != 0 || p.Cout_g % 4 != 0) return false;
    if (p.transposed && p.stride > 1 && (p.Cin_g % 16 != 0 || (p.stride & (p.stride - 1)) != 0)) return false;     // parity classes: taps change at k-step boundaries
    if (p.R > 30 || p.S > 26) return false;
    if (p.M >= (1L << 24) || p.Kp >= (1 << 22)) return false;
    if (p.x_bytes >= (1L << 31) || p.y_bytes >= (1L << 31)) return false;
    if ((long)p.groups * p.Cout_pad * p.Kp * 4 >= (1L << 31)) return false;
    if ((size_t)p.Kp * 2 > 128 * 1024) return false;                  // tap table in LDS (> 64 KB: launch<> raises the kernel's dynamic-LDS limit; 2048 -> 256 / 4x4 deconv: 64 KB)
    return true;
}

template <int MT, int NT>
void launch(const ConvF& p, hipStream_t st) {
    long mblk = p.M;
    int nz = p.groups;
    if (p.parity) {
        mblk = (long)p.B * ((p.OH + p.stride - 1) / p.stride) * ((p.OW + p.stride - 1) / p.stride);
        nz *= p.stride * p.stride;
    }
    const dim3 grid((unsigned)((mblk + 64 * MT - 1) / (64 * MT)), (unsigned)(p.Cout_pad / (16 * NT)), (unsigned)nz);
    const size_t lds = (size_t)(p.Kp / 4) * sizeof(i32x2);
    if (lds > 60 * 1024) {                                   // once per instantiation: the 160 KB of a gfx950 compute unit are opt-in above 64 KB
        static bool raised = false;
        if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32m_kernel<MT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); raised = true; }
    }
    hipLaunchKernelGGL((conv_f32m_kernel<MT, NT>), grid, dim3(256), lds, st, p);
}

bool wg_plan(WgF& p, int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups,
             int Cout_real, int Cin_g_real)
{
    if (!(B > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && R > 0 && S > 0 && stride > 0 && dil > 0)) return false;
    if (!(Cout_real > 0 && Cin_g_real > 0 && Cout_real <= Cout && Cin_g_real <= Cin / groups &&
          (groups == 1 || (Cout_real == Cout && Cin_g_real == Cin / groups)))) return false;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.dil = dil; p.groups = groups;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups;
    p.Cin_gr = Cin_g_real; p.Cout_gr = Cout_real / groups;
    p.M = (long)B * OH * OW;
    p.x_bytes = (long)B * H * W * Cin * 4; p.y_bytes = p.M * Cout * 4;
    if (!(p.x_bytes < (1L << 31) && p.y_bytes < (1L << 31) && p.M < (1L << 24))) return false;
    const int RS = R * S;
    // register tile: BC channels per lane on both sides, NTAP taps side by side (BC * BC * NTAP accumulator tiles <= 36)
    const bool c4 = p.Cin_g % 4 == 0 && p.Cout_g % 4 == 0, c2 = p.Cin_g % 2 == 0 && p.Cout_g % 2 == 0;
    int BC, NTAP;
    if (RS == 9 && p.Cin_g % 48 == 0 && p.Cout_g % 48 == 0) { BC = 3; NTAP = 3; }       // HRNet-W48 widths: 48 x 48 blocks fit exactly (one filter row per wave)
    else if (RS == 9 && c2) { BC = 2; NTAP = 9; }                             // 3x3: all taps at once, dY loaded once
    else if (c4 && p.Cin_g > 32 && p.Cout_g > 32) { BC = 4; NTAP = RS >= 2 ? 2 : 1; }
    else if (c2) { BC = 2; NTAP = S >= 7 ? 7 : (RS >= 3 ? 3 : 1); }
    else { BC = 1; NTAP = RS >= 9 ? 9 : (RS >= 3 ? 3 : 1); }
    p.bc = BC; p.ntap = NTAP;
    p.ntg = (RS + NTAP - 1) / NTAP;
    p.ncb = (p.Cout_g + 16 * BC - 1) / (16 * BC); p.nib = (p.Cin_g + 16 * BC - 1) / (16 * BC);
    p.nblk = groups * p.ncb * p.nib * p.ntg;
    const long wgs_per_chunk = (p.nblk + 3) / 4;
    const size_t blk_floats = (size_t)NTAP * BC * BC * 256;
    long chunks = (512 + wgs_per_chunk - 1) / wgs_per_chunk;                  // >= 512 workgroups ...
    const long cap = (long)((size_t)(64u << 20) / (blk_floats * 4 * (size_t)p.nblk));      // ... within 64 MB of partial sums
    if (chunks > cap) chunks = cap;
    if (chunks > p.M / 32) chunks = p.M / 32;
    if (chunks < 1) chunks = 1;
    long chunk = (p.M + chunks - 1) / chunks;
    chunk = (chunk + 3) / 4 * 4;
    p.chunk = (int)chunk;
    p.nchunk = (int)((p.M + chunk - 1) / chunk);
    return true;
}

}  // namespace

// Packed-operand element count of danet_conv_f32m_pack_weights (rows padded to the kernel's channel block, K to 16).
extern "C" size_t danet_conv_f32m_packed_elems(int Cout_g, int Cin_g, int R, int S, int groups, int mode)      // padded counts
{
    const int rows = mode == 0 ? Cout_g : Cin_g, inner = mode == 0 ? Cin_g : Cout_g;
    const int nt = danet_conv_nt(rows);
    const int rows_pad = (rows + 16 * nt - 1) / (16 * nt) * (16 * nt);
    const int Kp = (R * S * inner + 15) / 16 * 16;
    return (size_t)groups * rows_pad * Kp;
}

extern "C" int danet_conv_f32m_pack_weights(const float* w, float* wp, int Cout, int Cin_g, int R, int S, int groups, int mode,
                                            int Cout_gp, int Cin_gp, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(w && wp && groups > 0 && Cout % groups == 0 && (mode == 0 || mode == 1), "conv_f32m_pack_weights: bad arguments");
    const int Cout_g = Cout / groups;
    DANET_CHECK_ARG(Cout_gp >= Cout_g && Cin_gp >= Cin_g && (groups == 1 || (Cout_gp == Cout_g && Cin_gp == Cin_g)), "conv_f32m_pack_weights: bad padded channel counts");
    const int rows = mode == 0 ? Cout_gp : Cin_gp, inner = mode == 0 ? Cin_gp : Cout_gp;
    const int nt = danet_conv_nt(rows);
    const int rows_pad = (rows + 16 * nt - 1) / (16 * nt) * (16 * nt);
    const int Kp = (R * S * inner + 15) / 16 * 16;
    const long total = (long)groups * rows_pad * Kp;
    hipLaunchKernelGGL(pack_f32m_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout_g, Cin_g, Cout_gp, Cin_gp, R, S, groups, rows_pad, Kp, mode, total);
    DANET_CHECK_LAUNCH("pack_f32m_kernel");
    return DANET_OK;
}

// 1 when danet_conv_f32m_forward takes the problem (channel counts per group multiples of 4, strided data gradients with
// 16-channel granules), 0 when the caller has to use danet_conv_f32.
extern "C" int danet_conv_f32m_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups, int transposed)
{
    ConvF p{};
    return fill(p, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, 0) && supported(p) ? 1 : 0;
}

// y[B,OH,OW,Cout] = conv(x[B,H,W,Cin], wp) (+bias)(ReLU), fp32 NHWC, wp packed by danet_conv_f32m_pack_weights (mode 0; mode 1
// with transposed = 1: the data gradient / ConvTranspose2d, (H, W, Cin) then describe the tensor gathered FROM).
extern "C" int danet_conv_f32m_forward(const float* x, const float* wp, const float* bias, float* y,
                                       int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil,
                                       int groups, int transposed, int relu, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && wp && y, "conv_f32m_forward: null pointer");
    ConvF p{};
    DANET_CHECK_ARG(fill(p, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, relu) && supported(p),
                    "conv_f32m_forward: unsupported problem (see danet_conv_f32m_ok)");
    p.x = x; p.w = wp; p.bias = bias; p.y = y;
    const int nt = danet_conv_nt(p.Cout_g);
    const long mblk = p.parity ? (long)p.B * ((p.OH + p.stride - 1) / p.stride) * ((p.OW + p.stride - 1) / p.stride) : p.M;
    const int mt = f32m_mt(mblk, p.Cout_pad / (16 * nt), p.groups * (p.parity ? p.stride * p.stride : 1));
    hipStream_t st = (hipStream_t)stream;
#define F32M_CASE(M_, N_) if (mt == M_ && nt == N_) { launch<M_, N_>(p, st); } else
    F32M_CASE(1, 1) F32M_CASE(2, 1) F32M_CASE(4, 1) F32M_CASE(1, 2) F32M_CASE(2, 2) F32M_CASE(4, 2)
    F32M_CASE(1, 3) F32M_CASE(2, 3) F32M_CASE(4, 3) F32M_CASE(1, 4) F32M_CASE(2, 4) F32M_CASE(4, 4)
    return danet::fail(DANET_ERR_ARG, "conv_f32m_forward: no kernel for tiles %dx%d", mt, nt);
#undef F32M_CASE
    DANET_CHECK_LAUNCH("conv_f32m_kernel");
    return DANET_OK;
}



// Workspace (floats) danet_conv_f32m_wgrad needs for the problem; 0 = unsupported arguments.
extern "C" size_t danet_conv_f32m_wgrad_ws_floats(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil,
                                                  int groups, int Cout_real, int Cin_g_real)
{
    WgF p{};
    if (!wg_plan(p, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, Cout_real, Cin_g_real)) return 0;
    return (size_t)p.nchunk * p.nblk * p.ntap * p.bc * p.bc * 256;
}

// dW[Cout_real][Cin_g_real][R][S] (fp32, torch layout) = sum over pixels of dY (x) X: per-pixel-chunk partial blocks into ws
// (danet_conv_f32m_wgrad_ws_floats floats), then a fixed-order sum -- deterministic.  x / dy may carry zero-padded channels
// (Cin >= groups * Cin_g_real, Cout >= Cout_real; groups == 1 then).
extern "C" int danet_conv_f32m_wgrad(const float* x, const float* dy, float* dw, float* ws,
                                     int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups,
                                     int Cout_real, int Cin_g_real, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && dy && dw && ws, "conv_f32m_wgrad: null pointer");
    WgF p{};
    DANET_CHECK_ARG(wg_plan(p, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, Cout_real, Cin_g_real), "conv_f32m_wgrad: bad arguments or tensor too large");
    p.x = x; p.dy = dy; p.dw = dw; p.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.nchunk, (unsigned)((p.nblk + 3) / 4), 1);
    const int BC = p.bc, NTAP = p.ntap;
#define WG_CASE(B_, T_) if (BC == B_ && NTAP == T_) { hipLaunchKernelGGL((conv_f32m_wgrad_kernel<B_, T_>), grid, dim3(256), 0, st, p); } else
    WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(3, 3) WG_CASE(2, 9) WG_CASE(2, 7) WG_CASE(2, 3) WG_CASE(2, 1) WG_CASE(1, 9) WG_CASE(1, 3) WG_CASE(1, 1)
    return danet::fail(DANET_ERR_ARG, "conv_f32m_wgrad: no kernel for block %d taps %d", BC, NTAP);
#undef WG_CASE
    DANET_CHECK_LAUNCH("conv_f32m_wgrad_kernel");
    const long per_chunk = (long)p.nblk * p.ntap * p.bc * p.bc * 256;
    hipLaunchKernelGGL(conv_f32m_wgrad_reduce_kernel, dim3(danet::cdiv(per_chunk, 32)), dim3(256), 0, st, p);
    DANET_CHECK_LAUNCH("conv_f32m_wgrad_reduce_kernel");
    return DANET_OK;
}
