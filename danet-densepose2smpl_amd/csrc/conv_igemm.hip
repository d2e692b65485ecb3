// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// Replaces the cuDNN/ATen convolutions behind every nn.Conv2d of the reference's hot path
// (/root/reference/models/module/hr_module.py, res_module.py; shapes in SURVEY.md A.2).
//
// Layout.  Activations are NHWC bf16 (torch channels_last), so the GEMM K axis
// k = (r, s, cin) is contiguous in memory over cin.  Weights are re-packed once per step into
// bf16 [group][Cout_pad][Kp] (K zero-padded to a multiple of 32): both MFMA operands are then
// read as 16-byte, 8-element runs of K.
//
// Operand roles.  The MFMA computes D[i][j] = sum_k A[i][k] B[k][j] with the D fragment holding
// four consecutive ROWS i per lane.  Weights are the A operand (i = cout) and pixels the B
// operand (j = pixel), so every lane ends with 4 consecutive output channels of one pixel --
// an 8-byte contiguous NHWC store, no transposition, and the bias/ReLU epilogue is per-lane.
// Any consistent assignment of the 8 per-lane K elements works because A and B use the same
// one; lane group g = lane>>4 takes k = k0 + 8g .. 8g+7.
//
// One kernel covers forward convolution (any R,S,stride,pad,dilation,groups) and, through the
// `transposed` gather (t = o + pad - r, valid when stride divides t), data gradients and
// transposed convolutions.  Weight gradients are in conv_wgrad.hip.
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

template <int MT, int NT, bool VEC8>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p)
{
    extern __shared__ __attribute__((aligned(16))) int4 sTab[];     // [Kp/8] {dh, dw, cin, valid}
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // Parity classes (transposed gather with stride > 1): an output pixel (oy, ox) only sees the taps with
    // r = (oy + pad) mod stride (same for s), so the launch is split into stride^2 classes (blockIdx.z), each
    // visiting only its own pixels and only its own taps -- no multiply-by-zero k-steps.
    const int nclass = p.parity ? p.stride * p.stride : 1;
    const int g = blockIdx.z / nclass, cls = blockIdx.z - g * nclass;
    const int py = p.parity ? cls / p.stride : 0, px = p.parity ? cls - py * p.stride : 0;
    const int step = p.parity ? p.stride : 1;
    const int OHc = (p.OH - py + step - 1) / step, OWc = (p.OW - px + step - 1) / step;   // pixels of this class
    const long Mc = p.parity ? (long)p.B * OHc * OWc : p.M;
    if ((long)blockIdx.x * (64 * MT) >= Mc) return;                                      // block-uniform
    int nks = p.Kp / 32;

    if (VEC8) {
        if (p.parity) {
            const int r0 = (py + p.pad) % p.stride, s0 = (px + p.pad) % p.stride;
            const int nr = r0 < p.R ? (p.R - r0 + p.stride - 1) / p.stride : 0;
            const int ns = s0 < p.S ? (p.S - s0 + p.stride - 1) / p.stride : 0;
            nks = nr * ns * p.Cin_g / 32;                        // host guarantees Cin_g % 32 == 0
            for (int e = t; e < nks * 4; e += 256) {
                const int k = e * 8;
                const int ctap = k / p.Cin_g, cin = k - ctap * p.Cin_g;
                const int ri = ctap / ns, si = ctap - ri * ns;
                const int r = r0 + ri * p.stride, sx = s0 + si * p.stride;
                int4 v;
                v.x = p.pad - r; v.y = p.pad - sx; v.z = cin;
                v.w = 1 + ((r * p.S + sx) * p.Cin_g + cin) / 32;  // k-step of the packed weights
                sTab[e] = v;
            }
        } else {
            for (int e = t; e < p.Kp / 8; e += 256) {
                const int k = e * 8;
                int4 v = {0, 0, 0, 0};
                if (k < p.K) {
                    const int tap = k / p.Cin_g, cin = k - tap * p.Cin_g;
                    const int r = tap / p.S, s = tap - r * p.S;
                    v.x = p.transposed ? p.pad - r * p.dil : r * p.dil - p.pad;
                    v.y = p.transposed ? p.pad - s * p.dil : s * p.dil - p.pad;
                    v.z = cin;
                    v.w = 1;
                }
                sTab[e] = v;
            }
        }
        __syncthreads();
    }

    // pixels of this wave: MT tiles of 16 consecutive output pixels (of this parity class)
    const long m0 = (long)blockIdx.x * (64 * MT) + wave * (16 * MT);
    int pb[MT], ph[MT], pw[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        long m = m0 + mt * 16 + li;
        if (m >= Mc) m = Mc - 1;
        const int ohw = OHc * OWc;
        const int b = (int)(m / ohw), rem = (int)(m - (long)b * ohw);
        const int oh = rem / OWc, ow = rem - oh * OWc;
        pb[mt] = b;
        ph[mt] = p.transposed ? oh * step + py : oh * p.stride;
        pw[mt] = p.transposed ? ow * step + px : ow * p.stride;
    }
    const int n0 = blockIdx.y * (16 * NT);
    // packed weights are stored fragment-major: [group][16-row tile][k-step][lane][8] -- the 1 KB a wave
    // loads for one (tile, k-step) is contiguous (8 fully used 128-byte lines instead of 16 half-used ones)
    const bf16_t* wbase = p.w + ((size_t)g * (p.Cout_pad / 16) + n0 / 16) * (size_t)(p.Kp / 32) * 512 + lane * 8;
    const bf16_t* xg = p.x + (size_t)g * p.Cin_g;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment loads of one k-step (weights: A operand; gathered pixels: B operand)
    auto load_step = [&](int ks, bf16x8* a, bf16x8* bq) {
        int4 e = {0, 0, 0, 0};
        int wks = ks;
        if (VEC8) {
            e = sTab[ks * 4 + lg];
            if (p.parity) wks = e.w - 1;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            a[nt] = *reinterpret_cast<const bf16x8*>(wbase + ((size_t)nt * (p.Kp / 32) + wks) * 512);
        if (VEC8) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int ih = ph[mt] + e.x, iw = pw[mt] + e.y;
                bool ok = e.w != 0;
                if (p.transposed) {        // stride is a power of two (checked by the host): mask / shift
                    ok = ok && ((ih | iw) & (p.stride - 1)) == 0;
                    ih >>= p.sshift; iw >>= p.sshift;      // arithmetic shift keeps negatives negative
                }
                ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                // unconditional load from a clamped address + select (no branch around the load)
                const bf16_t* src = ok ? xg + (((size_t)pb[mt] * p.H + ih) * p.W + iw) * p.Cin + e.z : xg;
                const uint4 v = *reinterpret_cast<const uint4*>(src);
                uint4 raw;
                raw.x = ok ? v.x : 0u; raw.y = ok ? v.y : 0u; raw.z = ok ? v.z : 0u; raw.w = ok ? v.w : 0u;
                bq[mt] = __builtin_bit_cast(bf16x8, raw);
            }
        } else {
            // channel counts that are not a multiple of 8: element-wise gather; the host pads channels to 8
            // (conv.py), so this path only serves direct C-ABI callers with odd widths
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                unsigned short v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 32 + lg * 8 + j;
                    unsigned short val = 0;
                    if (k < p.K) {
                        const int tap = k / p.Cin_g, cin = k - tap * p.Cin_g;
                        const int r = tap / p.S, s = tap - r * p.S;
                        int ih, iw;
                        bool ok;
                        if (p.transposed) {
                            ih = ph[mt] + p.pad - r * p.dil; iw = pw[mt] + p.pad - s * p.dil;
                            ok = ih >= 0 && iw >= 0 && (ih % p.stride) == 0 && (iw % p.stride) == 0;
                            ih /= p.stride; iw /= p.stride;
                            ok = ok && ih < p.H && iw < p.W;
                        } else {
                            ih = ph[mt] + r * p.dil - p.pad; iw = pw[mt] + s * p.dil - p.pad;
                            ok = ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;
                        }
                        if (ok) val = xg[(((size_t)pb[mt] * p.H + ih) * p.W + iw) * p.Cin + cin];
                    }
                    v[j] = val;
                }
                uint4 raw;
                raw.x = v[0] | ((unsigned)v[1] << 16); raw.y = v[2] | ((unsigned)v[3] << 16);
                raw.z = v[4] | ((unsigned)v[5] << 16); raw.w = v[6] | ((unsigned)v[7] << 16);
                bq[mt] = __builtin_bit_cast(bf16x8, raw);
            }
        }
    };
    auto mma_step = [&](const bf16x8* a, const bf16x8* bq) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], bq[mt], acc[mt][nt], 0, 0, 0);
    };

    // Software pipeline: a ring of D register sets keeps the fragment loads of D k-steps in flight
    // (statically indexed: the k-loop is unrolled by D).  The small-M layers (192 ch @16x16, 384 ch @8x8)
    // run one wave per SIMD with 3 MFMAs per k-step, so a single k-step ahead still exposes almost a full
    // L2 round trip per step; D = 8 / 4 / 2 for MT = 1 / 2 / 4.
    constexpr int D = MT == 1 ? 8 : (MT == 2 ? 4 : 2);
    bf16x8 A[D][NT], Bq[D][MT];
    // branch-free main loop (prefetch indices are clamped instead of guarded, so the compiler can use
    // counted vmcnt waits); the last partial round is handled after it
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

    // optional fused BatchNorm statistics of the (bf16-rounded) output: in-lane over the MT pixels, butterfly
    // over the 16 pixel lanes, LDS over the 4 waves, then one atomic per channel into replica blockIdx.x % 32
    if (p.stats) {
        __shared__ float sStat[4][2][NT * 16];
        float s1[NT][4], s2[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float v = (m0 + mt * 16 + li < p.M) ? bf2f(f2bf(acc[mt][nt][r])) : 0.f;
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
                for (int r = 0; r < 4; ++r) { sStat[wave][0][nt * 16 + lg * 4 + r] = s1[nt][r]; sStat[wave][1][nt * 16 + lg * 4 + r] = s2[nt][r]; }
        }
        __syncthreads();
        if (t < 2 * NT * 16) {
            const int which = t / (NT * 16), c = t - which * (NT * 16);
            const float v = (sStat[0][which][c] + sStat[1][which][c]) + (sStat[2][which][c] + sStat[3][which][c]);
            const int cl = n0 + c;
            if (cl < p.Cout_g)
                bn_acc_add(p.stats, blockIdx.x, which, p.Cout, g * p.Cout_g + cl, v);
        }
    }

    // epilogue: lane holds couts n0 + nt*16 + lg*4 + {0..3} of pixel m0 + mt*16 + li
    const bool vec_ok = (p.Cout % 4 == 0) && (p.Cout_g % 4 == 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long mc = m0 + mt * 16 + li;
        if (mc >= Mc) continue;
        const long m = p.parity ? ((long)pb[mt] * p.OH + ph[mt]) * p.OW + pw[mt] : mc;    // NHWC pixel index
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cl = n0 + nt * 16 + lg * 4;          // channel within the group
            if (cl >= p.Cout_g) continue;
            const int c = g * p.Cout_g + cl;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[mt][nt][r];
                if (p.bias && cl + r < p.Cout_g) x += p.bias[c + r];
                if (p.relu) x = fmaxf(x, 0.f);
                v[r] = x;
            }
            const size_t off = (size_t)m * p.Cout + c;
            if (p.out_fp32) {
                float* y = reinterpret_cast<float*>(p.y) + off;
                if (vec_ok) *reinterpret_cast<float4*>(y) = float4{v[0], v[1], v[2], v[3]};
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (cl + r < p.Cout_g) y[r] = v[r];
            } else {
                bf16_t* y = reinterpret_cast<bf16_t*>(p.y) + off;
                if (vec_ok) {
                    uint2 pk;
                    pk.x = f2bf_pk(v[0], v[1]);
                    pk.y = f2bf_pk(v[2], v[3]);
                    *reinterpret_cast<uint2*>(y) = pk;
                } else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (cl + r < p.Cout_g) y[r] = f2bf(v[r]);
            }
        }
    }
}

// Weight packing: fp32 torch layout W[Cout][Cin_g][R][S] -> bf16, logically Wp[G][rows_pad][Kp], stored
// fragment-major (see the kernel).
//  mode 0 (forward):  rows = cout within group, k = (r*S+s)*Cin_g + cin
//  mode 1 (dgrad):    rows = cin  within group, k = (r*S+s)*Cout_g + cout   (used with the transposed gather)
// With chunk > 0 (a reserved packing: the C-ABI requires chunk = 0 today) the K order is (channel chunk, tap, channel within chunk) and every chunk is
// zero-padded to KpC = roundup(R*S*chunk, 32).
__device__ inline bool pack_decode(int k, int inner, int RS, int chunk, int& tap, int& ch) {
    if (chunk <= 0) { tap = k / inner; ch = k - tap * inner; return k < RS * inner; }
    const int KpC = (RS * chunk + 31) / 32 * 32;
    const int ci = k / KpC, kk = k - ci * KpC;
    tap = kk / chunk; ch = ci * chunk + (kk - tap * chunk);
    return tap < RS;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp,
                                    int Cout_g, int Cin_g, int R, int S, int G, int rows_pad, int Kp, int mode, int chunk, int sCout_g, int sCin_g)
{
    const long total = (long)G * rows_pad * Kp;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % Kp);
    const long rest = idx / Kp;
    const int row = (int)(rest % rows_pad), g = (int)(rest / rows_pad);
    const int inner = mode == 0 ? Cin_g : Cout_g;      // channels folded into K
    const int rows = mode == 0 ? Cout_g : Cin_g;
    float v = 0.f;
    int tap, ch;
    if (pack_decode(k, inner, R * S, chunk, tap, ch) && row < rows) {
        const int r = tap / S, s = tap - r * S;
        const int cout = mode == 0 ? row : ch, cin = mode == 0 ? ch : row;
        // (sCout_g x sCin_g: the source tensor's own per-group extent -- a weight whose widths are zero-padded to multiples of 8 is packed
        //  straight from the unpadded parameter)
        if (cout < sCout_g && cin < sCin_g) v = w[(((size_t)(g * sCout_g + cout) * sCin_g + cin) * R + r) * S + s];
    }
    // fragment-major destination: [g][row/16][k/32][(k%32)/8][row%16][k%8]
    const size_t dst = (((((size_t)g * (rows_pad / 16) + row / 16) * (Kp / 32) + k / 32) * 4 + (k % 32) / 8) * 16 + row % 16) * 8 + k % 8;
    wp[dst] = f2bf(v);
}

// One launch that (re)packs a whole table of weights: element i of the concatenated index space belongs to
// the job whose [start, start + total) range holds it (binary search over the table).
struct PackJob {
    const float* w; bf16_t* wp; long start; long bstart;
    int Cout_g, Cin_g, R, S, G, rows_pad, Kp, mode, chunk, brick_ch;
    int sCout_g, sCin_g;          // per-group extent of the fp32 source (<= Cout_g, Cin_g: the rest packs as zeros)
};

// Brick path of the batched packing (jobs with brick_ch > 0: unchunked K order, channel counts that are multiples of
// 8).  A workgroup owns one brick = 16 packed rows x CH folded channels x all R*S taps of one group: it loads the
// brick's fp32 source with coalesced reads (mode 0: 16 runs of CH*RS consecutive floats, one per output channel; mode 1:
// CH runs of 16*RS floats, one per output channel), parks it in LDS and writes the bf16 fragments as 16-byte vectors
// in destination order (16 rows x 16 bytes = 256 contiguous bytes per (tap, 8-channel group)).  Every source element is
// read from memory once; the per-element gather of the kernel below reads 4-byte elements 36 (3x3) .. 2304 bytes apart.
// Padding (rows beyond the last real row tile, the K tail) is never written: the destination is zeroed once at build time.
__global__ __launch_bounds__(256) void pack_weights_brick_kernel(const PackJob* __restrict__ jobs, int njobs)
{
    extern __shared__ float sBrick[];
    __shared__ int sJob;
    const long b = blockIdx.x;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].bstart <= b) lo = mid; else hi = mid - 1;
        }
        sJob = lo;
    }
    __syncthreads();
    const PackJob j = jobs[sJob];
    const int t = threadIdx.x;
    const int CH = j.brick_ch, RS = j.R * j.S;
    const int inner = j.mode == 0 ? j.Cin_g : j.Cout_g, rows = j.mode == 0 ? j.Cout_g : j.Cin_g;
    const int nch = inner / CH, ntile = (rows + 15) / 16;
    const int bl = (int)(b - j.bstart);
    const int cidx = bl % nch, rt = (bl / nch) % ntile, g = bl / (nch * ntile);
    const int r0 = rt * 16, c0 = cidx * CH;
    const int nrow = min(16, rows - r0);
    // source runs: mode 0: run = packed row (cout), L = CH*RS;  mode 1: run = folded channel (cout), L = nrow*RS
    const int nrun = j.mode == 0 ? nrow : CH;
    const int L = j.mode == 0 ? CH * RS : nrow * RS;
    const int Lp = (j.mode == 0 ? CH * RS : 16 * RS) | 1;
    const float* src = j.mode == 0 ? j.w + ((size_t)(g * j.Cout_g + r0) * j.Cin_g + c0) * RS
                                   : j.w + ((size_t)(g * j.Cout_g + c0) * j.Cin_g + r0) * RS;
    const size_t run_stride = (size_t)j.Cin_g * RS;          // one output channel further
    // flat index over (run, position): consecutive lanes read consecutive addresses; four (vector) or eight (scalar)
    // loads are in flight per lane before the first LDS store -- one load per round trip made the launch latency-bound
    if (L % 4 == 0 && run_stride % 4 == 0 && ((src - j.w) & 3) == 0) {
        const int L4 = L / 4, tot4 = nrun * L4;
        for (int base = 0; base < tot4; base += 256 * 4) {
            float4 v[4];
            int at[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + t;
                const int run = idx / L4, i4 = idx - run * L4;
                at[u] = idx < tot4 ? run * Lp + 4 * i4 : -1;
                v[u] = idx < tot4 ? *reinterpret_cast<const float4*>(src + run * run_stride + 4 * i4) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (at[u] >= 0) { sBrick[at[u]] = v[u].x; sBrick[at[u] + 1] = v[u].y; sBrick[at[u] + 2] = v[u].z; sBrick[at[u] + 3] = v[u].w; }
        }
    } else {
        const int tot = nrun * L;
        for (int base = 0; base < tot; base += 256 * 8) {
            float v[8];
            int at[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + t;
                const int run = idx / L, i = idx - run * L;
                at[u] = idx < tot ? run * Lp + i : -1;
                v[u] = idx < tot ? src[run * run_stride + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (at[u] >= 0) sBrick[at[u]] = v[u];
        }
    }
    __syncthreads();
    const int nq = CH / 8, nvec = 16 * nq * RS;
    for (int o = t; o < nvec; o += 256) {
        const int row = o & 15, q = (o >> 4) % nq, tap = (o >> 4) / nq;
        union { bf16_t h[8]; int4 v; } out;
        if (row < nrow) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                out.h[e] = f2bf(j.mode == 0 ? sBrick[row * Lp + (8 * q + e) * RS + tap] : sBrick[(8 * q + e) * Lp + row * RS + tap]);
        } else {
            out.v = int4{0, 0, 0, 0};
        }
        const int k0 = tap * inner + c0 + 8 * q;
        const size_t dst = (((((size_t)g * (j.rows_pad / 16) + rt) * (j.Kp / 32) + k0 / 32) * 4 + (k0 % 32) / 8) * 16 + row) * 8;
        *reinterpret_cast<int4*>(j.wp + dst) = out.v;
    }
}

__global__ void pack_weights_batched_kernel(const PackJob* __restrict__ jobs, int njobs, long total_all)
{
    // each thread packs 8 consecutive k of one row (one 16-byte store); the job of the block's first
    // element is found once per block, threads past its end walk forward (jobs are >= 512 elements)
    __shared__ int sJob;
    const long base = (long)blockIdx.x * (blockDim.x * 8);
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].start <= base) lo = mid; else hi = mid - 1;
        }
        sJob = lo;
    }
    __syncthreads();
    const long gidx = base + (long)threadIdx.x * 8;
    if (gidx >= total_all) return;
    int ji = sJob;
    while (ji + 1 < njobs && jobs[ji + 1].start <= gidx) ++ji;
    const PackJob j = jobs[ji];
    const long idx = gidx - j.start;
    const int k0 = (int)(idx % j.Kp);
    const long rest = idx / j.Kp;
    const int row = (int)(rest % j.rows_pad), g = (int)(rest / j.rows_pad);
    const int inner = j.mode == 0 ? j.Cin_g : j.Cout_g;
    const int rows = j.mode == 0 ? j.Cout_g : j.Cin_g;
    const int RS = j.R * j.S;
    union { bf16_t h[8]; int4 v; } o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        int tap, ch;
        if (pack_decode(k0 + e, inner, RS, j.chunk, tap, ch) && row < rows) {
            const int cout = j.mode == 0 ? row : ch, cin = j.mode == 0 ? ch : row;
            if (cout < j.sCout_g && cin < j.sCin_g) v = j.w[((size_t)(g * j.sCout_g + cout) * j.sCin_g + cin) * RS + tap];
        }
        o.h[e] = f2bf(v);
    }
    const size_t dst = (((((size_t)g * (j.rows_pad / 16) + row / 16) * (j.Kp / 32) + k0 / 32) * 4 + (k0 % 32) / 8) * 16 + row % 16) * 8;
    *reinterpret_cast<int4*>(j.wp + dst) = o.v;
}

bool g_no_parity = getenv("DANET_CONV_NO_PARITY") != nullptr;     // debugging / A-B timing knob

template <int MT, int NT>
int launch_conv(const ConvP& p, bool vec8, hipStream_t st) {
    long mblk = p.M;
    int nz = p.groups;
    if (p.parity) {
        mblk = (long)p.B * ((p.OH + p.stride - 1) / p.stride) * ((p.OW + p.stride - 1) / p.stride);    // largest class
        nz *= p.stride * p.stride;
    }
    const dim3 grid((unsigned)((mblk + 64 * MT - 1) / (64 * MT)), (unsigned)(p.Cout_pad / (16 * NT)), (unsigned)nz);
    const size_t lds = vec8 ? (size_t)(p.Kp / 8) * sizeof(int4) : 0;
    if (vec8) hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, false>), grid, dim3(256), lds, st, p);
    return 0;
}

}  // namespace

// Output-channel tile count per block chosen for a group width: the packed weights have
// Cout_pad = roundup(Cout_g, 16*NT) rows.
extern "C" int danet_conv_nt(int Cout_g) {
    if (Cout_g <= 16) return 1;
    if (Cout_g <= 32) return 2;
    if (Cout_g % 48 == 0 || Cout_g <= 48) return 3;
    return 4;
}
// Which template instance danet_conv_forward dispatches to: MT*100 + NT*10 + vec8
// (kernel name conv_igemm_kernel<MT, NT, vec8>); used by bench.py to attribute timings.
extern "C" int danet_conv_kernel_id(int B, int OH, int OW, int Cin, int Cout, int groups) {
    const int Cout_g = Cout / groups, Cin_g = Cin / groups;
    const int nt = danet_conv_nt(Cout_g);
    const int Cout_pad = (Cout_g + 16 * nt - 1) / (16 * nt) * (16 * nt);
    const long M = (long)B * OH * OW;
    // pixel tiles per wave: the largest of 4/2/1 that still gives the 256 CUs two workgroups each
    const long nb = (long)(Cout_pad / (16 * nt)) * groups;
    const int mt = (M + 255) / 256 * nb >= 512 ? 4 : ((M + 127) / 128 * nb >= 512 ? 2 : 1);
    const int vec8 = (Cin_g % 8 == 0) && (Cin % 8 == 0);
    return mt * 100 + nt * 10 + vec8;
}

// K extent of the packed operand: unchunked roundup(R*S*inner, 32); chunked (inner/chunk) * roundup(R*S*chunk, 32)
static int packed_kp(int R, int S, int inner, int chunk) {
    if (chunk <= 0) return (R * S * inner + 31) / 32 * 32;
    return (inner / chunk) * ((R * S * chunk + 31) / 32 * 32);
}

extern "C" size_t danet_conv_packed_elems(int Cout_g, int Cin_g, int R, int S, int groups, int mode, int chunk) {
    const int rows = mode == 0 ? Cout_g : Cin_g, inner = mode == 0 ? Cin_g : Cout_g;
    const int nt = danet_conv_nt(rows);
    const int rows_pad = (rows + 16 * nt - 1) / (16 * nt) * (16 * nt);
    return (size_t)groups * rows_pad * packed_kp(R, S, inner, chunk);
}

// src_Cout_g / src_Cin_g: the per-group extent of the fp32 source tensor W[groups * src_Cout_g][src_Cin_g][R][S]; channels beyond it (up to
// Cout / groups, Cin_g) pack as zeros -- layers whose widths are no multiple of 8 run zero-padded, and their parameters keep their shapes.
extern "C" int danet_conv_pack_weights_padded(const float* w, void* wp, int Cout, int Cin_g, int R, int S, int groups,
                                              int mode, int chunk, int src_Cout_g, int src_Cin_g, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(w && wp && Cout > 0 && Cin_g > 0 && R > 0 && S > 0 && groups > 0 && Cout % groups == 0 && (mode == 0 || mode == 1),
                    "conv_pack_weights: bad arguments");
    const int Cout_g = Cout / groups;
    DANET_CHECK_ARG(src_Cout_g > 0 && src_Cout_g <= Cout_g && src_Cin_g > 0 && src_Cin_g <= Cin_g, "conv_pack_weights: source extent %d x %d exceeds %d x %d",
                    src_Cout_g, src_Cin_g, Cout_g, Cin_g);
    const int rows = mode == 0 ? Cout_g : Cin_g, inner = mode == 0 ? Cin_g : Cout_g;
    DANET_CHECK_ARG(chunk >= 0 && (chunk == 0 || inner % chunk == 0), "conv_pack_weights: chunk %d does not divide %d channels", chunk, inner);
    const int nt = danet_conv_nt(rows);
    const int rows_pad = (rows + 16 * nt - 1) / (16 * nt) * (16 * nt);
    const int Kp = packed_kp(R, S, inner, chunk);
    const long total = (long)groups * rows_pad * Kp;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (bf16_t*)wp, Cout_g, Cin_g, R, S, groups, rows_pad, Kp, mode, chunk, src_Cout_g, src_Cin_g);
    DANET_CHECK_LAUNCH("pack_weights_kernel");
    return DANET_OK;
}

extern "C" int danet_conv_pack_weights(const float* w, void* wp, int Cout, int Cin_g, int R, int S, int groups,
                                       int mode, int chunk, void* stream)
{
    return danet_conv_pack_weights_padded(w, wp, Cout, Cin_g, R, S, groups, mode, chunk, groups > 0 ? Cout / groups : 0, Cin_g, stream);
}

// Batched packing: the caller fills a host table of jobs with danet_conv_pack_job_fill (entry i at byte offset
// i * danet_conv_pack_job_bytes()), copies it to the device and (re)packs every weight with two launches.  A job
// with danet_conv_pack_job_bricks(...) > 0 belongs to the brick launch: `bstart` = running sum of the brick counts of
// the jobs before it, and it does NOT advance `start`; the others belong to the per-element launch: `start` =
// running sum of the element counts danet_conv_pack_job_fill returned for the per-element jobs before it.  The
// destination must be zeroed once (the brick launch does not write padding).
extern "C" size_t danet_conv_pack_job_bytes(void) { return sizeof(PackJob); }

// Channels per brick of the brick path (0: the job takes the per-element path): unchunked K order, folded channel
// count a multiple of 8; the largest of 256..8 that divides it and keeps the brick (16 x CH x R*S floats) under 36 KB.
static int pack_brick_ch(int inner, int RS, int chunk) {
    if (chunk != 0 || inner % 8 != 0 || getenv("DANET_NO_PACK_BRICKS")) return 0;
    for (int ch = 256; ch >= 8; ch >>= 1)
        if (inner % ch == 0 && ch * RS <= 576) return ch;
    return 0;
}

// Workgroups (bricks) the job needs in the brick launch; 0 = per-element path.
extern "C" long danet_conv_pack_job_bricks(int Cout, int Cin_g, int R, int S, int groups, int mode, int chunk)
{
    if (groups <= 0 || Cout % groups != 0) return 0;
    const int Cout_g = Cout / groups;
    const int rows = mode == 0 ? Cout_g : Cin_g, inner = mode == 0 ? Cin_g : Cout_g;
    const int ch = pack_brick_ch(inner, R * S, chunk);
    return ch ? (long)groups * ((rows + 15) / 16) * (inner / ch) : 0;
}

extern "C" long danet_conv_pack_job_fill_padded(void* job_host, const float* w, void* wp, long start, long bstart,
                                                int Cout, int Cin_g, int R, int S, int groups, int mode, int chunk, int src_Cout_g, int src_Cin_g)
{
    if (!job_host || groups <= 0 || Cout % groups != 0 || chunk < 0) return -1;
    PackJob* j = (PackJob*)job_host;
    const int Cout_g = Cout / groups;
    if (src_Cout_g <= 0 || src_Cout_g > Cout_g || src_Cin_g <= 0 || src_Cin_g > Cin_g) return -1;
    const int rows = mode == 0 ? Cout_g : Cin_g, inner = mode == 0 ? Cin_g : Cout_g;
    const int nt = danet_conv_nt(rows);
    j->w = w; j->wp = (bf16_t*)wp; j->start = start; j->bstart = bstart;
    j->brick_ch = (src_Cout_g == Cout_g && src_Cin_g == Cin_g) ? pack_brick_ch(inner, R * S, chunk) : 0;      // (the brick path reads whole runs: unpadded sources only)
    j->Cout_g = Cout_g; j->Cin_g = Cin_g; j->R = R; j->S = S; j->G = groups; j->mode = mode;
    j->sCout_g = src_Cout_g; j->sCin_g = src_Cin_g;
    j->rows_pad = (rows + 16 * nt - 1) / (16 * nt) * (16 * nt);
    if (chunk > 0 && inner % chunk != 0) return -1;
    j->chunk = chunk;
    j->Kp = packed_kp(R, S, inner, chunk);
    return (long)groups * j->rows_pad * j->Kp;
}

extern "C" long danet_conv_pack_job_fill(void* job_host, const float* w, void* wp, long start, long bstart,
                                         int Cout, int Cin_g, int R, int S, int groups, int mode, int chunk)
{
    return danet_conv_pack_job_fill_padded(job_host, w, wp, start, bstart, Cout, Cin_g, R, S, groups, mode, chunk, groups > 0 ? Cout / groups : 0, Cin_g);
}

extern "C" int danet_conv_pack_weights_batched(const void* jobs_dev, int njobs, long total_elems, long total_bricks, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(jobs_dev && njobs > 0 && total_elems >= 0 && total_bricks >= 0 && total_elems + total_bricks > 0 &&
                    total_bricks < (1L << 31), "pack_weights_batched: empty job table");
    if (total_bricks > 0) {
        hipLaunchKernelGGL(pack_weights_brick_kernel, dim3((unsigned)total_bricks), dim3(256), (16 * 576 + 256) * sizeof(float),
                           (hipStream_t)stream, (const PackJob*)jobs_dev, njobs);
        DANET_CHECK_LAUNCH("pack_weights_brick_kernel");
    }
    if (total_elems > 0) {
        hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(danet::cdiv(total_elems, 2048)), dim3(256), 0, (hipStream_t)stream,
                           (const PackJob*)jobs_dev, njobs, total_elems);
        DANET_CHECK_LAUNCH("pack_weights_batched_kernel");
    }
    return DANET_OK;
}

// Problem description shared by danet_conv_forward and danet_conv_forward_kernel.
static bool fill_conv_params(ConvP& p, bool& vec8, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                             int R, int S, int stride, int pad, int dil, int groups, int transposed, int relu, int out_fp32)
{
    if (!(B > 0 && H > 0 && W > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 &&
          pad >= 0 && dil > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && (stride & (stride - 1)) == 0))
        return false;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.dil = dil; p.groups = groups; p.transposed = transposed;
    p.Cin_g = Cin / groups; p.Cout_g = Cout / groups;
    p.K = R * S * p.Cin_g; p.Kp = (p.K + 31) / 32 * 32;
    const int nt = danet_conv_nt(p.Cout_g);
    p.Cout_pad = (p.Cout_g + 16 * nt - 1) / (16 * nt) * (16 * nt);
    p.relu = relu; p.out_fp32 = out_fp32;
    p.sshift = 0;
    while ((1 << p.sshift) < stride) ++p.sshift;
    p.M = (long)B * OH * OW;
    vec8 = (p.Cin_g % 8 == 0) && (Cin % 8 == 0);
    p.parity = (transposed && stride > 1 && dil == 1 && vec8 && p.Cin_g % 32 == 0 && !g_no_parity) ? 1 : 0;
    p.x_bytes = (long)B * H * W * Cin * 2;
    p.y_bytes = p.M * Cout * (out_fp32 ? 4 : 2);
    return true;
}

// Which kernel danet_conv_forward launches for a problem: MT*1000 + NT*100 + vec8*10 + fast
// (fast = 1: conv_fast_kernel<MT, NT>, 0: conv_igemm_kernel<MT, NT, vec8>, 2: the LDS-tile 3x3 kernels (MT, NT, KW in the other
// digits), 3: conv_pw_kernel<NKS, NTB> (NKS, NTB in the first two), 4: a grouped 3x3 layer on conv3x3_stream_kernel<NT>, 5: a narrow-group
// 3x3 layer on conv_g3_kernel<Cin_g, MT> (MT in the first digit)); -1 for invalid sizes.
extern "C" int danet_conv_forward_kernel(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S,
                                         int stride, int pad, int dil, int groups, int transposed, int out_fp32)
{
    ConvP p{};
    bool vec8;
    if (!fill_conv_params(p, vec8, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, 0, out_fp32)) return -1;
    const int mt = danet_conv_kernel_id(B, OH, OW, Cin, Cout, groups) / 100;
    if (conv_g3_ok(p, vec8)) return (p.W / 16) * 1000 + danet_conv_nt(p.Cout_g) * 100 + 10 + 5;                                   // conv_g3_kernel<Cin_g, MT>
    if (conv3x3_ok(p, vec8)) { const int c = conv3x3_config(p, vec8, 1); return (c / 100) * 1000 + ((c / 10) % 10) * 100 + (c % 10) * 10 + 2; }
    if (conv_pw_ok(p, vec8)) { const int c = conv_pw_config(p); return (c / 10) * 1000 + (c % 10) * 100 + 10 + 3; }      // conv_pw_kernel<NKS, NTB>
    if (p.groups > 1 && vec8 && conv3x3_stream_first() && conv3x3s_launch(&p, 1, nullptr, true) == 0)                       // grouped 3x3 on conv3x3_stream_kernel<NT>
        return 4000 + danet_conv_nt(p.Cout_g) * 100 + 10 + 4;
    return mt * 1000 + danet_conv_nt(p.Cout_g) * 100 + (vec8 ? 10 : 0) + (conv_fast_ok(p, vec8, mt) ? 1 : 0);
}

// y[B,OH,OW,Cout] = conv(x[B,H,W,Cin], wp) (+bias)(ReLU).  `transposed` selects the
// fractionally-strided gather (data gradient / ConvTranspose2d), in which case (H,W) is the size
// of the tensor being gathered FROM and `Cin`/`Cout` are its / the result's channel counts.
extern "C" int danet_conv_forward(const void* x, const void* wp, const float* bias, void* y,
                                  int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                  int R, int S, int stride, int pad, int dil, int groups, int transposed,
                                  int relu, int out_fp32, float* bn_sums,
                                  const void* bn_x, const void* bn_y, const float* bn_saved, float* bn_red, const void* addend,
                                  int bn_gate, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && wp && y, "conv_forward: null pointer");
    ConvP p{};
    bool vec8;
    DANET_CHECK_ARG(fill_conv_params(p, vec8, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, relu, out_fp32),
                    "conv_forward: bad sizes B=%d H=%d W=%d Cin=%d OH=%d OW=%d Cout=%d R=%d S=%d stride=%d (power of two) pad=%d groups=%d",
                    B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, groups);
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wp; p.bias = bias; p.y = y; p.stats = bn_sums;
    p.bn_x = (const bf16_t*)bn_x; p.bn_y = (const bf16_t*)bn_y; p.bn_saved = bn_saved; p.bn_red = bn_red; p.bn_gate = bn_gate;
    p.addend = (const bf16_t*)addend;
    const int mt = danet_conv_kernel_id(B, OH, OW, Cin, Cout, groups) / 100;
    const bool c3 = conv3x3_ok(p, vec8);
    DANET_CHECK_ARG(bn_gate == 0 || (bn_red && c3 && bn_gate == 2 && bn_y),
                    "conv_forward: bn_gate %d: only 2 (byte mask in bn_y, LDS-tile 3x3 kernel) is defined", bn_gate);
    const bool pw = !c3 && conv_pw_ok(p, vec8);
    DANET_CHECK_ARG(!addend || (!out_fp32 && (c3 || pw || (conv_fast_ok(p, vec8, mt) && !p.stats && !p.bn_red))),
                    "conv_forward: the fused addend needs a bf16 output and the 3x3 LDS kernels, or the lean gather kernel without fused statistics (check danet_conv_forward_kernel)");
    DANET_CHECK_ARG(!bn_red || (bn_x && bn_saved && !bias && !relu && !out_fp32 && (c3 || conv_fast_ok(p, vec8, mt))),
                    "conv_forward: the fused BatchNorm-backward reduction needs the fast kernel and a plain bf16 output (check danet_conv_forward_kernel)");
    const int nt = danet_conv_nt(p.Cout_g);
    DANET_CHECK_ARG(!bn_sums || (!bias && !relu && !out_fp32), "conv_forward: fused BN statistics need a plain bf16 output");
    DANET_CHECK_ARG((size_t)(p.Kp / 8) * 16 <= 64 * 1024, "conv_forward: K=%d too large for the tap table", p.K);
    hipStream_t st = (hipStream_t)stream;
    if (conv_g3_ok(p, vec8)) {
        // the 24-group partial-IUV head and its data gradient (csrc/conv_g3.hip): one (image, group, row band) per workgroup
        if (conv_g3_launch(p, stream) != 0) return danet::fail(DANET_ERR_ARG, "conv_forward: no narrow-group instantiation");
        DANET_CHECK_LAUNCH("conv_g3_kernel");
        return DANET_OK;
    }
    if (c3) {
        if (conv3x3_launch(&p, 1, stream) != 0) return danet::fail(DANET_ERR_ARG, "conv_forward: no 3x3 tiling");
        DANET_CHECK_LAUNCH("conv3x3_tile_kernel");
        return DANET_OK;
    }
    if (pw) {
        if (conv_pw_launch(p, stream) != 0) return danet::fail(DANET_ERR_ARG, "conv_forward: no pointwise instantiation");
        DANET_CHECK_LAUNCH("conv_pw_kernel");
        return DANET_OK;
    }
    if (p.groups > 1 && vec8 && !p.bn_red && conv3x3_stream_first() && conv3x3s_launch(&p, 1, stream, false) == 0) {
        // grouped 3x3 / stride-1 layers (the 24-group partial-IUV head): one group's channels of a pixel tile per tile of the streamed kernel
        DANET_CHECK_LAUNCH("conv3x3_stream_kernel");
        return DANET_OK;
    }
    if (conv_fast_ok(p, vec8, mt)) {
        if (conv_fast_launch(p, mt, nt, stream) != 0) return danet::fail(DANET_ERR_ARG, "conv_forward: no fast kernel for tiles %dx%d", mt, nt);
        DANET_CHECK_LAUNCH("conv_fast_kernel");
        return DANET_OK;
    }
#define CONV_CASE(M_, N_) if (mt == M_ && nt == N_) launch_conv<M_, N_>(p, vec8, st); else
    CONV_CASE(1, 1) CONV_CASE(2, 1) CONV_CASE(4, 1) CONV_CASE(1, 2) CONV_CASE(2, 2) CONV_CASE(4, 2)
    CONV_CASE(1, 3) CONV_CASE(2, 3) CONV_CASE(4, 3) CONV_CASE(1, 4) CONV_CASE(2, 4) CONV_CASE(4, 4)
    return danet::fail(DANET_ERR_ARG, "conv_forward: no kernel for tiles %dx%d", mt, nt);
#undef CONV_CASE
    DANET_CHECK_LAUNCH("conv_igemm_kernel");
    return DANET_OK;
}


// ---------------------------------------------------------------------------------------------
// Up to 4 independent convolutions in one launch (all on the fast kernel, same danet_conv_nt of their Cout_g).
// job = { x, wp, y, bn_sums, bn_x, bn_y, bn_saved, bn_red, addend; int B,H,W,Cin,OH,OW,Cout,R,S,stride,pad,dil,groups,transposed,bn_gate }
// danet_conv_forward_multi_ok says whether a set qualifies (then the call cannot fail for shape reasons).
struct ConvJob { const void* x; const void* wp; void* y; float* bn_sums; const void* bn_x; const void* bn_y; const float* bn_saved; float* bn_red; const void* addend;
                 int B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, bn_gate; };

// *all3 = every problem runs on the LDS-tile 3x3 kernel (then nt / mts are not needed)
static int conv_multi_prepare(const ConvJob* jobs, int n, ConvP* ps, int* mts, int* nt_out, bool* all3) {
    if (!jobs || n < 1 || n > 12) return -1;
    int nt = -1;
    *all3 = n <= 4;                         // (the LDS-tile 3x3 kernel takes up to 4 problems, the gather kernel up to 12)
    for (int i = 0; i < n && *all3; ++i) {
        const ConvJob& j = jobs[i];
        bool vec8;
        ps[i] = ConvP{};
        if (!fill_conv_params(ps[i], vec8, j.B, j.H, j.W, j.Cin, j.OH, j.OW, j.Cout, j.R, j.S, j.stride, j.pad, j.dil, j.groups, j.transposed, 0, 0)) return -1;
        if (!conv3x3_ok(ps[i], vec8)) { *all3 = false; break; }
        ps[i].x = (const bf16_t*)j.x; ps[i].w = (const bf16_t*)j.wp; ps[i].bias = nullptr; ps[i].y = j.y; ps[i].stats = j.bn_sums;
        ps[i].bn_x = (const bf16_t*)j.bn_x; ps[i].bn_y = (const bf16_t*)j.bn_y; ps[i].bn_saved = j.bn_saved; ps[i].bn_red = j.bn_red;
        ps[i].addend = (const bf16_t*)j.addend; ps[i].bn_gate = j.bn_gate;
    }
    if (*all3 && conv3x3_launch(ps, n, nullptr, true) != 0) *all3 = false;      // e.g. a tiling the multi-problem kernel lacks
    if (*all3) { *nt_out = 0; return 0; }
    for (int i = 0; i < n; ++i) if (jobs[i].addend || jobs[i].bn_gate) return -1;      // the gather kernel has no fused addend / mask gate
    for (int i = 0; i < n; ++i) {
        const ConvJob& j = jobs[i];
        bool vec8;
        ps[i] = ConvP{};
        if (!fill_conv_params(ps[i], vec8, j.B, j.H, j.W, j.Cin, j.OH, j.OW, j.Cout, j.R, j.S, j.stride, j.pad, j.dil, j.groups, j.transposed, 0, 0)) return -1;
        const int nti = danet_conv_nt(ps[i].Cout_g);
        if (nt < 0) nt = nti; else if (nt != nti) return -1;
        mts[i] = danet_conv_kernel_id(j.B, j.OH, j.OW, j.Cin, j.Cout, j.groups) / 100;
        if (!conv_fast_ok(ps[i], vec8, mts[i])) return -1;
        ps[i].x = (const bf16_t*)j.x; ps[i].w = (const bf16_t*)j.wp; ps[i].bias = nullptr; ps[i].y = j.y; ps[i].stats = j.bn_sums;
        ps[i].bn_x = (const bf16_t*)j.bn_x; ps[i].bn_y = (const bf16_t*)j.bn_y; ps[i].bn_saved = j.bn_saved; ps[i].bn_red = j.bn_red;
        ps[i].addend = (const bf16_t*)j.addend;
    }
    *nt_out = nt;
    return 0;
}

extern "C" int danet_conv_forward_multi_ok(const void* jobs, int n)
{
    ConvP ps[12]; int mts[12], nt; bool all3;
    if (conv_multi_prepare((const ConvJob*)jobs, n, ps, mts, &nt, &all3) != 0) return 0;
    return all3 ? 2 : 1;                  // 2: one conv3x3_tile_kernel launch, 1: one conv_fast_multi_kernel launch
}

// Which kernel danet_conv_forward_multi launches for the set: 0 none (unsupported), 1 conv_fast_multi_kernel, 2 conv3x3_tile_kernel,
// 3 conv3x3_stream_kernel (profilers label their records with it).
extern "C" int danet_conv_forward_multi_kernel(const void* jobs, int n)
{
    ConvP ps[12]; int mts[12], nt; bool all3;
    if (conv_multi_prepare((const ConvJob*)jobs, n, ps, mts, &nt, &all3) != 0) return 0;
    if (!all3) return 1;
    return conv3x3s_launch(ps, n, nullptr, true) == 0 && conv3x3_stream_first() ? 3 : 2;
}

// ---- convolutions + the training-mode BatchNorms that follow them, one launch when the streamed 3x3 kernel takes the set ----------
// (conv3x3s.hip s3_bn_tail; /root/reference/models/module/res_module.py:39-56, hr_module.py:155-177.)  bn jobs: the forward jobs of
// danet_bn_forward_multi (norm_act.hip), job i normalising convolution i's output -- x = that output, sums = its bn_sums, sums_state 2.
struct BnFwdJobC { const void* x; const void* res; void* y; const float* gamma; const float* beta; float* running_mean; float* running_var;
                   float* saved; float* sums; void* mask; int64_t M; int C, sums_state, relu; };

// >= 0: the number of problems when the set runs as ONE launch (convolutions + BatchNorm tail), 0 when it does not qualify
static int conv_bn_fusable(const ConvJob* jobs, int n, const BnFwdJobC* bn, ConvP* ps, BnApply* ba, float momentum, float eps, void* bar)
{
    static const bool off = getenv("DANET_NO_CONV_BN") != nullptr;          // A-B timing knob
    if (off || !bar || !bn || n < 1 || n > 4 || !conv3x3_stream_first()) return 0;
    int mts[12], nt; bool all3;
    if (conv_multi_prepare(jobs, n, ps, mts, &nt, &all3) != 0 || !all3) return 0;
    for (int i = 0; i < n; ++i) {
        const BnFwdJobC& b = bn[i];
        if (!ps[i].stats || jobs[i].transposed || jobs[i].bn_red || jobs[i].addend) return 0;
        if (b.x != jobs[i].y || b.sums != jobs[i].bn_sums || b.sums_state != 2 || b.C != jobs[i].Cout) return 0;
        if (b.M != (int64_t)jobs[i].B * jobs[i].OH * jobs[i].OW || !b.y || !b.saved || !b.gamma || !b.beta) return 0;
        ba[i] = BnApply{(const bf16_t*)b.res, (bf16_t*)b.y, b.gamma, b.beta, b.running_mean, b.running_var, b.saved, (unsigned char*)b.mask, b.relu,
                        momentum, eps, (unsigned*)bar};
        ps[i].bna = &ba[i];
    }
    return conv3x3s_launch(ps, n, nullptr, true) == 0 ? n : 0;
}

extern "C" int danet_conv_bn_forward_multi_ok(const void* jobs, int n, const void* bn_jobs, void* bar)
{
    ConvP ps[12]; BnApply ba[4];
    return conv_bn_fusable((const ConvJob*)jobs, n, (const BnFwdJobC*)bn_jobs, ps, ba, 0.1f, 1e-5f, bar) > 0 ? 1 : 0;
}

extern "C" int danet_conv_bn_forward_multi(const void* jobs, int n, const void* bn_jobs, float momentum, float eps, void* bar, int* fused, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(jobs && bn_jobs && n >= 1 && n <= 12, "conv_bn_forward_multi: 1..12 jobs with their BatchNorm jobs");
    ConvP ps[12]; BnApply ba[4];
    if (fused) *fused = 0;
    if (conv_bn_fusable((const ConvJob*)jobs, n, (const BnFwdJobC*)bn_jobs, ps, ba, momentum, eps, bar) > 0) {
        for (int i = 0; i < n; ++i) DANET_CHECK_ARG(ps[i].x && ps[i].w && ps[i].y, "conv_bn_forward_multi: job %d: null pointer", i);
        if (conv3x3s_launch(ps, n, stream, false) == 0) {              // (-1: e.g. a tap table that is new while the stream captures)
            DANET_CHECK_LAUNCH("conv3x3_stream_bn_kernel");
            if (fused) *fused = 1;
            return DANET_OK;
        }
    }
    // two launches: the set as danet_conv_forward_multi runs it, then danet_bn_forward_multi
    const int rc = danet_conv_forward_multi(jobs, n, stream);
    if (rc != DANET_OK) return rc;
    return danet_bn_forward_multi(bn_jobs, n, momentum, eps, stream);
}

extern "C" int danet_conv_forward_multi(const void* jobs, int n, void* stream)
{
    DANET_ENTER();
    ConvP ps[12]; int mts[12], nt; bool all3;
    DANET_CHECK_ARG(conv_multi_prepare((const ConvJob*)jobs, n, ps, mts, &nt, &all3) == 0, "conv_forward_multi: unsupported set (see danet_conv_forward_multi_ok)");
    for (int i = 0; i < n; ++i) {
        DANET_CHECK_ARG(ps[i].x && ps[i].w && ps[i].y, "conv_forward_multi: job %d: null pointer", i);
        DANET_CHECK_ARG(!ps[i].bn_red || (ps[i].bn_x && ps[i].bn_saved), "conv_forward_multi: job %d: incomplete BatchNorm-backward arguments", i);
    }
    if (all3) {
        DANET_CHECK_ARG(conv3x3_launch(ps, n, stream) == 0, "conv_forward_multi: no 3x3 tiling");
        DANET_CHECK_LAUNCH("conv3x3_tile_kernel");
        return DANET_OK;
    }
    DANET_CHECK_ARG(conv_fast_launch_multi(ps, mts, n, nt, stream) == 0, "conv_forward_multi: no kernel for %d tiles per block", nt);
    DANET_CHECK_LAUNCH("conv_fast_multi_kernel");
    return DANET_OK;
}
