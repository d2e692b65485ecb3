// Lean implicit-GEMM convolution: the same algorithm and operand layout as conv_igemm.hip, written for
// instruction economy.  PMC counters on MI355X showed the generic kernel spending ~23 VALU instructions
// per MFMA (64-bit address arithmetic, per-gather bounds logic, integer divisions in the prologue, a
// software bf16 rounding in the epilogue): a 48-channel 3x3 layer issued ~3900 VALU + 1400 SALU
// instructions per wave for 168 MFMAs and was bound by instruction issue, not by memory or the matrix
// cores.  Here
//   * every tensor is addressed through a buffer resource with a 32-bit byte offset; invalid gathers
//     (padding, ragged tiles) get an out-of-range offset and the hardware returns zeros / drops the store,
//   * a gather's offset is  pixel_base + tap_delta  (one add): tap deltas come from the LDS table, validity
//     from per-pixel row / column bit masks built once in the prologue (two shifts, two ands),
//   * weight fragment loads take the k-step as a scalar offset (no VALU),
//   * divisions use a float reciprocal with a fix-up (operands < 2^24, checked by the host),
//   * the epilogue converts with v_cvt_pk_bf16_f32 and the BatchNorm statistics reduce with DPP adds.
// Handles: 8-channel granules (Cin % 8 == 0, Cin_g % 8 == 0), forward with any stride / dilation,
// transposed gather with stride 1 or in parity-class mode; everything else stays on conv_igemm_kernel.
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

__device__ inline unsigned udiv24(unsigned n, unsigned d, float rcp) {      // n < 2^24
    unsigned q = (unsigned)((float)n * rcp);
    const int r = (int)(n - q * d);
    if (r < 0) --q; else if (r >= (int)d) ++q;
    return q;
}

template <int CTRL>
__device__ inline float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, o);
}
// sum over the 16 lanes of a DPP row, result in every lane (xor 1, xor 2, half-row mirror, row mirror)
__device__ inline float row_sum16(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    return v;
}

constexpr int OOB = 0x7fffffff;      // beyond every buffer (sizes are checked to be < 2^31 bytes)

// Per-lane partial sums s1/s2[nt][r] of channels n0 + nt*16 + lg*4 + r (over the lane's pixels) -> summed over
// the 16 pixel lanes (DPP), over the 4 waves (LDS), then one atomic per channel into replica blockIdx.x % 32 of
// dst [32][2][Ctot].
template <int NT>
__device__ inline void channel_sums_to_replicas(float (*s1)[4], float (*s2)[4], float* __restrict__ dst, int Ctot, int cbase,
                                                int n0, int Cg, int t, int li, int lg, int wave, int rep)
{
    __shared__ float sStat[4][2][NT * 16];
    __syncthreads();                                   // a previous use of sStat by this block is finished
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = row_sum16(s1[nt][r]), b = row_sum16(s2[nt][r]);
            if (li == 0) { sStat[wave][0][nt * 16 + lg * 4 + r] = a; sStat[wave][1][nt * 16 + lg * 4 + r] = b; }
        }
    __syncthreads();
    if (t < 2 * NT * 16) {
        const int which = t / (NT * 16), c = t - which * (NT * 16);
        const float v = (sStat[0][which][c] + sStat[1][which][c]) + (sStat[2][which][c] + sStat[3][which][c]);
        const int cl = n0 + c;
        if (cl < Cg) bn_acc_add(dst, rep, which, Ctot, cbase + cl, v);
    }
}

template <int MT, int NT>
__device__ __forceinline__ void conv_fast_body(const ConvP& p, const int bx_, const int by_, const int bz_)
{
    extern __shared__ __attribute__((aligned(16))) i32x2 sTab[];     // per 8-channel k group: {byte delta, r | s<<5 | wks<<10}
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;

    // parity classes of the strided transposed gather (see conv_igemm.hip)
    const int nclass = p.parity ? p.stride * p.stride : 1;
    const int g = bz_ / nclass, cls = bz_ - g * nclass;
    const int py = p.parity ? cls / p.stride : 0, px = p.parity ? cls - py * p.stride : 0;
    const int step = p.parity ? p.stride : 1;
    const int OHc = (p.OH - py + step - 1) / step, OWc = (p.OW - px + step - 1) / step;
    const int Mc = p.parity ? p.B * OHc * OWc : (int)p.M;
    // (an XCD-contiguous tile order -- every XCD's L2 serving one contiguous pixel range -- was measured and did
    // not help: the 256 MB infinity cache already absorbs the cross-XCD halo re-reads)
    const int bx = bx_;
    if (bx * (64 * MT) >= Mc) return;

    // tap geometry: tap row index i (0 .. nr-1) reads input row  ph + i*dh ; likewise columns
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
    const int nks = p.parity ? nr * ns * p.Cin_g / 32 : p.Kp / 32;
    const int Kreal = p.parity ? nks * 32 : p.K;

    {
        const float rc_c = 1.0f / (float)p.Cin_g, rc_s = 1.0f / (float)(ns > 0 ? ns : 1);
        for (int e = t; e < nks * 4; e += 256) {
            const int k = e * 8;
            i32x2 v = {0, 31};                                          // row bit 31 is never set: invalid
            if (k < Kreal) {
                const int ctap = (int)udiv24((unsigned)k, (unsigned)p.Cin_g, rc_c), cin = k - ctap * p.Cin_g;
                const int ri = (int)udiv24((unsigned)ctap, (unsigned)ns, rc_s), si = ctap - ri * ns;
                const int wk = p.parity ? (((r0 + ri * rstep) * p.S + s0 + si * rstep) * p.Cin_g + cin) >> 5 : e >> 2;
                v.x = ((ri * dh * p.W + si * dh) * p.Cin + cin) * 2;
                v.y = ri | (si << 5) | (wk << 10);
            } else {
                v.y = 31 | ((e >> 2) << 10);
            }
            sTab[e] = v;
        }
    }

    // pixels of this wave
    const int m0 = bx * (64 * MT) + wave * (16 * MT);
    int pixoff[MT], outoff[MT];
    unsigned rowmask[MT], colmask[MT];
    {
        const int ohw = OHc * OWc;
        const float rc_ohw = 1.0f / (float)ohw, rc_ow = 1.0f / (float)OWc;
        const int osz = p.out_fp32 ? 4 : 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mreal = m0 + mt * 16 + li;
            const int m = mreal < Mc ? mreal : Mc - 1;
            const int b = (int)udiv24((unsigned)m, (unsigned)ohw, rc_ohw), rem = m - b * ohw;
            const int oh = (int)udiv24((unsigned)rem, (unsigned)OWc, rc_ow), ow = rem - oh * OWc;
            const int ph = (p.transposed ? oh : oh * p.stride) + orig_h;
            const int pw = (p.transposed ? ow : ow * p.stride) + orig_w;
            pixoff[mt] = (((b * p.H + ph) * p.W + pw) * p.Cin + g * p.Cin_g) * 2;
            unsigned rm = 0, cm = 0;
            for (int i = 0; i < nr; ++i) rm |= ((unsigned)(ph + i * dh) < (unsigned)p.H ? 1u : 0u) << i;
            for (int i = 0; i < ns; ++i) cm |= ((unsigned)(pw + i * dh) < (unsigned)p.W ? 1u : 0u) << i;
            rowmask[mt] = rm; colmask[mt] = cm;
            const int opix = p.parity ? (b * p.OH + oh * step + py) * p.OW + ow * step + px : m;
            outoff[mt] = mreal < Mc ? (opix * p.Cout + g * p.Cout_g) * osz : OOB;
        }
    }
    __syncthreads();

    const int n0 = by_ * (16 * NT);
    const int nks_w = p.Kp / 32;                                       // k-steps of the packed weights
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const bf16_t* wblk = p.w + ((size_t)g * (p.Cout_pad / 16) + n0 / 16) * (size_t)nks_w * 512;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wblk), 0, NT * nks_w * 1024, 0x00020000);
    const int wlane = lane * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_step = [&](int ks, bf16x8* a, bf16x8* bq) {
        const i32x2 e = sTab[ks * 4 + lg];
        const int wks = p.parity ? __builtin_amdgcn_readfirstlane(e.y >> 10) : ks;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            a[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, (nt * nks_w + wks) * 1024, 0));
        const int rb = e.y & 31, sb = (e.y >> 5) & 31;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned ok = (rowmask[mt] >> rb) & (colmask[mt] >> sb) & 1u;
            const int off = ok ? pixoff[mt] + e.x : OOB;
            bq[mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
    };
    auto mma_step = [&](const bf16x8* a, const bf16x8* bq) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], bq[mt], acc[mt][nt], 0, 0, 0);
    };

    if constexpr (MT == 1) {
        // Small-M layers (one pixel tile per wave: 192 ch @16x16, 384 ch @8x8, the regressor tails) are bound by
        // streaming the weights: with private fragment loads every wave of a block pulls the block's whole
        // (48..64 couts x K) weight slab through L1 -- 4x redundantly.  Here the four waves SHARE the A operand:
        // per round of G k-steps each wave loads two k-steps' fragments, parks them in LDS (double-buffered), and all
        // waves read every fragment from there (ds_read_b128, lane-linear, conflict-free).  The gathered B
        // operands of the next round are loaded into a second register set while the current round's MFMAs run.
        constexpr int G = 8;
        unsigned char* const sA = reinterpret_cast<unsigned char*>(sTab) + p.Kp;      // [2][G][NT] KB behind the tap table
        const int nrounds = (nks + G - 1) / G;
        const int last = nks > 0 ? nks - 1 : 0;
        bf16x8 Areg[2][NT], Bq2[2][G];
        auto load_round = [&](int r, bf16x8* bq) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ks = __builtin_amdgcn_readfirstlane(r * G + 2 * wave + j);
                const int kc = min(ks, last);
                const int wks = p.parity ? __builtin_amdgcn_readfirstlane(sTab[kc * 4].y >> 10) : kc;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    Areg[j][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, ks < nks ? wlane : OOB, (nt * nks_w + wks) * 1024, 0));
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int ks = r * G + q;
                const i32x2 e = sTab[min(ks, last) * 4 + lg];
                const unsigned ok = (rowmask[0] >> (e.y & 31)) & (colmask[0] >> ((e.y >> 5) & 31)) & 1u;
                bq[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, (ok && ks < nks) ? pixoff[0] + e.x : OOB, 0, 0));
            }
        };
        auto round_body = [&](int r, bf16x8* bcur, bf16x8* bnxt) {
            unsigned char* const buf = sA + (size_t)(r & 1) * (G * NT * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    *reinterpret_cast<bf16x8*>(buf + ((2 * wave + j) * NT + nt) * 1024 + lane * 16) = Areg[j][nt];
            __syncthreads();                  // round r's fragments visible; everyone is done with round r-1 (other buffer)
            if (r + 1 < nrounds) load_round(r + 1, bnxt);
            // the weight fragments of k-step q + 1 are read from LDS before the MFMAs of k-step q (two register sets; left to the
            // compiler every MFMA waited for a ds_read issued one MFMA earlier: lgkmcnt(1) in front of each of them)
            bf16x8 a[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[0][nt] = *reinterpret_cast<const bf16x8*>(buf + nt * 1024 + lane * 16);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                if (q + 1 < G) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        a[(q + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(buf + ((q + 1) * NT + nt) * 1024 + lane * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q & 1][nt], bcur[q], acc[0][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (nrounds > 0) load_round(0, Bq2[0]);
        for (int r = 0; r < nrounds; r += 2) {
            round_body(r, Bq2[0], Bq2[1]);
            if (r + 1 < nrounds) round_body(r + 1, Bq2[1], Bq2[0]);
        }
    } else {
    // register ring of D k-steps in flight (see conv_igemm.hip)
    constexpr int D = MT == 2 ? 4 : 2;
    bf16x8 A[D][NT], Bq[D][MT];
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
    }

    // fused BatchNorm statistics of the bf16-rounded output (forward) ...
    if (p.stats) {
        float s1[NT][4], s2[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float v = outoff[mt] != OOB ? bf2f(f2bf(acc[mt][nt][r])) : 0.f;
                    a += v; b += v * v;
                }
                s1[nt][r] = a; s2[nt][r] = b;
            }
        channel_sums_to_replicas<NT>(s1, s2, p.stats, p.Cout, g * p.Cout_g, n0, p.Cout_g, t, li, lg, wave, bx);
    }
    // ... or (data gradient) the two BatchNorm-backward sums of the BN that produced this conv's input
    if (p.bn_red) {
        const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_x), 0, (int)p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_y ? p.bn_y : p.bn_x), 0, (int)p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t svr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bn_saved), 0, p.Cout * 8, 0x00020000);
        float s1[NT][4], s2[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cl = n0 + nt * 16 + lg * 4;
            const int c = g * p.Cout_g + cl;
            const bool cok = cl < p.Cout_g;
            const f32x4 mean = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(svr, cok ? c * 4 : OOB, 0, 0));
            const f32x4 invs = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(svr, cok ? (p.Cout + c) * 4 : OOB, 0, 0));
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int off = (cok && outoff[mt] != OOB) ? outoff[mt] + cl * 2 : OOB;
                const i32x2 xq = __builtin_amdgcn_raw_buffer_load_b64(bxr, off, 0, 0);
                i32x2 yq = {0x3f803f80, 0x3f803f80};                               // "positive" when there is no ReLU
                if (p.bn_y) yq = __builtin_amdgcn_raw_buffer_load_b64(byr, off, 0, 0);
                const float xv[4] = {__uint_as_float((unsigned)xq.x << 16), __uint_as_float((unsigned)xq.x & 0xffff0000u),
                                     __uint_as_float((unsigned)xq.y << 16), __uint_as_float((unsigned)xq.y & 0xffff0000u)};
                const float yv[4] = {__uint_as_float((unsigned)yq.x << 16), __uint_as_float((unsigned)yq.x & 0xffff0000u),
                                     __uint_as_float((unsigned)yq.y << 16), __uint_as_float((unsigned)yq.y & 0xffff0000u)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gv = (off != OOB && yv[r] > 0.f) ? bf2f(f2bf(acc[mt][nt][r])) : 0.f;
                    s1[nt][r] += gv; s2[nt][r] += gv * (xv[r] - mean[r]) * invs[r];
                }
            }
        }
        channel_sums_to_replicas<NT>(s1, s2, p.bn_red, p.Cout, g * p.Cout_g, n0, p.Cout_g, t, li, lg, wave, bx);
    }

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
            const bool ok = cok && outoff[mt] != OOB;
            if (p.addend) {               // (bf16 outputs only, checked by the host) the residual branch's gradient, added before rounding
                const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.addend), 0, (int)p.y_bytes, 0x00020000);
                const i32x2 aq = __builtin_amdgcn_raw_buffer_load_b64(ar, ok ? outoff[mt] + cl * 2 : OOB, 0, 0);
                v[0] += __uint_as_float((unsigned)aq.x << 16); v[1] += __uint_as_float((unsigned)aq.x & 0xffff0000u);
                v[2] += __uint_as_float((unsigned)aq.y << 16); v[3] += __uint_as_float((unsigned)aq.y & 0xffff0000u);
            }
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (p.out_fp32) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yr, ok ? outoff[mt] + cl * 4 : OOB, 0, 0);
            } else {
                i32x2 pk = {(int)f2bf_pk(v[0], v[1]), (int)f2bf_pk(v[2], v[3])};
                __builtin_amdgcn_raw_buffer_store_b64(pk, yr, ok ? outoff[mt] + cl * 2 : OOB, 0, 0);
            }
        }
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_fast_kernel(ConvP p)
{
    conv_fast_body<MT, NT>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Up to 4 independent convolutions with the same NT in one launch (HRNet branches in lockstep): workgroup id ->
// problem through the prefix table, then the problem's own (MT, grid).
constexpr int NCM = 12;     // problems per multi launch (kernel arguments: 12 x ~300 B; 12 = the exchange paths of a four-branch module's first stage)
struct ConvMulti { ConvP p[NCM]; int start[NCM + 1]; int gx[NCM], gy[NCM], mt[NCM]; int n; };
static_assert(sizeof(ConvMulti) <= 4096, "kernel arguments");

template <int NT>
__global__ __launch_bounds__(256) void conv_fast_multi_kernel(ConvMulti m)
{
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const int l = blockIdx.x - m.start[i];
    const int bx = l % m.gx[i], rest = l / m.gx[i];
    const int by = rest % m.gy[i], bz = rest / m.gy[i];
    const ConvP& p = m.p[i];
    if (m.mt[i] == 4) conv_fast_body<4, NT>(p, bx, by, bz);
    else if (m.mt[i] == 2) conv_fast_body<2, NT>(p, bx, by, bz);
    else conv_fast_body<1, NT>(p, bx, by, bz);
}

template <int MT, int NT>
void launch_fast(const ConvP& p, hipStream_t st) {
    long mblk = p.M;
    int nz = p.groups;
    if (p.parity) {
        mblk = (long)p.B * ((p.OH + p.stride - 1) / p.stride) * ((p.OW + p.stride - 1) / p.stride);
        nz *= p.stride * p.stride;
    }
    const dim3 grid((unsigned)((mblk + 64 * MT - 1) / (64 * MT)), (unsigned)(p.Cout_pad / (16 * NT)), (unsigned)nz);
    const size_t lds = (size_t)(p.Kp / 8) * sizeof(i32x2) + (MT == 1 ? (size_t)2 * 8 * NT * 1024 : 0);
    if (MT == 1) {
        static bool attr_set = false;          // more than the default 64 KB of dynamic LDS
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fast_kernel<MT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((conv_fast_kernel<MT, NT>), grid, dim3(256), lds, st, p);
}

const bool g_no_fast = getenv("DANET_CONV_NO_FAST") != nullptr;     // A-B timing knob

}  // namespace

namespace danet_conv {

// Can the lean kernel run this problem?  (p fully populated by danet_conv_forward.)
bool conv_fast_ok(const ConvP& p, bool vec8, int mt) {
    if (g_no_fast || !vec8) return false;
    if (p.transposed && p.stride > 1 && !p.parity) return false;
    if (p.R > 30 || p.S > 26 || p.Cout_g % 4 != 0 || p.Cout % 4 != 0) return false;
    if (p.M >= (1L << 24) || p.Kp >= (1 << 24)) return false;
    if (p.x_bytes >= (1L << 31) || p.y_bytes >= (1L << 31)) return false;
    if ((long)p.groups * p.Cout_pad * p.Kp * 2 >= (1L << 31)) return false;
    // LDS: tap table (Kp bytes) + for one-tile-per-wave launches up to 64 KB of shared weight fragments
    if ((size_t)p.Kp > (size_t)(mt == 1 ? 28 : 60) * 1024) return false;
    return true;
}

int conv_fast_launch(const ConvP& p, int mt, int nt, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define FAST_CASE(M_, N_) if (mt == M_ && nt == N_) { launch_fast<M_, N_>(p, st); return 0; }
    FAST_CASE(1, 1) FAST_CASE(2, 1) FAST_CASE(4, 1) FAST_CASE(1, 2) FAST_CASE(2, 2) FAST_CASE(4, 2)
    FAST_CASE(1, 3) FAST_CASE(2, 3) FAST_CASE(4, 3) FAST_CASE(1, 4) FAST_CASE(2, 4) FAST_CASE(4, 4)
#undef FAST_CASE
    return -1;
}

// All problems must pass conv_fast_ok and share NT.  Returns 0 on launch, -1 if the set is not supported.
int conv_fast_launch_multi(const ConvP* ps, const int* mts, int n, int nt, void* stream) {
    if (n < 1 || n > NCM) return -1;
    ConvMulti m;
    m.n = n; m.start[0] = 0;
    size_t lds = 0;
    // longest k-loops first: workgroups are dispatched in id order, so the long-running ones (small-M, large-K
    // problems) start at once and the short ones fill in behind them instead of leaving a tail
    int order[NCM];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if ((long)ps[order[b]].Kp * mts[order[b]] > (long)ps[order[a]].Kp * mts[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    for (int i = 0; i < n; ++i) {
        const ConvP& p = ps[order[i]];
        long mblk = p.M;
        int nz = p.groups;
        if (p.parity) {
            mblk = (long)p.B * ((p.OH + p.stride - 1) / p.stride) * ((p.OW + p.stride - 1) / p.stride);
            nz *= p.stride * p.stride;
        }
        const int mti = mts[order[i]];
        m.p[i] = p; m.mt[i] = mti;
        m.gx[i] = (int)((mblk + 64 * mti - 1) / (64 * mti));
        m.gy[i] = p.Cout_pad / (16 * nt);
        m.start[i + 1] = m.start[i] + m.gx[i] * m.gy[i] * nz;
        const size_t l = (size_t)(p.Kp / 8) * sizeof(i32x2) + (mti == 1 ? (size_t)2 * 8 * nt * 1024 : 0);
        if (l > lds) lds = l;
    }
    hipStream_t st = (hipStream_t)stream;
#define MULTI_CASE(N_) if (nt == N_) { \
        static bool attr_set = false; \
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fast_multi_kernel<N_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_set = true; } \
        hipLaunchKernelGGL((conv_fast_multi_kernel<N_>), dim3((unsigned)m.start[n]), dim3(256), lds, st, m); return 0; }
    MULTI_CASE(1) MULTI_CASE(2) MULTI_CASE(3) MULTI_CASE(4)
#undef MULTI_CASE
    return -1;
}

}  // namespace danet_conv
