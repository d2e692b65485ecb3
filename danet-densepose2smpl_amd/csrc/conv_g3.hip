// Grouped 3x3 / stride-1 / pad-1 convolution with NARROW groups, forward and data gradient: the 24-group partial-IUV head
// (/root/reference/models/module/res_module.py:281-390 `predict_partial_iuv`, Conv2d(24 * 48, 24 * 21, 3, groups = 24), called at
// /root/reference/models/danet/iuv_estimator.py:206) -- per group 48 -> 21 (24 with padding) channels forward, 24 -> 48 backward, on
// the 302 MB crop tensor of the 24 joint-centric STN resamplings.
//
// Such a layer is 432 (216) multiply-adds per output value: 0.9 GMAC per image but 453 MB of tensors per pass at 32 images -- it is bound
// by moving the tensors (90 us at 5 TB/s), not by the matrix cores.  Until round 6 the forward ran on the streamed 3x3 kernel
// (csrc/conv3x3s.hip: 6 144 tiles of 14 k-steps each, 231 us -- its per-tile fixed cost dominates) and the data gradient on the gather
// kernel (csrc/conv_fast.hip: 24 channels per group are no multiple of the streamed kernel's 16-channel slab pairs; every gradient byte
// crosses the L1 path nine times: 353 us).  Here a workgroup takes one (image, group, row band):
//   * the band's input rows (+ halo, zero padding from out-of-range buffer offsets) are copied to LDS ONCE, 16-byte pieces, as
//     [row][column][channels of the group] with a pixel stride that keeps a fragment read's 16 lanes on distinct banks (48 B at 24
//     channels, 112 B at 48);
//   * the group's packed weights (the fragment-major operand of danet_conv_pack_weights, mode 0 / 1: <= 28 KB) stay in REGISTERS:
//     7 k-steps x NT fragments per lane (the 48-channel case makes two passes of 7 over the accumulators);
//   * a wave owns whole output rows: MT = W / 16 pixel fragments x NT channel blocks of accumulators, one ds_read_b128 per
//     (fragment, k-step) at base + tap offset (a k-step's 8-channel lane chunk lies inside one tap because 24 and 48 are multiples of 8);
//   * outputs leave as 8-byte NHWC stores (four consecutive channels of a pixel per lane), bias added in fp32.
// Two workgroups per compute unit (<= 74 KB of LDS each) overlap one band's loads with the other's MFMAs.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
constexpr int OOB = 0x7fffffff;
constexpr int G3_KH = 7;                                      // k-steps of weight fragments held in registers at a time

template <int CIN> struct G3Cfg;
template <> struct G3Cfg<24> { static constexpr int PS = 48, TH = 16, NT = 3, NHALF = 1; };     // data gradient: 24 -> 48 per group
template <> struct G3Cfg<48> { static constexpr int PS = 112, TH = 8, NT = 2, NHALF = 2; };     // forward:       48 -> 24 per group

template <int CIN, int MT>
__global__ __launch_bounds__(256, 2) void conv_g3_kernel(ConvP p)
{
    using C = G3Cfg<CIN>;
    constexpr int PS = C::PS, TH = C::TH, NT = C::NT, NHALF = C::NHALF;
    constexpr int ROWS = TH / 4;                              // output rows per wave
    constexpr int PC = CIN / 8;                               // 16-byte pieces per staged pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char g3_smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int W = p.W, H = p.H, WP = W + 2;
    const int nband = (H + TH - 1) / TH;
    int id = (int)blockIdx.x;
    const int band = id % nband; id /= nband;
    const int g = id % p.groups, b = id / p.groups;
    const int y0 = band * TH;

    // ---- stage rows y0 - 1 .. y0 + TH, columns -1 .. W of the group's CIN channels
    {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)p.x_bytes, 0x00020000);
        const int npieces = (TH + 2) * WP * PC;
        const int pixb = p.Cin * 2;
        constexpr int NB = 8;                                 // loads in flight per lane (a band is 14 - 16 pieces per lane: two batches)
        for (int i0 = 0; i0 < npieces; i0 += 256 * NB) {
            i32x4 v[NB];
            int dst[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int i = i0 + u * 256 + t;
                const int pc = i % PC, q = i / PC;
                const int col = q % WP, row = q / WP;
                const int y = y0 - 1 + row, x = col - 1;
                const bool ok = i < npieces && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? ((b * H + y) * W + x) * pixb + (g * CIN + pc * 8) * 2 : OOB, 0, 0);
                dst[u] = i < npieces ? (row * WP + col) * PS + pc * 16 : -1;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (dst[u] >= 0) *reinterpret_cast<i32x4*>(g3_smem + dst[u]) = v[u];
        }
    }

    // ---- per-lane tap offsets of the k-steps: k = ks * 32 + lg * 8 -> (tap, first channel); taps mirrored for the data gradient
    constexpr int NKS = NHALF * G3_KH;
    int tapoff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int k = ks * 32 + lg * 8;
        int tap = k / CIN;
        const int ch = k - tap * CIN;
        tap = tap > 8 ? 8 : tap;                              // (K padding: the weights there are zero, any valid cell will do)
        const int r = tap / 3, s = tap - r * 3;
        const int rr = p.transposed ? 2 - r : r, ss = p.transposed ? 2 - s : s;
        tapoff[ks] = (rr * WP + ss) * PS + ch * 2;
    }
    const int nks_w = p.Kp / 32;
    const bf16_t* const wg = p.w + (size_t)g * (p.Cout_pad / 16) * nks_w * 512 + lane * 8;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const float* const bias = p.bias;
    __syncthreads();

    f32x4 acc[NHALF > 1 ? ROWS : 1][MT][NT];
    auto zero = [&](int r) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[r][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_row = [&](int r, int yl) {
        const int y = y0 + yl;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = nt * 16 + lg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && c < p.Cout_g) bv = *reinterpret_cast<const f32x4*>(bias + g * p.Cout_g + c);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 v = acc[r][mt][nt] + bv;
                const int x = mt * 16 + li;
                const bool ok = y < H && c < p.Cout_g;
                const i32x2 o = {(int)f2bf_pk(v[0], v[1]), (int)f2bf_pk(v[2], v[3])};
                __builtin_amdgcn_raw_buffer_store_b64(o, yr, ok ? (((b * H + y) * W + x) * p.Cout + g * p.Cout_g + c) * 2 : OOB, 0, 0);
            }
        }
    };
#pragma unroll
    for (int half = 0; half < NHALF; ++half) {
        bf16x8 A[NT][G3_KH];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int kk = 0; kk < G3_KH; ++kk) {
                const int ks = half * G3_KH + kk;
                A[nt][kk] = ks < nks_w ? *reinterpret_cast<const bf16x8*>(wg + ((size_t)nt * nks_w + ks) * 512) : bf16x8{};
            }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int yl = wave * ROWS + r;
            const int ar = NHALF > 1 ? r : 0;
            if (half == 0) zero(ar);
            const int rowbase = (yl * WP + li) * PS;
#pragma unroll
            for (int kk = 0; kk < G3_KH; ++kk) {
                const int off = tapoff[half * G3_KH + kk];
                bf16x8 bq[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    bq[mt] = *reinterpret_cast<const bf16x8*>(g3_smem + rowbase + mt * 16 * PS + off);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[ar][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[nt][kk], bq[mt], acc[ar][mt][nt], 0, 0, 0);
            }
            if (half == NHALF - 1) store_row(ar, yl);
        }
    }
}

template <int CIN, int MT>
int g3_launch(const ConvP& p, void* stream)
{
    using C = G3Cfg<CIN>;
    const int nband = (p.H + C::TH - 1) / C::TH;
    const size_t lds = (size_t)(C::TH + 2) * (p.W + 2) * C::PS + 16;
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_g3_kernel<CIN, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); raised = true; }
    hipLaunchKernelGGL((conv_g3_kernel<CIN, MT>), dim3((unsigned)(p.B * p.groups * nband)), dim3(256), lds, (hipStream_t)stream, p);
    return 0;
}

bool g_g3_on = true;

}  // namespace

namespace danet_conv {

// The narrow-group 3x3 layers this kernel takes: per group 48 -> 17..32 channels (forward of the partial-IUV head) or 24 -> 33..48
// (its data gradient), stride 1, pad 1, rows of 16 .. 64 pixels in multiples of 16, bf16 output, no fused statistics / addend / ReLU.
bool conv_g3_ok(const ConvP& p, bool vec8)
{
    static const bool off = getenv("DANET_NO_CONV_G3") != nullptr;          // A-B knob
    if (off || !g_g3_on || !vec8 || p.groups < 2) return false;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.H != p.OH || p.W != p.OW) return false;
    if (p.out_fp32 || p.relu || p.stats || p.bn_red || p.addend || p.bna) return false;
    if (p.W % 16 != 0 || p.W < 16 || p.W > 64) return false;
    if (p.x_bytes >= (1L << 31) || p.y_bytes >= (1L << 31)) return false;
    const int nt = danet_conv_nt(p.Cout_g);
    if (p.Cin_g == 48 && !p.transposed && nt == 2 && p.Cout_g % 4 == 0 && p.Kp == 448) return true;
    if (p.Cin_g == 24 && p.transposed && nt == 3 && p.Cout_g % 4 == 0 && p.Kp == 224) return true;
    return false;
}

long conv_g3_knob(long value) { const long prev = g_g3_on ? 1 : 0; if (value >= 0) g_g3_on = value != 0; return prev; }

int conv_g3_launch(const ConvP& p, void* stream)
{
    const int mt = p.W / 16;
#define G3_CASE(C_, M_) if (p.Cin_g == C_ && mt == M_) return g3_launch<C_, M_>(p, stream);
    G3_CASE(48, 4) G3_CASE(48, 3) G3_CASE(48, 2) G3_CASE(48, 1)
    G3_CASE(24, 4) G3_CASE(24, 3) G3_CASE(24, 2) G3_CASE(24, 1)
#undef G3_CASE
    return -1;
}

}  // namespace danet_conv
