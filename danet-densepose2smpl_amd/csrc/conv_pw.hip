// Pointwise (1x1, stride 1) convolution, forward and data gradient: the Bottleneck projections 64 <-> 256 at 64x64
// (/root/reference/models/module/res_module.py:64-97), the limb regressor's 24 -> 64 input projection over the 768 part crops,
// the heat-map head's narrow bottlenecks.  A 1x1 layer is a GEMM  Y[pixels, Cout] = X[pixels, Cin] W^T  whose whole cost is
// moving X in and Y out (51 FLOP per byte at 64 -> 256: 10 % of the matrix cores at the HBM rate); the gather kernel
// (conv_fast.hip) ran them at 2-3x their HBM time: its workgroup grid is (pixel tiles) x (48..64-channel blocks), so a
// 256-channel layer re-read X four times through L1 / L2, every workgroup of ~0.3 us of work paid the full prologue (tap table,
// integer divisions, bit masks), and every wave pulled its weight slab through the texture path.  Here
//   * the packed weights of the WHOLE layer (<= 64 KB: 256 x 64) are copied into LDS once per workgroup; workgroups are
//     persistent (two per compute unit) and walk over pixel tiles;
//   * a wave takes 64 pixels of a tile and one group of <= 4 sixteen-channel output blocks: it loads its X fragments -- all
//     k-steps of the 64 pixels, 16 bytes per lane and load, straight in MFMA operand layout -- ONCE, up front (up to 32 loads in
//     flight per lane: memory-level parallelism comes from the loads of eight resident waves per compute unit, not from a
//     software pipeline), multiplies against the weight fragments read from LDS and stores; with more than one group the waves of
//     a workgroup take the groups of the SAME pixel tile (X reaches the workgroup once from L2, the other waves hit L1);
//   * no tap table, no masks: a pixel's byte offset is its index times the row pitch, rows beyond the tensor and channel
//     chunks beyond Cin carry an out-of-range buffer offset (zeros in, nothing out);
//   * the optional BatchNorm statistics of the output are accumulated per lane across ALL tiles of the wave (packed fp32 math)
//     and leave once per workgroup: DPP row sums, LDS across the waves of a group, one atomic per channel into a replica.
// Same packed-weight layout, same epilogue options (bias, ReLU, fp32 output, bf16 addend) as conv_fast.hip, so the two are
// interchangeable per problem (conv_pw_ok decides).
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int OOB = 0x7fffffff;
constexpr int PW_MT = 4;                      // 16-pixel fragments per wave and tile (64 pixels)
constexpr int PW_LDS_MAX = 64 * 1024;         // packed weights of a layer (two workgroups per compute unit)

template <int CTRL>
__device__ inline float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, o);
}
__device__ inline float row_sum16(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    return v;
}

struct PwP {
    const bf16_t* x; const bf16_t* w; const float* bias; void* y; float* stats; const bf16_t* addend;
    int M, Cin, Cout, nks, NB;                // pixels, channels, k-steps (Kp / 32), sixteen-row blocks (Cout_pad / 16)
    int ngroups, tiles_per_iter, niter;       // waves per tile = ngroups (1, 2, 4); tiles a workgroup handles per iteration = 4 / ngroups
    int relu, out_fp32;
    int x_bytes, y_bytes;
};

// NKS: k-steps a wave keeps in registers (>= nks), NTB: output blocks per group
template <int NKS, int NTB>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(PwP p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sW[];       // [NB][nks] fragments of 1 KB, then the statistics scratch
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nks = p.nks, NB = p.NB;
    {
        // the layer's packed weights: NB * nks contiguous KB (fragment-major: a wave's 1 KB piece is one fragment)
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, NB * nks * 1024, 0x00020000);
        for (int i = t; i < NB * nks * 64; i += 256)
            *reinterpret_cast<i32x4*>(sW + (size_t)i * 16) = __builtin_amdgcn_raw_buffer_load_b128(wr, i * 16, 0, 0);
    }
    const int grp = wave % p.ngroups, sub = wave / p.ngroups;           // this wave's output group and its tile within the iteration
    const int nb0 = grp * NTB;                                          // first sixteen-row block of the group
    const int n0 = nb0 * 16;
    const int pitch = p.Cin * 2;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    // channel chunk of this lane in k-step ks: bytes (ks * 32 + lg * 8) * 2 of a pixel's row; chunks beyond Cin read as zeros
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = (ks < nks && ks * 32 + lg * 8 < p.Cin) ? (ks * 32 + lg * 8) * 2 : OOB;
    float s1[NTB][4], s2[NTB][4];
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    __syncthreads();                                                     // weights in LDS

    // X fragments of the NEXT tile are requested before the current tile's stores are issued (NKS <= 4: two register sets):
    // the vector-memory counter returns in order, so loads queued behind a tile's 16 stores would wait for every store's
    // acknowledgement first (measured: 24 -> 64 channels over 768 crops at 2.3 TB/s without, slower than the gather kernel).
    constexpr bool PF = NKS <= 2 || (NKS == 4 && NTB <= 3);           // (<4, 4> with two sets: 256 registers + scratch)
    auto load_tile = [&](int it, bf16x8 (*xq)[PW_MT], int* pix) {
        const int m0 = (it * p.tiles_per_iter + sub) * (16 * PW_MT);
#pragma unroll
        for (int mt = 0; mt < PW_MT; ++mt) {
            const int m = m0 + mt * 16 + li;
            pix[mt] = (it < p.niter && m < p.M) ? m : -1;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int mt = 0; mt < PW_MT; ++mt)
                xq[ks][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, (pix[mt] >= 0 && koff[ks] != OOB) ? pix[mt] * pitch + koff[ks] : OOB, 0, 0));
    };
    auto do_tile = [&](const bf16x8 (*xq)[PW_MT], const int* pix) {
        f32x4 acc[PW_MT][NTB];
#pragma unroll
        for (int mt = 0; mt < PW_MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < nks) {                                              // (uniform)
                bf16x8 a[NTB];
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt)
                    a[nt] = *reinterpret_cast<const bf16x8*>(sW + (size_t)((min(nb0 + nt, NB - 1)) * nks + ks) * 1024 + lane * 16);
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                    for (int mt = 0; mt < PW_MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], xq[ks][mt], acc[mt][nt], 0, 0, 0);
            }
        }
        // epilogue: the lane holds output channels n0 + nt * 16 + lg * 4 + {0..3} of pixel li of each fragment
        f32x4 bv[NTB];
        bool cok[NTB];
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
            const int cl = n0 + nt * 16 + lg * 4;
            cok[nt] = cl < p.Cout && nb0 + nt < NB;
            bv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.Cout * 4, 0x00020000);
                bv[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(br, cok[nt] ? cl * 4 : OOB, 0, 0));
            }
        }
        auto value = [&](int mt, int nt, bool& ok) {
            const int cl = n0 + nt * 16 + lg * 4;
            ok = cok[nt] && pix[mt] >= 0;
            f32x4 v = acc[mt][nt] + bv[nt];
            if (p.addend) {
                const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.addend), 0, p.y_bytes, 0x00020000);
                const i32x2 aq = __builtin_amdgcn_raw_buffer_load_b64(ar, ok ? (pix[mt] * p.Cout + cl) * 2 : OOB, 0, 0);
                v[0] += __uint_as_float((unsigned)aq.x << 16); v[1] += __uint_as_float((unsigned)aq.x & 0xffff0000u);
                v[2] += __uint_as_float((unsigned)aq.y << 16); v[3] += __uint_as_float((unsigned)aq.y & 0xffff0000u);
            }
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            return v;
        };
        auto stat = [&](int nt, const i32x2& pk, bool ok) {               // statistics of the bf16-ROUNDED output, as conv_fast.hip takes them
            const float msk = ok ? 1.f : 0.f;
            f32x2_ lo = {__uint_as_float((unsigned)pk.x << 16) * msk, __uint_as_float((unsigned)pk.x & 0xffff0000u) * msk};
            f32x2_ hi = {__uint_as_float((unsigned)pk.y << 16) * msk, __uint_as_float((unsigned)pk.y & 0xffff0000u) * msk};
            f32x2_& a0 = *reinterpret_cast<f32x2_*>(&s1[nt][0]); f32x2_& a1 = *reinterpret_cast<f32x2_*>(&s1[nt][2]);
            f32x2_& q0 = *reinterpret_cast<f32x2_*>(&s2[nt][0]); f32x2_& q1 = *reinterpret_cast<f32x2_*>(&s2[nt][2]);
            a0 += lo; a1 += hi;
            q0 = __builtin_elementwise_fma(lo, lo, q0); q1 = __builtin_elementwise_fma(hi, hi, q1);
        };
        if (p.out_fp32) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                for (int mt = 0; mt < PW_MT; ++mt) {
                    bool ok;
                    const f32x4 v = value(mt, nt, ok);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yr, ok ? (pix[mt] * p.Cout + n0 + nt * 16 + lg * 4) * 4 : OOB, 0, 0);
                }
        } else {
            // bf16: two output blocks at a time leave as 16-byte stores.  v_permlane16_swap exchanges lane rows 1 <-> 0 and 3 <-> 2
            // between two registers: from (block a: channels 4 lg .. + 3, block b: the same) a lane ends up with EIGHT consecutive
            // channels -- rows 0 / 2 of block a (channels 0-7 / 8-15), rows 1 / 3 of block b -- instead of two 8-byte pieces
            // 32 bytes apart (the write path handles a 64-lane store as 16-byte pieces per lane: half the instructions).
#pragma unroll
            for (int np = 0; np + 1 < NTB; np += 2)
#pragma unroll
                for (int mt = 0; mt < PW_MT; ++mt) {
                    bool oka, okb;
                    const f32x4 va = value(mt, np, oka), vb = value(mt, np + 1, okb);
                    const i32x2 pa = {(int)f2bf_pk(va[0], va[1]), (int)f2bf_pk(va[2], va[3])}, pb = {(int)f2bf_pk(vb[0], vb[1]), (int)f2bf_pk(vb[2], vb[3])};
                    if (p.stats) { stat(np, pa, oka); stat(np + 1, pb, okb); }
                    const auto sx = __builtin_amdgcn_permlane16_swap((unsigned)pa.x, (unsigned)pb.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap((unsigned)pa.y, (unsigned)pb.y, false, false);
                    const i32x4 q = {(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
                    const int blk = np + (lg & 1), c8 = n0 + blk * 16 + (lg >> 1) * 8;      // the lane's eight channels after the exchange
                    const bool ok = pix[mt] >= 0 && c8 < p.Cout && nb0 + blk < NB;         // (Cout % 8 == 0 on this path, see conv_pw_ok)
                    __builtin_amdgcn_raw_buffer_store_b128(q, yr, ok ? (pix[mt] * p.Cout + c8) * 2 : OOB, 0, 0);
                }
            if constexpr (NTB % 2 == 1) {
                constexpr int nt = NTB - 1;
#pragma unroll
                for (int mt = 0; mt < PW_MT; ++mt) {
                    bool ok;
                    const f32x4 v = value(mt, nt, ok);
                    const i32x2 pk = {(int)f2bf_pk(v[0], v[1]), (int)f2bf_pk(v[2], v[3])};
                    if (p.stats) stat(nt, pk, ok);
                    __builtin_amdgcn_raw_buffer_store_b64(pk, yr, ok ? (pix[mt] * p.Cout + n0 + nt * 16 + lg * 4) * 2 : OOB, 0, 0);
                }
            }
        }
    };
    if constexpr (PF) {
        bf16x8 xa[NKS][PW_MT], xb[NKS][PW_MT];
        int pa[PW_MT], pb[PW_MT];
        load_tile(blockIdx.x, xa, pa);
        for (int it = blockIdx.x; it < p.niter; it += 2 * gridDim.x) {
            load_tile(it + gridDim.x, xb, pb);
            do_tile(xa, pa);
            if (it + (int)gridDim.x >= p.niter) break;
            load_tile(it + 2 * gridDim.x, xa, pa);
            do_tile(xb, pb);
        }
    } else {
        for (int it = blockIdx.x; it < p.niter; it += gridDim.x) {
            bf16x8 xq[NKS][PW_MT];
            int pix[PW_MT];
            load_tile(it, xq, pix);
            do_tile(xq, pix);
        }
    }
    if (p.stats) {
        // 16 pixel lanes (DPP) -> the waves of the group (LDS, behind the weights) -> one atomic per channel into replica blockIdx % ncopy
        float* const sStat = reinterpret_cast<float*>(sW + (size_t)NB * nks * 1024);          // [4 waves][2][NTB * 16]
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = row_sum16(s1[nt][r]), b = row_sum16(s2[nt][r]);
                if (li == 0) { sStat[(wave * 2 + 0) * (NTB * 16) + nt * 16 + lg * 4 + r] = a; sStat[(wave * 2 + 1) * (NTB * 16) + nt * 16 + lg * 4 + r] = b; }
            }
        __syncthreads();
        const int nsub = 4 / p.ngroups;
        for (int i = t; i < p.ngroups * 2 * NTB * 16; i += 256) {
            const int g2 = i / (2 * NTB * 16), rem = i - g2 * (2 * NTB * 16), which = rem / (NTB * 16), c = rem - which * (NTB * 16);
            float v = 0.f;
            for (int s = 0; s < nsub; ++s) v += sStat[((s * p.ngroups + g2) * 2 + which) * (NTB * 16) + c];
            const int ch = g2 * NTB * 16 + c;
            if (ch < p.Cout && g2 * NTB + c / 16 < NB)
                bn_acc_add(p.stats, blockIdx.x, which, p.Cout, ch, v);
        }
    }
}

bool g_pw_on = getenv("DANET_CONV_NO_PW") == nullptr;              // A-B timing knob (danet_conv_pw_set)

struct PwPlan { int nks_t, ntb, ngroups; };

bool pw_plan(const ConvP& p, PwPlan& pl) {
    const int nks = p.Kp / 32, NB = p.Cout_pad / 16;
    if (nks > 8) return false;
    pl.nks_t = nks <= 2 ? 2 : (nks <= 4 ? 4 : 8);
    int ng = NB <= 4 ? 1 : (NB <= 8 ? 2 : 4);
    int ntb = (NB + ng - 1) / ng;
    if (ntb > 4) return false;
    // registers: NKS * 16 (X fragments) + NTB * 16 (accumulators) + NTB * 8 (statistics)
    if (pl.nks_t == 8 && ntb > 4) return false;
    pl.ntb = ntb; pl.ngroups = ng;
    return true;
}

// Workgroups: persistent, as many as are resident at once (registers and the layer's LDS footprint decide: 2 .. 8 per compute unit)
template <int NKS, int NTB>
void pw_launch_t(const PwP& q, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    static int per_cu[3] = {0, 0, 0};                 // by LDS class: <= 16 KB, <= 32 KB, more
    static int cus = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<NKS, NTB>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS_MAX + 4096);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr_set = true;
    }
    const int cls = lds <= 16 * 1024 ? 0 : (lds <= 32 * 1024 ? 1 : 2);
    if (per_cu[cls] == 0) {
        int n = 0;
        const size_t probe = cls == 0 ? 16 * 1024 : (cls == 1 ? 32 * 1024 : PW_LDS_MAX + 2048);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&conv_pw_kernel<NKS, NTB>), 256, probe) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 2; }
        per_cu[cls] = n > 8 ? 8 : n;
    }
    const int cap = cus * per_cu[cls];
    hipLaunchKernelGGL((conv_pw_kernel<NKS, NTB>), dim3((unsigned)(q.niter < cap ? q.niter : cap)), dim3(256), lds, st, q);
}

}  // namespace

namespace danet_conv {

// 1x1 / stride 1 / no padding / one group, channel counts in 8-channel granules, the whole packed weight operand in LDS,
// and none of the fused BatchNorm-backward reductions (conv_fast.hip keeps those).
bool conv_pw_ok(const ConvP& p, bool vec8) {
    if (!g_pw_on || !vec8) return false;
    if (p.R != 1 || p.S != 1 || p.stride != 1 || p.pad != 0 || p.dil != 1 || p.groups != 1) return false;
    if (p.H != p.OH || p.W != p.OW || p.Cout % 8 != 0 || p.bn_red) return false;        // (16-byte stores: eight channels per lane)
    if (p.M >= (1L << 24) || p.x_bytes >= (1L << 31) || p.y_bytes >= (1L << 31)) return false;
    if ((long)(p.Cout_pad / 16) * (p.Kp / 32) * 1024 > PW_LDS_MAX) return false;
    if (p.M < 8192) return false;                       // (small layers: the gather kernel's grid fills the chip better than 64-pixel tiles)
    PwPlan pl;
    return pw_plan(p, pl);
}

int conv_pw_config(const ConvP& p) {                    // NKS * 10 + NTB (profilers label their records with it)
    PwPlan pl;
    return pw_plan(p, pl) ? pl.nks_t * 10 + pl.ntb : 0;
}

int conv_pw_launch(const ConvP& p, void* stream) {
    PwPlan pl;
    if (!pw_plan(p, pl)) return -1;
    PwP q{};
    q.x = p.x; q.w = p.w; q.bias = p.bias; q.y = p.y; q.stats = p.stats; q.addend = p.addend;
    q.M = (int)p.M; q.Cin = p.Cin; q.Cout = p.Cout; q.nks = p.Kp / 32; q.NB = p.Cout_pad / 16;
    q.ngroups = pl.ngroups; q.tiles_per_iter = 4 / pl.ngroups;
    const long ntiles = (p.M + 16 * PW_MT - 1) / (16 * PW_MT);
    q.niter = (int)((ntiles + q.tiles_per_iter - 1) / q.tiles_per_iter);
    q.relu = p.relu; q.out_fp32 = p.out_fp32;
    q.x_bytes = (int)p.x_bytes; q.y_bytes = (int)p.y_bytes;
    const size_t lds = (size_t)q.NB * q.nks * 1024 + 4 * 2 * 4 * 16 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define PW_CASE(K_, N_) if (pl.nks_t == K_ && pl.ntb == N_) { pw_launch_t<K_, N_>(q, lds, st); return 0; }
    PW_CASE(2, 1) PW_CASE(2, 2) PW_CASE(2, 3) PW_CASE(2, 4)
    PW_CASE(4, 1) PW_CASE(4, 2) PW_CASE(4, 3) PW_CASE(4, 4)
    PW_CASE(8, 1) PW_CASE(8, 2) PW_CASE(8, 3) PW_CASE(8, 4)
#undef PW_CASE
    return -1;
}

}  // namespace danet_conv

// Run-time switch of the pointwise kernel (A-B timing, tests): enable 0 / 1 (-1 keeps); returns the previous setting.
long danet_conv::conv_pw_knob(long enable) {
    const long prev = g_pw_on ? 1 : 0;
    if (enable >= 0) g_pw_on = enable != 0;
    return prev;
}
