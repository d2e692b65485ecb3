// 3x3 / stride-1 / pad-1 convolution with 64 input and 64 output channels on the stem kernels' machinery (conv_stem_dgrad.hip): forward and
// data gradient of the BasicBlocks of the regressor ResNets' layer1 -- /root/reference/models/module/res_module.py:27-56 as used by
// SmplResNet (:404) over the 768 part crops (16 x 16 maps) and over the 32 body maps (64 x 64) -- 17 launches per step that ran on
// conv3x3_tile_kernel at 12-15 % of the bf16 MFMA peak.  It is also the single-shape test bed of the successor of conv3x3_stream_kernel
// (DESIGN.md 8.15): one workgroup per CU, 512 registers per lane, 8 x 4 accumulator tiles per wave in AGPRs behind inline-asm MFMAs.
//   * a tile = 256 output pixels = TH rows x TW columns of one image (TW = 16: a whole 16 x 16 map; TW = 64: four rows); its input --
//     (TH + 2) x (TW + 2) cells with the halo, all 64 channels as four 16-channel planes of 32-byte cells -- reaches LDS once by LDS-DMA
//     (per-lane source offsets from a table, out-of-image cells requested out of range = zeros), two slots: the next tile's travel
//     while this one is multiplied;
//   * a k-step = ONE tap x 32 channels (lanes 0-31 / 32-63: two neighbouring channel planes); 18 per tile, cut 12 | 6 between the two
//     K halves of a pixel half (multiples of the three-deep weight ring); the waves with the 6 issue the next tile's copies first and sit
//     them out (the ring wait, "at most 8 outstanding", counts them); per (K half, k-step) a table entry holds the tap's LDS offset and the weight offsets of the two half fragments
//     (chunk-16 packing of danet_conv_pack_weights: a tap's 16 channels are 512 contiguous bytes);
//   * the data gradient is the same loop with the taps mirrored (cell (2 - r, 2 - s)) over the mode-1 packing;
//   * the K halves meet in LDS, then the epilogue: optional residual addend (the identity shortcut's gradient, conv.ResLink), 16-byte
//     stores through v_permlane16_swap, the output's BatchNorm statistics (forward) or the producing BatchNorm's backward sums gated by
//     its output or its byte mask (data gradient) -- the contracts of danet_conv_forward's bn_sums / bn_* / addend arguments.
#include "common.h"
#include "conv_common.h"
#include <type_traits>

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int OOB = 0x7fffffff;
constexpr int CA_NT = 4, CA_MT = 8, CA_D = 3, CA_KS = 5;    // 5 k-steps per 16-channel slab in the chunk-16 packing (9 taps + 1 zero tap)
constexpr int CA_NE0 = 12, CA_NE1 = 6, CA_NEMAX = 16;        // k-steps per tile of the two K halves (+ wrap-around entries)
constexpr int CA_EXCH = 32768;

template <int TW> struct Geo {
    static constexpr int TH = 256 / TW, ROWS = TH + 2, CELLS = TW + 2;
    static constexpr int ROWB = CELLS * 32, PLANE = ROWS * ROWB, SLOT = 4 * PLANE;
    static constexpr int PIECES = SLOT / 16, NDMA = (PIECES + 63) / 64, NCP = (NDMA + 1) / 2;
    static constexpr int TAB = 2 * CA_NEMAX * 2 * 4, OFFT = 2 * NCP * 64 * 4;
    static constexpr int LDS = 2 * SLOT + CA_EXCH + 2 * TAB + OFFT + 512 + 4 * 512;   // + mean / invstd [2][64] + per-wave statistics accumulators [4][2][64]
};

struct C3aP {
    const bf16_t* x; const bf16_t* w; void* y;
    float* stats;                                            // forward: [BN_NCOPY][2][64] statistics of the output, or NULL
    const bf16_t* bn_x; const void* bn_y; const float* bn_saved; float* bn_red; int bn_gate;      // data gradient: fused BatchNorm-backward sums
    const bf16_t* addend;
    int B, H, W, mirrored;
    int ntiles, strips;                                      // tiles per image = H / TH
    int bytes;                                               // of x (= of y: same shape)
};

__device__ __forceinline__ void dma16(unsigned lds_addr, int voff, const i32x4& desc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ inline i32x4 raw_desc(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000};
}
__device__ inline void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int CTRL>
__device__ inline float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, o);
}
__device__ inline float row_sum16(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    return v;
}

template <int TW>
__global__ __launch_bounds__(256, 1) void conv3x3a_kernel(C3aP p)
{
    using G_ = Geo<TW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave & 1, kw = wave >> 1;
    unsigned char* const ring = smem;
    unsigned char* const sR = smem + 2 * G_::SLOT;
    int* const sTabA = reinterpret_cast<int*>(smem + 2 * G_::SLOT + CA_EXCH);           // [role][entry][half]
    int* const sTabB = sTabA + 2 * CA_NEMAX * 2;
    int* const sOff = sTabB + 2 * CA_NEMAX * 2;                                            // [issuing wave][instruction][lane]
    float* const sMean = reinterpret_cast<float*>(sOff + 2 * G_::NCP * 64);                // [2][64] mean, invstd
    float* const sAcc = sMean + 128;                                                       // [wave][2][64]: every wave adds to its own copy (no LDS atomics: their order is not fixed)
    // ---- k-step tables: k-step kidx of the tile = (tap = kidx >> 1, channel pair = kidx & 1); lane half h takes channel plane 2 pair + h
    for (int e = t; e < 2 * CA_NEMAX * 2; e += 256) {
        const int half = e & 1, ent = (e >> 1) % CA_NEMAX, role = (e >> 1) / CA_NEMAX;
        const int ne = role == 0 ? CA_NE0 : CA_NE1;
        const int kidx = (role == 0 ? 0 : CA_NE0) + ent % ne;
        const int tap = kidx >> 1, plane = 2 * (kidx & 1) + half;
        const int r = tap / 3, s = tap - 3 * r;
        const int cr = p.mirrored ? 2 - r : r, cs = p.mirrored ? 2 - s : s;                // cell of the tap relative to the output pixel's (staged origin (-1, -1))
        sTabA[e] = (plane * CA_KS + (tap >> 1)) * 1024 + (tap & 1) * 512;
        sTabB[e] = plane * G_::PLANE + (cr * G_::CELLS + cs) * 32;
    }
    // ---- the tile copy: piece q of the slot = (plane, row, cell, half): input pixel (row - 1, cell - 1) of the tile, channels plane * 16 + half * 8 ..
    const int pixb = 128, rowb = p.W * pixb;
    for (int e = t; e < 2 * G_::NCP * 64; e += 256) {
        const int ln = e & 63, u = (e >> 6) % G_::NCP, half_w = (e >> 6) / G_::NCP;
        const int i = half_w + 2 * u;
        const int q = i * 64 + ln;
        const int plane = q / (G_::PLANE / 16), rem = q - plane * (G_::PLANE / 16);
        const int row = rem / (G_::ROWB / 16), r2 = rem - row * (G_::ROWB / 16);
        const int cell = r2 >> 1, half = r2 & 1;
        const int ox = cell - 1;                                                           // (tiles span the image's full width: the column test is static)
        int v = -1;
        if (i < G_::NDMA && q < G_::PIECES) v = ((unsigned)ox < (unsigned)p.W ? row * rowb + ox * pixb + plane * 32 + half * 16 : 0x00ffffff) | (row << 24);
        sOff[e] = v;
    }
    sAcc[t] = 0.f; sAcc[256 + t] = 0.f;
    if (t < 128) sMean[t] = p.bn_red ? p.bn_saved[t] : 0.f;
    const i32x4 xdesc = raw_desc(p.x, p.bytes);
    int lanebase[CA_MT];
#pragma unroll
    for (int mt = 0; mt < CA_MT; ++mt) {
        const int f = pw * 8 + mt;                                                         // fragment of the tile: row f / (TW / 16), 16 columns from (f % (TW / 16)) * 16
        const int row = f / (TW / 16), col = (f % (TW / 16)) * 16 + li;
        lanebase[mt] = (row * G_::CELLS + col) * 32 + (lg & 1) * 16;
    }
    constexpr int NKS = 4 * CA_KS;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.bytes, 0x00020000);
    const i32x4 wdesc = raw_desc(p.w, CA_NT * NKS * 1024);
    int wso[CA_NT];
#pragma unroll
    for (int nt = 0; nt < CA_NT; ++nt) wso[nt] = __builtin_amdgcn_readfirstlane(nt * NKS * 1024);
    const int wlane = (lane & 31) * 16;
    const int h = lg >> 1;
    bf16x8 A[CA_D][CA_NT];
    auto load_a = [&](int ao, bf16x8* a) {
        const int voff = wlane + ao;
#pragma unroll
        for (int nt = 0; nt < CA_NT; ++nt)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a[nt]) : "v"(voff), "s"(wdesc), "s"(wso[nt]) : "memory");
    };
    int g = 0;
    auto issue_mine = [&](int tile, int slot) {             // the copy instructions of this (kw = 1) wave
        if (tile >= p.ntiles) return;
        const int b = tile / p.strips, strip = tile - b * p.strips;
        const int y0 = G_::TH * strip - 1;
        const int soff = __builtin_amdgcn_readfirstlane((b * p.H + y0) * rowb);
        const unsigned dst0 = (unsigned)(unsigned long long)(lds_ptr_t)(ring + slot * G_::SLOT);
        const int* const mine = sOff + (pw * G_::NCP) * 64 + lane;
#pragma unroll
        for (int u = 0; u < G_::NCP; ++u) {
            const int e = mine[u * 64];
            const int row = e >> 24, off = e & 0x00ffffff;
            const bool ok = off != 0x00ffffff && (unsigned)(y0 + row) < (unsigned)p.H;
            if (e >= 0) dma16(dst0 + (unsigned)((pw + 2 * u) * 1024), ok ? off + soff : OOB, xdesc, 0);
        }
    };
    __syncthreads();                                        // tables written
    if (kw == 1) issue_mine(blockIdx.x, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    // data gradient: per-lane BatchNorm-backward sums of its 2 x 8 channels over every tile of the workgroup
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[a][k] = 0.f; s2[a][k] = 0.f; }
    const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_x ? p.bn_x : p.x), 0, p.bn_x ? p.bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_y ? p.bn_y : (const void*)p.x), 0,
                                                                         p.bn_y ? (p.bn_gate == 2 ? p.bytes / 2 : p.bytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t adr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.addend ? p.addend : p.x), 0, p.addend ? p.bytes : 0, 0x00020000);

    auto run = [&](auto kwc) {
    constexpr int KW = decltype(kwc)::value;
    constexpr int NE = KW == 0 ? CA_NE0 : CA_NE1;
    const int* const tA = sTabA + KW * CA_NEMAX * 2 + h;
    const int* const tB = sTabB + KW * CA_NEMAX * 2 + h;
    int pf = 0;
    auto refill = [&](bf16x8* a, int ao) {
        load_a(ao, a);
        pf = pf + 1 == NE ? 0 : pf + 1;
    };
#pragma unroll
    for (int d = 0; d < CA_D; ++d) refill(A[d], tA[2 * d]);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const unsigned char* const sX = ring + (g & 1) * G_::SLOT;
        const int b = tile / p.strips, strip = tile - b * p.strips;
        if constexpr (KW == 1) issue_mine(tile + gridDim.x, (g + 1) & 1);       // the next tile starts travelling: waited out in this wave's first ring turn
        f32x4 acc[CA_MT][CA_NT];                            // written by the tile's first k-step (C = 0)
        bf16x8 Bq[2][CA_MT];
        auto load_b = [&](int to, bf16x8* bq) {
#pragma unroll
            for (int mt = 0; mt < CA_MT; ++mt) bq[mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + to);
        };
        int tnext = tB[2];
        load_b(tB[0], Bq[0]);
#define CA_KSTEP(a, bc, bn, e, more, first) do { \
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]) : "n"(4 * (CA_D - 1))); \
            const int ao_ = tA[2 * pf]; \
            if (more) { load_b(tnext, bn); tnext = tB[2 * ((e) + 2)]; } \
            if (first) { \
                _Pragma("unroll") for (int nt = 0; nt < CA_NT; ++nt) \
                    _Pragma("unroll") for (int mt = 0; mt < CA_MT; ++mt) \
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"((a)[nt]), "v"((bc)[mt])); \
            } else { \
                _Pragma("unroll") for (int nt = 0; nt < CA_NT; ++nt) \
                    _Pragma("unroll") for (int mt = 0; mt < CA_MT; ++mt) \
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"((a)[nt]), "v"((bc)[mt])); \
            } \
            refill(a, ao_); \
            __builtin_amdgcn_sched_barrier(0); \
        } while (0)
#pragma unroll
        for (int q = 0; q < NE; ++q) CA_KSTEP(A[q % CA_D], Bq[q & 1], Bq[(q + 1) & 1], q, q + 1 < NE, q == 0);
#undef CA_KSTEP
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                   // (inline-asm MFMAs: the hazard recogniser does not see them)
        // ---- the two K halves meet in LDS: wave (pw, 0) finishes fragments 0-3, wave (pw, 1) fragments 4-7
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            unsigned char* const mine = sR + wave * 8192 + lane * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                constexpr int base = (1 - KW) * 4;
#pragma unroll
                for (int nt = 0; nt < CA_NT; ++nt) {
                    f32x4 v = acc[base + round * 2 + m][nt];
                    asm volatile("" : "+v"(v));
                    *reinterpret_cast<f32x4*>(mine + (m * CA_NT + nt) * 1024) = v;
                }
            }
            lds_barrier();
            const unsigned char* const theirs = sR + (wave ^ 2) * 8192 + lane * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < CA_NT; ++nt) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + (m * CA_NT + nt) * 1024);
                    acc[KW * 4 + round * 2 + m][nt] += v;
                }
            lds_barrier();
        }
        // ---- epilogue on this wave's four fragments
        float t1[CA_NT][4], t2[CA_NT][4];                   // forward: statistics of the tile
#pragma unroll
        for (int nt = 0; nt < CA_NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { t1[nt][r] = 0.f; t2[nt][r] = 0.f; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int mt = KW * 4 + m;
            const int f = pw * 8 + mt;
            const int oy = strip * G_::TH + f / (TW / 16), ox = (f % (TW / 16)) * 16 + li;
            const int pix = (b * p.H + oy) * p.W + ox;
#pragma unroll
            for (int np = 0; np < CA_NT; np += 2) {
                f32x4 va = acc[mt][np], vb = acc[mt][np + 1];
                asm volatile("" : "+v"(va), "+v"(vb));
                if (p.addend) {                             // before the exchange a lane holds channels nt * 16 + lg * 4 .. + 3 of pixel li
                    const i32x2 qa = __builtin_amdgcn_raw_buffer_load_b64(adr, (pix * 64 + np * 16 + lg * 4) * 2, 0, 0);
                    const i32x2 qb = __builtin_amdgcn_raw_buffer_load_b64(adr, (pix * 64 + (np + 1) * 16 + lg * 4) * 2, 0, 0);
                    va[0] += __uint_as_float((unsigned)qa.x << 16); va[1] += __uint_as_float((unsigned)qa.x & 0xffff0000u);
                    va[2] += __uint_as_float((unsigned)qa.y << 16); va[3] += __uint_as_float((unsigned)qa.y & 0xffff0000u);
                    vb[0] += __uint_as_float((unsigned)qb.x << 16); vb[1] += __uint_as_float((unsigned)qb.x & 0xffff0000u);
                    vb[2] += __uint_as_float((unsigned)qb.y << 16); vb[3] += __uint_as_float((unsigned)qb.y & 0xffff0000u);
                }
                const i32x2 pa = {(int)f2bf_pk(va[0], va[1]), (int)f2bf_pk(va[2], va[3])}, pb = {(int)f2bf_pk(vb[0], vb[1]), (int)f2bf_pk(vb[2], vb[3])};
                if (p.stats) {
                    auto stat = [&](int nt, const i32x2& pk) {
                        f32x2_ lo = {__uint_as_float((unsigned)pk.x << 16), __uint_as_float((unsigned)pk.x & 0xffff0000u)};
                        f32x2_ hi = {__uint_as_float((unsigned)pk.y << 16), __uint_as_float((unsigned)pk.y & 0xffff0000u)};
                        f32x2_& a0 = *reinterpret_cast<f32x2_*>(&t1[nt][0]); f32x2_& a1 = *reinterpret_cast<f32x2_*>(&t1[nt][2]);
                        f32x2_& q0 = *reinterpret_cast<f32x2_*>(&t2[nt][0]); f32x2_& q1 = *reinterpret_cast<f32x2_*>(&t2[nt][2]);
                        a0 += lo; a1 += hi;
                        q0 = __builtin_elementwise_fma(lo, lo, q0); q1 = __builtin_elementwise_fma(hi, hi, q1);
                    };
                    stat(np, pa); stat(np + 1, pb);
                }
                const auto sx = __builtin_amdgcn_permlane16_swap((unsigned)pa.x, (unsigned)pb.x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap((unsigned)pa.y, (unsigned)pb.y, false, false);
                const i32x4 q = {(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
                const int c8 = (np + (lg & 1)) * 16 + (lg >> 1) * 8;               // the lane's eight channels after the exchange (conv_pw.hip)
                const int off = (pix * 64 + c8) * 2;
                __builtin_amdgcn_raw_buffer_store_b128(q, yr, off, 0, 0);
                if (p.bn_red) {
                    const i32x4 xq = __builtin_amdgcn_raw_buffer_load_b128(bxr, off, 0, 0);
                    i32x4 yq = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};           // "positive" when there is no ReLU
                    if (p.bn_y) {
                        if (p.bn_gate == 2) {                                               // byte mask: one byte per element, non-zero = passed
                            const i32x2 mq = __builtin_amdgcn_raw_buffer_load_b64(byr, off >> 1, 0, 0);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned lo = ((unsigned)(k < 2 ? mq.x : mq.y) >> (16 * (k & 1))) & 0xffu;
                                const unsigned hi = ((unsigned)(k < 2 ? mq.x : mq.y) >> (16 * (k & 1) + 8)) & 0xffu;
                                yq[k] = (lo ? 0x3f80 : 0) | (hi ? 0x3f800000 : 0);
                            }
                        } else {
                            yq = __builtin_amdgcn_raw_buffer_load_b128(byr, off, 0, 0);
                        }
                    }
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(sMean + c8), m1 = *reinterpret_cast<const f32x4*>(sMean + c8 + 4);
                    const f32x4 i0 = *reinterpret_cast<const f32x4*>(sMean + 64 + c8), i1 = *reinterpret_cast<const f32x4*>(sMean + 64 + c8 + 4);
                    const float mean[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
                    const float invs[8] = {i0[0], i0[1], i0[2], i0[3], i1[0], i1[1], i1[2], i1[3]};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const unsigned gw = (unsigned)q[k >> 1], xw = (unsigned)xq[k >> 1], yw = (unsigned)yq[k >> 1];
                        const float gv0 = (k & 1) ? __uint_as_float(gw & 0xffff0000u) : __uint_as_float(gw << 16);
                        const float xv = (k & 1) ? __uint_as_float(xw & 0xffff0000u) : __uint_as_float(xw << 16);
                        const float yv = (k & 1) ? __uint_as_float(yw & 0xffff0000u) : __uint_as_float(yw << 16);
                        const float gv = yv > 0.f ? gv0 : 0.f;
                        s1[np >> 1][k] += gv;
                        s2[np >> 1][k] += gv * (xv - mean[k]) * invs[k];
                    }
                }
            }
        }
        if (p.stats) {                                      // 16 pixel lanes (DPP) -> the workgroup's accumulators in LDS
#pragma unroll
            for (int nt = 0; nt < CA_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row_sum16(t1[nt][r]), bq = row_sum16(t2[nt][r]);
                    if (li == 0) { sAcc[wave * 128 + nt * 16 + lg * 4 + r] += a; sAcc[wave * 128 + 64 + nt * 16 + lg * 4 + r] += bq; }
                }
        }
        if constexpr (KW == 1) {
            // before the barrier that publishes the other slot: every load of this wave has landed (belt and braces, see conv_stem.hip)
#pragma unroll
            for (int d = 0; d < CA_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
        }
        lds_barrier();                                      // slot g & 1 consumed by every wave, slot (g + 1) & 1 complete
        ++g;
    }
#pragma unroll
    for (int d = 0; d < CA_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
    };
    if (kw == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    if (p.bn_red) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float u = row_sum16(s1[a][k]), v = row_sum16(s2[a][k]);
                const int c = (2 * a + (lg & 1)) * 16 + (lg >> 1) * 8 + k;
                if (li == 0) { sAcc[wave * 128 + c] += u; sAcc[wave * 128 + 64 + c] += v; }
            }
    }
    float* const dst = p.bn_red ? p.bn_red : p.stats;
    if (dst) {
        __syncthreads();
        if (t < 128) {
            const int which = t >> 6, c = t & 63;
            bn_acc_add(dst, blockIdx.x, which, 64, c, (sAcc[t] + sAcc[128 + t]) + (sAcc[256 + t] + sAcc[384 + t]));
        }
    }
}

bool g_c3a_on = getenv("DANET_NO_CONV3X3A") == nullptr;

template <int TW>
int c3a_launch(const C3aP& p, hipStream_t st) {
    static bool attr_set = false;
    static int cus = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3a_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<TW>::LDS);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr_set = true;
    }
    const int grid = p.ntiles < cus ? p.ntiles : cus;
    hipLaunchKernelGGL(conv3x3a_kernel<TW>, dim3((unsigned)grid), dim3(256), (size_t)Geo<TW>::LDS, st, p);
    return 0;
}

}  // namespace

// 3x3 / stride 1 / pad 1 / one group, 64 -> 64 channels, maps 16 wide (16 x 16 tiles: H % 16 == 0) or 64 wide (4 x 64 tiles: H % 4 == 0), at
// least 256 tiles: 1 when csrc/conv3x3a.hip takes the problem (forward: weights packed mode 0 / chunk 16; data gradient: mode 1 / chunk 16,
// `transposed` = 1), 0 when danet_conv_forward's kernels do.
extern "C" int danet_conv3x3a_ok(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int groups)
{
    if (!g_c3a_on) return 0;
    if (R != 3 || S != 3 || stride != 1 || pad != 1 || dil != 1 || groups != 1 || Cin != 64 || Cout != 64) return 0;
    if (!((W == 16 && H % 16 == 0) || (W == 64 && H % 4 == 0))) return 0;
    if ((long)B * H * W * 128 >= (1L << 31)) return 0;
    if ((long)B * H * W / 256 < 256) return 0;
    return 1;
}

// x, y: [B,H,W,64] bf16 NHWC.  transposed = 0: y = conv(x, w) and bn_sums (optional, [BN_NCOPY][2][64], pre-zeroed) receives the output's
// statistics; transposed = 1: x is dy, y is dx, wp the mode-1 packing; optional fused BatchNorm-backward sums (bn_x, bn_y, bn_saved, bn_red,
// bn_gate: 0 = bn_y is the BatchNorm's bf16 output, 2 = its byte mask) and residual addend, as danet_conv_forward's arguments of those names.
extern "C" int danet_conv3x3a(const void* x, const void* wp, void* y, int B, int H, int W, int transposed, float* bn_sums,
                              const void* bn_x, const void* bn_y, const float* bn_saved, float* bn_red, int bn_gate, const void* addend, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && wp && y, "conv3x3a: null pointer");
    DANET_CHECK_ARG(danet_conv3x3a_ok(B, H, W, 64, 64, 3, 3, 1, 1, 1, 1), "conv3x3a: unsupported problem (see danet_conv3x3a_ok)");
    DANET_CHECK_ARG(!bn_red || (bn_x && bn_saved && transposed && !bn_sums), "conv3x3a: the fused BatchNorm-backward sums belong to the data gradient and need bn_x, bn_saved");
    DANET_CHECK_ARG(bn_gate == 0 || (bn_gate == 2 && bn_red && bn_y), "conv3x3a: bn_gate %d", bn_gate);
    C3aP p{};
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wp; p.y = y; p.stats = bn_sums;
    p.bn_x = bn_red ? (const bf16_t*)bn_x : nullptr; p.bn_y = bn_red ? bn_y : nullptr; p.bn_saved = bn_saved; p.bn_red = bn_red; p.bn_gate = bn_gate;
    p.addend = (const bf16_t*)addend;
    p.B = B; p.H = H; p.W = W; p.mirrored = transposed ? 1 : 0;
    p.bytes = (int)((long)B * H * W * 128);
    if (W == 16) { p.strips = H / 16; p.ntiles = B * p.strips; c3a_launch<16>(p, (hipStream_t)stream); }
    else { p.strips = H / 4; p.ntiles = B * p.strips; c3a_launch<64>(p, (hipStream_t)stream); }
    DANET_CHECK_LAUNCH("conv3x3a_kernel");
    return DANET_OK;
}

long danet_conv::conv3x3a_knob(long enable) { const long old = g_c3a_on ? 1 : 0; if (enable >= 0) g_c3a_on = enable != 0; return old; }
