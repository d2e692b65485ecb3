// Data gradient of the 7x7 / stride-2 / pad-3 stem convolution (/root/reference/models/module/res_module.py:404 SmplResNet.conv1 over
// the 768 part crops: dy [B,32,32,64] -> dx [B,64,64,64], 315 GFLOP): the second half of csrc/conv_stem.hip's job.  The gather kernel
// (conv_fast.hip, parity classes) ran it at 15 % of the bf16 MFMA peak (843 us incl. the fused BatchNorm-backward sums).
//
//   dx[iy, ix, ci] = sum over (r, s, co) with (iy + 3 - r), (ix + 3 - s) even of dy[(iy + 3 - r) / 2, (ix + 3 - s) / 2, co] * w[co][ci][r][s]
//
// Output pixels of one parity class (py, px) = (iy & 1, ix & 1) share their tap set -- 3 or 4 rows x 3 or 4 columns, 9 / 12 / 12 / 16
// taps -- and read dy at unit stride: class pixel (i, j) = dx(2 i + py, 2 j + px) reads dy(i + dr, j + dc), dr = (py + 3 - r) / 2.
//   * a tile = the 16 dx rows x 64 columns behind 8 class rows: its dy rows (11 x 35 cells with the halo, all four 16-channel slabs:
//     49 KB) reach LDS ONCE by LDS-DMA into a two-slot ring (the next tile's travel while this one is multiplied) and serve all four
//     classes, which are computed one after the other: 256 class pixels x 64 channels each, the same wave geometry as the forward
//     kernel (2 pixel halves x 2 K halves, 8 x 4 accumulator tiles per wave in AGPRs, inline-asm MFMAs);
//   * a k-step = two taps of the class x 16 dy channels; which taps, slab and weight half-fragments is a table entry per (K half, k-step)
//     -- the weights are the chunk-16 mode-1 packing of danet_conv_pack_weights (K order: channel slab, tap, channel), of which a tap's
//     16 channels are one contiguous 512-byte half fragment, so lanes 0-31 / 32-63 fetch the halves of two arbitrary taps;
//   * the K halves are cut 12|8, 12|12, 12|12, 16|16 k-steps (multiples of the four-deep weight ring): the waves with the 8 issue the
//     next tile's copies first and wait them out in their first ring turn (the ring wait counts them: see conv_stem.hip), the K halves meet in LDS once per class;
//   * epilogue per class: 16-byte stores through v_permlane16_swap to the class's pixels, and -- when the consumer is the BatchNorm
//     that produced the stem's input -- that BatchNorm's two backward sums (sum g, sum g * xhat, gated by its ReLU) from the rounded
//     outputs, kept in registers over all tiles and flushed once per workgroup (the gather kernel's bn_red contract).
#include "common.h"
#include "conv_common.h"
#include <type_traits>

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int OOB = 0x7fffffff;
constexpr int SD_R = 7, SD_TH = 8, SD_JW = 32, SD_NT = 4, SD_MT = 8, SD_NSLAB = 4, SD_KS = 25;
constexpr int SD_ROWS = SD_TH + 3, SD_CELLS = SD_JW + 3;     // 11 dy rows x 35 cells of a tile (halo: one before, two behind)
constexpr int SD_ROWB = SD_CELLS * 32;                       // bytes per row of a 16-channel plane
constexpr int SD_PLANE = SD_ROWS * SD_ROWB;                  // 12 320
constexpr int SD_SLOT = SD_NSLAB * SD_PLANE;                 // 49 280: all 64 dy channels of a tile
constexpr int SD_PIECES = SD_SLOT / 16;                      // 3 080
constexpr int SD_NDMA = (SD_PIECES + 63) / 64;               // 49 copy instructions per tile
constexpr int SD_NCP = (SD_NDMA + 1) / 2;                    // per issuing wave
constexpr int SD_D = 4;                                      // weight ring depth (k-steps)
constexpr int SD_NE0 = 52, SD_NE1 = 48, SD_NEMAX = 56;       // k-steps per tile of the two K halves (+ wrap-around entries)
constexpr int SD_EXCH = 32768;
constexpr int SD_TABB = 2 * SD_NEMAX * 2 * 4;                // [role][entry][half] ints
constexpr int SD_OFFT = 2 * SD_NCP * 64 * 4;
constexpr int SD_LDS = 2 * SD_SLOT + SD_EXCH + 2 * SD_TABB + SD_OFFT + 512 + 4 * 512;   // + mean / invstd [2][64] + per-wave statistics accumulators [4][2][64]

struct StemDP {
    const bf16_t* dy; const bf16_t* w; void* dx;
    const bf16_t* bn_x; const bf16_t* bn_y; const float* bn_saved; float* bn_red;
    int B, H, W, OH;                                         // dx is [B, H, W, 64], dy [B, OH, 32, 64]
    int ntiles, strips;
    int dy_bytes, dx_bytes;
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

// k-steps of the classes (c = py * 2 + px) and their cut between the two K halves
__device__ __host__ constexpr int sd_pairs(int c) { return c == 0 ? 5 : (c == 3 ? 8 : 6); }
__device__ __host__ constexpr int sd_cnt(int c, int kw) { return kw == 0 ? (c == 3 ? 16 : 12) : (c == 0 ? 8 : (c == 3 ? 16 : 12)); }
__device__ __host__ constexpr int sd_base(int c, int kw) { int b = 0; for (int q = 0; q < c; ++q) b += sd_cnt(q, kw); return b; }

__global__ __launch_bounds__(256, 1) void conv_stem_dgrad_kernel(StemDP p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave & 1, kw = wave >> 1;
    unsigned char* const ring = smem;
    unsigned char* const sR = smem + 2 * SD_SLOT;
    int* const sTabA = reinterpret_cast<int*>(smem + 2 * SD_SLOT + SD_EXCH);          // [role][entry][half]: weight byte offset of the half fragment
    int* const sTabB = sTabA + 2 * SD_NEMAX * 2;                                        // ... LDS byte offset of the tap's cell relative to the pixel's
    int* const sOff = sTabB + 2 * SD_NEMAX * 2;                                         // [issuing wave][instruction][lane]
    float* const sMean = reinterpret_cast<float*>(sOff + 2 * SD_NCP * 64);              // [2][64] mean, invstd
    float* const sAcc = sMean + 128;                                                    // [wave][2][64]: a copy per wave (the order of LDS atomics is not fixed)
    // ---- k-step tables
    for (int e = t; e < 2 * SD_NEMAX * 2; e += 256) {
        const int half = e & 1, ent = (e >> 1) % SD_NEMAX, role = (e >> 1) / SD_NEMAX;
        const int ne = role == 0 ? SD_NE0 : SD_NE1;
        const int en = ent % ne;                                                        // (entries behind the tile's last wrap to its first)
        int c = 0, q = en;
        while (q >= sd_cnt(c, role)) { q -= sd_cnt(c, role); ++c; }
        const int kidx = role == 0 ? q : sd_cnt(c, 0) + q;                               // k-step of the class: (slab, tap pair)
        const int np = sd_pairs(c), slab = kidx / np, pair = kidx - slab * np;
        const int py = c >> 1, px = c & 1, ns = px ? 4 : 3, nr = py ? 4 : 3;
        int n = 2 * pair + half;
        const bool dummy = n >= nr * ns;                                                // odd tap count: zero weights (tap 49 of the packing) on any valid cell
        if (dummy) n = 2 * pair;
        const int r = (py ? 0 : 1) + 2 * (n / ns), s = (px ? 0 : 1) + 2 * (n % ns);
        const int dr = (py + 3 - r) / 2, dc = (px + 3 - s) / 2;
        const int tap = dummy ? 49 : r * SD_R + s;
        sTabA[e] = (slab * SD_KS + (tap >> 1)) * 1024 + (tap & 1) * 512;
        sTabB[e] = slab * SD_PLANE + ((1 + dr) * SD_CELLS + (1 + dc)) * 32;
    }
    // ---- the tile copy: piece q of the slot = (plane, row, cell, half): dy row (row), column (cell - 1), channels plane * 16 + half * 8 ..
    const int pixb = 128, rowb = SD_JW * pixb;
    for (int e = t; e < 2 * SD_NCP * 64; e += 256) {
        const int ln = e & 63, u = (e >> 6) % SD_NCP, half_w = (e >> 6) / SD_NCP;
        const int i = half_w + 2 * u;
        const int q = i * 64 + ln;
        const int plane = q / (SD_PLANE / 16), rem = q - plane * (SD_PLANE / 16);
        const int row = rem / (SD_ROWB / 16), r2 = rem - row * (SD_ROWB / 16);
        const int cell = r2 >> 1, half = r2 & 1;
        const int ox = cell - 1;
        int v = -1;
        if (i < SD_NDMA && q < SD_PIECES) v = ((unsigned)ox < (unsigned)SD_JW ? row * rowb + ox * pixb + plane * 32 + half * 16 : 0x00ffffff) | (row << 24);
        sOff[e] = v;
    }
    sAcc[t] = 0.f; sAcc[256 + t] = 0.f;
    if (t < 128) sMean[t] = p.bn_red ? p.bn_saved[t] : 0.f;                             // [2][64]: mean, invstd
    const i32x4 ydesc = raw_desc(p.dy, p.dy_bytes);
    int lanebase[SD_MT];
#pragma unroll
    for (int mt = 0; mt < SD_MT; ++mt) {
        const int il = pw * 4 + (mt >> 1), j = (mt & 1) * 16 + li;
        lanebase[mt] = (il * SD_CELLS + j) * 32 + (lg & 1) * 16;
    }
    constexpr int NKS = SD_NSLAB * SD_KS;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.dx, 0, p.dx_bytes, 0x00020000);
    const i32x4 wdesc = raw_desc(p.w, SD_NT * NKS * 1024);
    int wso[SD_NT];
#pragma unroll
    for (int nt = 0; nt < SD_NT; ++nt) wso[nt] = __builtin_amdgcn_readfirstlane(nt * NKS * 1024);
    const int wlane = (lane & 31) * 16;
    const int h = lg >> 1;
    bf16x8 A[SD_D][SD_NT];
    auto load_a = [&](int ao, bf16x8* a) {
        const int voff = wlane + ao;
#pragma unroll
        for (int nt = 0; nt < SD_NT; ++nt)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a[nt]) : "v"(voff), "s"(wdesc), "s"(wso[nt]) : "memory");
    };
    int g = 0;                                              // tiles completed (slot of the current tile = g & 1)
    auto issue_mine = [&](int tile, int slot) {             // the copy instructions of this (kw = 1) wave
        if (tile >= p.ntiles) return;
        const int b = tile / p.strips, strip = tile - b * p.strips;
        const int oy0 = SD_TH * strip - 1;
        const int soff = __builtin_amdgcn_readfirstlane((b * p.OH + oy0) * rowb);
        const unsigned dst0 = (unsigned)(unsigned long long)(lds_ptr_t)(ring + slot * SD_SLOT);
        const int* const mine = sOff + (pw * SD_NCP) * 64 + lane;
#pragma unroll
        for (int u = 0; u < SD_NCP; ++u) {
            const int e = mine[u * 64];
            const int row = e >> 24, off = e & 0x00ffffff;
            const bool ok = off != 0x00ffffff && (unsigned)(oy0 + row) < (unsigned)p.OH;
            if (e >= 0) dma16(dst0 + (unsigned)((pw + 2 * u) * 1024), ok ? off + soff : OOB, ydesc, 0);
        }
    };
    __syncthreads();                                        // tables written
    if (kw == 1) issue_mine(blockIdx.x, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    // per-lane BatchNorm-backward sums of its 2 x 8 channels, over every tile of the workgroup
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[a][k] = 0.f; s2[a][k] = 0.f; }
    const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_x ? p.bn_x : p.dy), 0, p.bn_x ? p.dx_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_y ? p.bn_y : p.dy), 0, p.bn_y ? p.dx_bytes : 0, 0x00020000);

    auto run = [&](auto kwc) {
    constexpr int KW = decltype(kwc)::value;
    constexpr int NE = KW == 0 ? SD_NE0 : SD_NE1;
    const int* const tA = sTabA + KW * SD_NEMAX * 2 + h;    // entry e of this K half: tA[2 e], tB[2 e]
    const int* const tB = sTabB + KW * SD_NEMAX * 2 + h;
    int pf = 0;                                             // the entry the next refill loads (D ahead of the MFMAs, wraps at the tile's end)
    auto refill = [&](bf16x8* a, int ao) {
        load_a(ao, a);
        pf = pf + 1 == NE ? 0 : pf + 1;
    };
#pragma unroll
    for (int d = 0; d < SD_D; ++d) refill(A[d], tA[2 * d]);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const unsigned char* const sX = ring + (g & 1) * SD_SLOT;
        const int b = tile / p.strips, strip = tile - b * p.strips;
        if constexpr (KW == 1) issue_mine(tile + gridDim.x, (g + 1) & 1);       // the next tile starts travelling: waited out in this wave's first ring turn
        auto do_class = [&](auto cc) {
            constexpr int C = decltype(cc)::value;
            constexpr int cnt = sd_cnt(C, KW), e0 = sd_base(C, KW), py = C >> 1, px = C & 1;
            f32x4 acc[SD_MT][SD_NT];                        // written by the class's first k-step (C = 0)
            bf16x8 Bq[2][SD_MT];
            auto load_b = [&](int to, bf16x8* bq) {
#pragma unroll
                for (int mt = 0; mt < SD_MT; ++mt) bq[mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + to);
            };
            int tnext = tB[2 * (e0 + 1)];                   // tap cell of the k-step after the next fragment read (one k-step ahead of its use)
            load_b(tB[2 * e0], Bq[0]);
#define SD_KSTEP(a, bc, bn, e, more, first) do { \
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]) : "n"(4 * (SD_D - 1))); \
                const int ao_ = tA[2 * pf]; \
                if (more) { load_b(tnext, bn); tnext = tB[2 * ((e) + 2)]; } \
                if (first) { \
                    _Pragma("unroll") for (int nt = 0; nt < SD_NT; ++nt) \
                        _Pragma("unroll") for (int mt = 0; mt < SD_MT; ++mt) \
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"((a)[nt]), "v"((bc)[mt])); \
                } else { \
                    _Pragma("unroll") for (int nt = 0; nt < SD_NT; ++nt) \
                        _Pragma("unroll") for (int mt = 0; mt < SD_MT; ++mt) \
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"((a)[nt]), "v"((bc)[mt])); \
                } \
                refill(a, ao_); \
                __builtin_amdgcn_sched_barrier(0); \
            } while (0)
#pragma unroll
            for (int q = 0; q < cnt; ++q) SD_KSTEP(A[q % SD_D], Bq[q & 1], Bq[(q + 1) & 1], e0 + q, q + 1 < cnt, q == 0);
#undef SD_KSTEP
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // (inline-asm MFMAs: the hazard recogniser does not see them)
#ifdef SD_DRAIN
#pragma unroll
            for (int d = 0; d < SD_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
#endif
            // ---- the two K halves meet in LDS: wave (pw, 0) finishes fragments 0-3, wave (pw, 1) fragments 4-7
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                unsigned char* const mine = sR + wave * 8192 + lane * 16;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    constexpr int base = (1 - KW) * 4;
#pragma unroll
                    for (int nt = 0; nt < SD_NT; ++nt) {
                        f32x4 v = acc[base + round * 2 + m][nt];
                        asm volatile("" : "+v"(v));
                        *reinterpret_cast<f32x4*>(mine + (m * SD_NT + nt) * 1024) = v;
                    }
                }
                lds_barrier();
                const unsigned char* const theirs = sR + (wave ^ 2) * 8192 + lane * 16;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nt = 0; nt < SD_NT; ++nt) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + (m * SD_NT + nt) * 1024);
                        acc[KW * 4 + round * 2 + m][nt] += v;
                    }
                lds_barrier();
            }
            // ---- epilogue on this wave's four fragments of the class
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int mt = KW * 4 + m;
                const int il = pw * 4 + (mt >> 1), j = (mt & 1) * 16 + li;
                const int iy = 2 * (strip * SD_TH + il) + py, ix = 2 * j + px;
                const int pix = (b * p.H + iy) * p.W + ix;
#pragma unroll
                for (int np = 0; np < SD_NT; np += 2) {
                    f32x4 va = acc[mt][np], vb = acc[mt][np + 1];
                    asm volatile("" : "+v"(va), "+v"(vb));
                    const i32x2 pa = {(int)f2bf_pk(va[0], va[1]), (int)f2bf_pk(va[2], va[3])}, pb = {(int)f2bf_pk(vb[0], vb[1]), (int)f2bf_pk(vb[2], vb[3])};
                    const auto sx = __builtin_amdgcn_permlane16_swap((unsigned)pa.x, (unsigned)pb.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap((unsigned)pa.y, (unsigned)pb.y, false, false);
                    const i32x4 q = {(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
                    const int c8 = (np + (lg & 1)) * 16 + (lg >> 1) * 8;           // the lane's eight channels after the exchange (conv_pw.hip)
                    const int off = (pix * 64 + c8) * 2;
                    __builtin_amdgcn_raw_buffer_store_b128(q, xr, off, 0, 0);
                    if (p.bn_red) {
                        const i32x4 xq = __builtin_amdgcn_raw_buffer_load_b128(bxr, off, 0, 0);
                        i32x4 yq = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};       // "positive" when there is no ReLU
                        if (p.bn_y) yq = __builtin_amdgcn_raw_buffer_load_b128(byr, off, 0, 0);
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
        };
        do_class(std::integral_constant<int, 0>{});
        do_class(std::integral_constant<int, 1>{});
        do_class(std::integral_constant<int, 2>{});
        do_class(std::integral_constant<int, 3>{});
        if constexpr (KW == 1) {
            // the next tile's copies were requested ~48 k-steps ago and every ring wait since has counted them; before the barrier that
            // publishes the slot, wait for everything all the same (belt and braces: in-order return is observed -- tools/experiments/
            // dma_order.hip -- not documented; one refill latency per tile)
#pragma unroll
            for (int d = 0; d < SD_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
        }
        lds_barrier();                                      // slot g & 1 consumed by every wave, slot (g + 1) & 1 complete
        ++g;
    }
    // the ring's refills past the last tile are still in flight
#pragma unroll
    for (int d = 0; d < SD_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
    };
    if (kw == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    if (p.bn_red) {                                         // 16 pixel lanes (DPP) -> the workgroup's accumulators in LDS -> one replica
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float u = row_sum16(s1[a][k]), v = row_sum16(s2[a][k]);
                const int c = (2 * a + (lg & 1)) * 16 + (lg >> 1) * 8 + k;
                if (li == 0) { sAcc[wave * 128 + c] += u; sAcc[wave * 128 + 64 + c] += v; }
            }
        __syncthreads();
        if (t < 128) {
            const int which = t >> 6, c = t & 63;
            bn_acc_add(p.bn_red, blockIdx.x, which, 64, c, (sAcc[t] + sAcc[128 + t]) + (sAcc[256 + t] + sAcc[384 + t]));
        }
    }
}

bool g_stem_dgrad_on = getenv("DANET_NO_CONV_STEM_DGRAD") == nullptr;

}  // namespace

// The data gradient of a convolution danet_conv_stem_ok takes, with 64 input and 64 output channels: dims as for danet_conv_forward
// with transposed = 1 turned around -- (B, H, W, Cin) describe dx (the convolution's input), (OH, OW, Cout) dy.  1: this kernel takes it
// (weights: danet_conv_pack_weights mode 1, chunk 16), 0: danet_conv_forward(transposed = 1) does.
extern "C" int danet_conv_stem_dgrad_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups)
{
    if (!g_stem_dgrad_on) return 0;
    if (R != SD_R || S != SD_R || stride != 2 || pad != 3 || dil != 1 || groups != 1) return 0;
    if (Cout != 16 * SD_NSLAB || Cin != 16 * SD_NT) return 0;
    if (OW != SD_JW || W != 2 * SD_JW || OH % SD_TH != 0 || H != 2 * OH) return 0;
    if ((long)B * H * W * Cin * 2 >= (1L << 31) || (long)B * OH * OW * Cout * 2 >= (1L << 30)) return 0;
    if ((long)B * (OH / SD_TH) < 256) return 0;
    return 1;
}

// dy [B,OH,OW,Cout] bf16, wp = danet_conv_pack_weights(..., mode 1, chunk 16), dx [B,H,W,Cin] bf16.  Optional fused BatchNorm-backward sums of
// the BatchNorm that produced the convolution's input (the contract of danet_conv_forward's bn_* arguments): bn_x = that BatchNorm's input,
// bn_y = its output when it ends in a ReLU (NULL otherwise), bn_saved [2][Cin] = mean, invstd, bn_red [BN_NCOPY][2][Cin] (pre-zeroed)
// receives sum g and sum g * xhat over the rounded dx.  All four NULL: no reduction.
extern "C" int danet_conv_stem_dgrad(const void* dy, const void* wp, void* dx, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                     const void* bn_x, const void* bn_y, const float* bn_saved, float* bn_red, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(dy && wp && dx, "conv_stem_dgrad: null pointer");
    DANET_CHECK_ARG(danet_conv_stem_dgrad_ok(B, H, W, Cin, OH, OW, Cout, SD_R, SD_R, 2, 3, 1, 1), "conv_stem_dgrad: unsupported problem (see danet_conv_stem_dgrad_ok)");
    DANET_CHECK_ARG(!bn_red || (bn_x && bn_saved), "conv_stem_dgrad: the fused BatchNorm-backward sums need bn_x and bn_saved");
    StemDP p{};
    p.dy = (const bf16_t*)dy; p.w = (const bf16_t*)wp; p.dx = dx;
    p.bn_x = bn_red ? (const bf16_t*)bn_x : nullptr; p.bn_y = bn_red ? (const bf16_t*)bn_y : nullptr; p.bn_saved = bn_saved; p.bn_red = bn_red;
    p.B = B; p.H = H; p.W = W; p.OH = OH;
    p.strips = OH / SD_TH; p.ntiles = B * p.strips;
    p.dy_bytes = (int)((long)B * OH * OW * Cout * 2); p.dx_bytes = (int)((long)B * H * W * Cin * 2);
    static bool attr_set = false;
    static int cus = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr_set = true;
    }
    const int grid = p.ntiles < cus ? p.ntiles : cus;
    hipLaunchKernelGGL(conv_stem_dgrad_kernel, dim3((unsigned)grid), dim3(256), (size_t)SD_LDS, (hipStream_t)stream, p);
    DANET_CHECK_LAUNCH("conv_stem_dgrad_kernel");
    return DANET_OK;
}

long danet_conv::conv_stem_dgrad_knob(long enable) { const long old = g_stem_dgrad_on ? 1 : 0; if (enable >= 0) g_stem_dgrad_on = enable != 0; return old; }
