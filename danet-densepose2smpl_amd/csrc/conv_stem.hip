// 7x7 / stride-2 / pad-3 stem convolution, forward: the regressor ResNets' first layer (/root/reference/models/module/res_module.py:404
// SmplResNet.conv1, :118) over the 768 part crops of a 32-image batch -- 64 -> 64 channels on 64x64 maps, 315 GFLOP, the largest
// single layer of the step.  The gather kernel (conv_fast.hip) ran it at 17 % of the bf16 MFMA peak; shared weights through LDS
// or 128-pixel wave tiles did not move it (measured: 733 / 691 us against 735): its limit is the B-operand gather itself -- a stride-2
// fragment load touches 16 pixels 256 bytes apart, 64 useful bytes per cache line visit, every input value 12 times.
// Here the input reaches LDS ONCE per tile, in full rows, and the MFMAs read it from there:
//   * a tile = 8 output rows x 32 columns (256 pixels) of one image x all 64 output channels.  Its input -- 21 rows x 70 columns
//     (3-pixel halo), one 16-channel slab at a time -- is copied by LDS-DMA (`buffer_load_dwordx4 ... lds`, per-lane source
//     offsets, out-of-range lanes write zeros = padding) into a two-slot ring; slab s + 1 travels while slab s is multiplied,
//     across tile boundaries as well (workgroups are persistent: one per compute unit, 94 KB of LDS);
//   * the LDS image keeps EVEN and ODD input columns in separate planes ([row][plane][column / 2][2 x 16 bytes]): the 16 output pixels
//     of a fragment read 16 consecutive 32-byte cells of one plane -- the same conflict-free pattern as the 3x3 kernels -- instead
//     of cells 64 bytes apart;
//   * a k-step (32 K values) = two filter taps x 16 channels, in the chunked packing of danet_conv_pack_weights (chunk = 16: K order
//     channel slab, tap, channel): lanes 0-31 take tap 2j, lanes 32-63 tap 2j + 1, a tap being an LDS byte offset from a 50-entry table;
//   * the four waves are 2 (pixel halves of the tile: 128 pixels = 8 fragments each) x 2 (halves of a slab's 25 k-steps): a wave
//     keeps 8 x 4 accumulator tiles (AGPRs), so a weight fragment (1 KB) feeds 8 MFMAs and a pixel fragment 4 -- 12 loads per 32
//     MFMAs, the layer's weights cross L2 once per 256 pixels (1.2 GB per launch instead of 4.9);
//   * the two K halves meet in LDS (two rounds of 32 KB), then the epilogue: BatchNorm statistics from the bf16-rounded outputs
//     (per lane over all tiles, flushed once per workgroup), 16-byte stores through v_permlane16_swap.
#include "common.h"
#include "conv_common.h"
#include <type_traits>

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int OOB = 0x7fffffff;
constexpr int ST_R = 7, ST_TAPS = 49, ST_TH = 8, ST_OW = 32, ST_NT = 4, ST_MT = 8;
constexpr int ST_ROWS = 2 * ST_TH + 5;                      // 21 input rows of a tile
constexpr int ST_CELLS = 35;                                // cells per plane: input columns -3 .. 66 in two planes
constexpr int ST_ROWB = 2 * ST_CELLS * 32;                  // bytes per input row in LDS (two planes of 32-byte cells)
constexpr int ST_SLOT = ST_ROWS * ST_ROWB;                  // 47 040 bytes: one 16-channel slab of a tile
constexpr int ST_PIECES = ST_SLOT / 16;                     // 2 940 sixteen-byte pieces
constexpr int ST_NDMA = (ST_PIECES + 63) / 64;              // 46 copy instructions per slab
constexpr int ST_KS = 25;                                   // k-steps per slab: roundup(49 * 16, 32) / 32
constexpr int ST_TAB = 64 * 4;                              // tap table (bytes)
constexpr int ST_NCP = (ST_NDMA + 1) / 2;                   // copy instructions per issuing wave
constexpr int ST_OFFT = 2 * ST_NCP * 64 * 4;                // per-lane source offsets of the copy instructions (two issuing waves)
constexpr int ST_LDS = 2 * ST_SLOT + ST_TAB + ST_OFFT + 4 * 512;  // + per-wave, per-channel statistics accumulators [4][2][64]

struct StemP {
    const bf16_t* x; const bf16_t* w; void* y; float* stats;
    int B, H, W, Cin, Cout, OH;
    int nslab, ntiles, strips;                               // Cin / 16; B * strips; OH / 8
    int x_bytes, y_bytes;
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

#ifndef ST_STRICT_COPIES
#define ST_STRICT_COPIES 1
#endif


#ifdef ST_DIAG        // tools/experiments/stem_diag.hip: cycles per wave in the k-steps / at the slab barrier / in the exchange / in the epilogue
__device__ unsigned long long g_stem_diag[256 * 4 * 8];
#define ST_CLK(v) v = __builtin_readcyclecounter()
#else
#define ST_CLK(v) ((void)0)
#endif

__global__ __launch_bounds__(256, 1) void conv_stem_kernel(StemP p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave & 1, kw = wave >> 1;
    unsigned char* const ring = smem;
    int* const sTap = reinterpret_cast<int*>(smem + 2 * ST_SLOT);
    if (t < 64) {                                           // tap t: LDS byte offset of its cell relative to the output pixel's (tap 49: any valid cell, its weights are zero)
        const int tp = t < ST_TAPS ? t : ST_TAPS - 1;
        const int r = tp / ST_R, s = tp - r * ST_R;
        sTap[t] = ((r * 2 + (s & 1)) * ST_CELLS + (s >> 1)) * 32;
    }
    const int nslab = p.nslab;
    const i32x4 xdesc = raw_desc(p.x, p.x_bytes);
    const int pixb = p.Cin * 2, rowb = p.W * pixb;
    // ---- the slab copy: piece q = i * 64 + lane of the slot = (row, plane, cell, half): source = input pixel (2 cell + plane - 3) of input
    // row (row), channels 8 half .. + 7 of the slab.  What a lane copies does not depend on the tile: offsets relative to the tile's first
    // input row, and the row index for the bounds test of the image's top and bottom strips.
    int* const sOff = reinterpret_cast<int*>(smem + 2 * ST_SLOT + ST_TAB);          // [pixel half][instruction][lane]: offset | row << 24, -1 = nothing to copy
    float* const sAcc = reinterpret_cast<float*>(smem + 2 * ST_SLOT + ST_TAB + ST_OFFT);   // [wave][2][64] statistics of the workgroup's tiles (a copy per wave: the order of LDS atomics is not fixed)
    for (int e = t; e < 2 * ST_NCP * 64; e += 256) {
        const int ln = e & 63, u = (e >> 6) % ST_NCP, half_w = (e >> 6) / ST_NCP;
        const int i = half_w + 2 * u;
        const int q = i * 64 + ln;
        const int row = q / (ST_ROWB / 16), rem = q - row * (ST_ROWB / 16);
        const int plane = rem / (2 * ST_CELLS), r2 = rem - plane * (2 * ST_CELLS);
        const int cell = r2 >> 1, half = r2 & 1;
        const int ix = 2 * cell + plane - 3;
        int v = -1;                                           // beyond the slot's last piece: the lane is masked out (an out-of-range lane still WRITES zeros)
        if (i < ST_NDMA && q < ST_PIECES) v = ((unsigned)ix < (unsigned)p.W ? row * rowb + ix * pixb + half * 16 : 0x00ffffff) | (row << 24);
        sOff[e] = v;
    }
    sAcc[t] = 0.f; sAcc[256 + t] = 0.f;
    // ---- per-lane pixel geometry: fragment mt = output row pw * 4 + (mt >> 1), columns (mt & 1) * 16 + li
    int lanebase[ST_MT];
#pragma unroll
    for (int mt = 0; mt < ST_MT; ++mt) {
        const int oyl = pw * 4 + (mt >> 1), ox = (mt & 1) * 16 + li;
        lanebase[mt] = (2 * oyl * 2 * ST_CELLS + ox) * 32 + (lg & 1) * 16;
    }
    const int nks_tot = nslab * ST_KS;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int wlane = lane * 16;

    // ---- weight fragments ride a hand-counted ring of D = 5 k-steps (inline asm: the compiler's own bookkeeping would drain it at
    // every loop header; vmcnt returns in order, so "slot d is complete" = "at most 4 (D - 1) younger loads outstanding").  A wave
    // alone on its SIMD has nobody to hide an L2 round trip (~1.5 k cycles) behind: with a one-step prefetch a k-step (32 MFMAs, 512
    // cycles) took ~1.7 k cycles.  The K split is 15 | 10 k-steps (= 3 | 2 ring turns, no ragged ends) and the waves with 10 issue the
    // slab copies: a copy is a long-latency entry in THEIR queue, which their next ring wait has to sit out; the 15-step waves never
    // see it, and the 5 k-steps the copying waves do less are the time that wait may take.
    constexpr int D = 5;                      // copy instructions per issuing wave (pixel half pw takes instructions pw, pw + 2, ...)
    const i32x4 wdesc = raw_desc(p.w, ST_NT * nks_tot * 1024);
    int wso[ST_NT];
#pragma unroll
    for (int nt = 0; nt < ST_NT; ++nt) wso[nt] = __builtin_amdgcn_readfirstlane(nt * nks_tot * 1024);
    bf16x8 A[D][ST_NT];
    auto load_a = [&](int ks, bf16x8* a) {                  // ks: k-step among all slabs (slab * 25 + j)
        const int voff = wlane + ks * 1024;
#pragma unroll
        for (int nt = 0; nt < ST_NT; ++nt)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a[nt]) : "v"(voff), "s"(wdesc), "s"(wso[nt]) : "memory");
    };
    // the wave's k-steps of slab s: [kbeg(s), kbeg(s) + nk) with nk = 15 (kw 0) or 10 (kw 1)
    int g = 0;                                              // stages completed (slot of the current stage = g & 1)
    auto issue_mine = [&](int tile, int slab, int slot) {   // the copy instructions of this (kw = 1) wave; false: nothing left to copy
        if (tile >= p.ntiles) return false;
        const int b = tile / p.strips, strip = tile - b * p.strips;
        const int iy0 = 2 * ST_TH * strip - 3;
        const int soff = __builtin_amdgcn_readfirstlane((b * p.H + iy0) * rowb + slab * 32);      // (negative for the first image's top strip: those rows are masked)
        const unsigned dst0 = (unsigned)(unsigned long long)(lds_ptr_t)(ring + slot * ST_SLOT);
        const int* const mine = sOff + (pw * ST_NCP) * 64 + lane;
#pragma unroll
        for (int u = 0; u < ST_NCP; ++u) {
            const int e = mine[u * 64];
            const int row = e >> 24, off = e & 0x00ffffff;
            const bool ok = off != 0x00ffffff && (unsigned)(iy0 + row) < (unsigned)p.H;
            if (e >= 0) dma16(dst0 + (unsigned)((pw + 2 * u) * 1024), ok ? off + soff : OOB, xdesc, 0);
        }
        return true;
    };
    __syncthreads();                                        // tables written
    if (kw == 1) issue_mine(blockIdx.x, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                          // tap table + first slab published
    // The two K halves run their own instantiation of everything below (KW static): accumulator tiles are then indexed statically in
    // the exchange and the epilogue (run-time selects between accumulator registers cost copies of both candidates).
    auto run = [&](auto kwc) {
    constexpr int KW = decltype(kwc)::value;
    constexpr int nk = KW == 0 ? 15 : 10, kb0 = KW == 0 ? 0 : 15;
    int pf_slab = 0, pf_j = 0;                              // prefetch cursor: the k-step the NEXT refill loads (D ahead of the MFMAs)
    [[maybe_unused]] unsigned long long dk = 0, db = 0, dx = 0, de = 0, c0 = 0, c1 = 0, life0 = 0;
    ST_CLK(life0);
    auto refill = [&](bf16x8* a) {
        load_a(pf_slab * ST_KS + kb0 + pf_j, a);
        if (++pf_j == nk) { pf_j = 0; if (++pf_slab == nslab) pf_slab = 0; }
    };
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        f32x4 acc[ST_MT][ST_NT];                            // written by the tile's first k-step (C = 0): zeros in 128 VGPRs first would spill
        pf_slab = 0; pf_j = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) refill(A[d]);           // the ring starts a tile full (drained before the epilogue, see below)
        for (int slab = 0; slab < nslab; ++slab) {
            ST_CLK(c0);
            const unsigned char* const sX = ring + (g & 1) * ST_SLOT;
            const int jb = kb0;                             // first k-step of this wave in the slab
            bf16x8 Bq[2][ST_MT];
            auto load_b = [&](int j, bf16x8* bq) {
                const int to = sTap[2 * j + (lg >> 1)];
#pragma unroll
                for (int mt = 0; mt < ST_MT; ++mt) bq[mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + to);
            };
            // one k-step on ring slot a: wait for it, request the next k-step's pixel fragments, multiply, refill the slot
#define ST_KSTEP(WAITN, a, bc, bn, j, more, first) do { \
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]) : "n"(WAITN)); \
                if (more) load_b((j) + 1, bn); \
                if (first) { \
                    _Pragma("unroll") for (int nt = 0; nt < ST_NT; ++nt) \
                        _Pragma("unroll") for (int mt = 0; mt < ST_MT; ++mt) \
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"((a)[nt]), "v"((bc)[mt])); \
                } else { \
                    _Pragma("unroll") for (int nt = 0; nt < ST_NT; ++nt) \
                        _Pragma("unroll") for (int mt = 0; mt < ST_MT; ++mt) \
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"((a)[nt]), "v"((bc)[mt])); \
                } \
                refill(a); \
                __builtin_amdgcn_sched_barrier(0);                  /* nothing moves across a k-step: the scheduler would hoist later fragment reads (registers) */ \
            } while (0)
            const bool first = slab == 0;                   // the tile's first k-step of this wave initialises the accumulators
            if constexpr (KW == 0) {
                load_b(jb, Bq[0]);
                // ten k-steps with static ring slots and buffer parity, then five
#pragma unroll
                for (int q = 0; q < 10; ++q) ST_KSTEP(4 * (D - 1), A[q % D], Bq[q & 1], Bq[(q + 1) & 1], jb + q, true, q == 0 && first);
#pragma unroll
                for (int q = 0; q < 5; ++q) ST_KSTEP(4 * (D - 1), A[q], Bq[q & 1], Bq[(q + 1) & 1], jb + 10 + q, q < 4, false);
            } else {
                // the next stage (next slab, or the next tile's first one) starts travelling into the other slot: requested here, behind the
                // ring's loads in flight
                const bool copying = slab + 1 < nslab ? issue_mine(tile, slab + 1, (g + 1) & 1) : issue_mine(tile + gridDim.x, 0, (g + 1) & 1);
                (void)copying;
                load_b(jb, Bq[0]);                          // (after the copies were requested: nothing is carried across their issue)
                // The copies just requested sit in this wave's queue behind the ring's 20 loads; the ring wait below (at most 16 outstanding)
                // therefore also waits most of them out -- the 5 k-steps (~2.5 k cycles) this wave does less than the other K half absorb that.
                // Stepping over them (vmcnt(16 + copies) when copies were requested, vmcnt(16) in the last stage) was tried and gave NaNs:
                // NOT because loads return out of order (tools/experiments/dma_order.hip: 0 violations in 196 k wave-rounds, this very
                // sequence included) but because the two copies of the unrolled ring turn under the run-time branch took the kernel to
                // 256 + 256 registers and 168 bytes of scratch: the compiler then moves and spills registers that inline-asm loads are
                // still in flight to, which it cannot know.  One code path, no spills (200 VGPRs + 128 AGPRs, no scratch).
#pragma unroll
                for (int q = 0; q < 5; ++q) ST_KSTEP(4 * (D - 1), A[q], Bq[q & 1], Bq[(q + 1) & 1], jb + q, true, q == 0 && first);
#pragma unroll
                for (int q = 5; q < 10; ++q) ST_KSTEP(4 * (D - 1), A[q - 5], Bq[q & 1], Bq[(q + 1) & 1], jb + q, q < 9, false);
            }
#undef ST_KSTEP
#if ST_STRICT_COPIES
            if constexpr (KW == 1) {
                // before the barrier that publishes the other slot: every load of this copying wave has landed (belt and braces: the ring waits
                // since the copies already imply it if loads return in order, which tools/experiments/dma_order.hip observes but no document
                // at hand states; +1 %)
#pragma unroll
                for (int d = 0; d < D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
            }
#endif
            ST_CLK(c1);
#ifdef ST_DIAG
            dk += c1 - c0;
#endif
            lds_barrier();                                        // slot g & 1 consumed; the copying waves have waited their copies out (second ring turn)
            ++g;
#ifdef ST_DIAG
            ST_CLK(c0); db += c0 - c1;
#endif
        }
        ST_CLK(c0);
        // (the MFMAs are inline asm with AGPR accumulators -- left to the compiler 128 accumulator registers + the 80-register ring + 64
        // pixel-fragment registers were juggled between the two files: 900 v_accvgpr moves, 376 bytes of scratch -- so the hazard
        // recogniser does not see them: the last results are read 2 x 16 wait states later, well past a 8-pass MFMA)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        // the ring's refills past the tile's end are still in flight: nobody needs them, but the compiler knows nothing of them and
        // would hand their registers to the exchange / epilogue below
#pragma unroll
        for (int d = 0; d < D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]), "+v"(A[d][3]));
        // ---- the two K halves of a pixel half meet in LDS: wave (pw, 0) finishes fragments 0-3, wave (pw, 1) fragments 4-7; two rounds of
        // two fragments x 4 blocks per wave (8 KB each, 32 KB in all) through the slot consumed last (the other one is receiving the next
        // tile's first slab)
        unsigned char* const sR = ring + ((g + 1) & 1) * ST_SLOT;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            unsigned char* const mine = sR + wave * 8192 + lane * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                constexpr int base = (1 - KW) * 4;                   // what the OTHER K half finishes
#pragma unroll
                for (int nt = 0; nt < ST_NT; ++nt) {
                    f32x4 v = acc[base + round * 2 + m][nt];
                    asm volatile("" : "+v"(v));                       // (through a VGPR: see DESIGN.md, lessons -- stores straight from accumulator registers)
                    *reinterpret_cast<f32x4*>(mine + (m * ST_NT + nt) * 1024) = v;
                }
            }
            lds_barrier();
            const unsigned char* const theirs = sR + (wave ^ 2) * 8192 + lane * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < ST_NT; ++nt) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + (m * ST_NT + nt) * 1024);
                    acc[KW * 4 + round * 2 + m][nt] += v;
                }
            lds_barrier();
        }
#ifdef ST_DIAG
        ST_CLK(c1); dx += c1 - c0;
#endif
        // ---- epilogue on this wave's four fragments
        const int b = tile / p.strips, strip = tile - b * p.strips;
        const int oy0 = strip * ST_TH;
        float s1[ST_NT][4], s2[ST_NT][4];                   // statistics of the tile (they live in registers only here: the k-loop needs the room)
#pragma unroll
        for (int nt = 0; nt < ST_NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // (fragment index within the wave's pixel half is static per K half: two instantiations of the loop body)
            auto body = [&](const int mt) {
                const int oy = oy0 + pw * 4 + (mt >> 1), ox = (mt & 1) * 16 + li;
                const int pix = (b * p.OH + oy) * ST_OW + ox;
#pragma unroll
                for (int np = 0; np < ST_NT; np += 2) {
                    f32x4 va = acc[mt][np], vb = acc[mt][np + 1];
                    asm volatile("" : "+v"(va), "+v"(vb));
                    const i32x2 pa = {(int)f2bf_pk(va[0], va[1]), (int)f2bf_pk(va[2], va[3])}, pb = {(int)f2bf_pk(vb[0], vb[1]), (int)f2bf_pk(vb[2], vb[3])};
                    if (p.stats) {
                        auto stat = [&](int nt, const i32x2& pk) {
                            f32x2_ lo = {__uint_as_float((unsigned)pk.x << 16), __uint_as_float((unsigned)pk.x & 0xffff0000u)};
                            f32x2_ hi = {__uint_as_float((unsigned)pk.y << 16), __uint_as_float((unsigned)pk.y & 0xffff0000u)};
                            f32x2_& a0 = *reinterpret_cast<f32x2_*>(&s1[nt][0]); f32x2_& a1 = *reinterpret_cast<f32x2_*>(&s1[nt][2]);
                            f32x2_& q0 = *reinterpret_cast<f32x2_*>(&s2[nt][0]); f32x2_& q1 = *reinterpret_cast<f32x2_*>(&s2[nt][2]);
                            a0 += lo; a1 += hi;
                            q0 = __builtin_elementwise_fma(lo, lo, q0); q1 = __builtin_elementwise_fma(hi, hi, q1);
                        };
                        stat(np, pa); stat(np + 1, pb);
                    }
                    const auto sx = __builtin_amdgcn_permlane16_swap((unsigned)pa.x, (unsigned)pb.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap((unsigned)pa.y, (unsigned)pb.y, false, false);
                    const i32x4 q = {(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
                    const int c8 = (np + (lg & 1)) * 16 + (lg >> 1) * 8;           // the lane's eight channels after the exchange (conv_pw.hip)
                    __builtin_amdgcn_raw_buffer_store_b128(q, yr, (pix * p.Cout + c8) * 2, 0, 0);
                }
            };
            body(KW * 4 + m);
        }
        if (p.stats) {                                      // 16 pixel lanes (DPP) -> the workgroup's accumulators in LDS
#pragma unroll
            for (int nt = 0; nt < ST_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row_sum16(s1[nt][r]), bq = row_sum16(s2[nt][r]);
                    if (li == 0) { sAcc[wave * 128 + nt * 16 + lg * 4 + r] += a; sAcc[wave * 128 + 64 + nt * 16 + lg * 4 + r] += bq; }
                }
        }
#ifdef ST_DIAG
        ST_CLK(c0); de += c0 - c1;
#endif
    }
#ifdef ST_DIAG
    ST_CLK(c1);
    if (lane == 0) {
        unsigned long long* d = g_stem_diag + (blockIdx.x * 4 + wave) * 8;
        d[0] = dk; d[1] = db; d[2] = dx; d[3] = de; d[4] = c1 - life0;
    }
#endif
    };
    if (kw == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    if (p.stats) {
        __syncthreads();
        if (t < 128) {
            const int which = t >> 6, c = t & 63;
            bn_acc_add(p.stats, blockIdx.x, which, p.Cout, c, (sAcc[t] + sAcc[128 + t]) + (sAcc[256 + t] + sAcc[384 + t]));
        }
    }
}

bool g_stem_on = getenv("DANET_NO_CONV_STEM") == nullptr;

}  // namespace

// 7x7 / stride 2 / pad 3 / one group, 64 output channels, input channels in 16-channel slabs, 64 x 64 -> 32 x 32 maps (the regressor
// stems); forward only.  1: csrc/conv_stem.hip takes it (weights packed with chunk = 16), 0: danet_conv_forward's kernels do.
extern "C" int danet_conv_stem_ok(int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups)
{
    if (!g_stem_on) return 0;
    if (R != ST_R || S != ST_R || stride != 2 || pad != 3 || dil != 1 || groups != 1) return 0;
    if (Cout != 16 * ST_NT || Cin % 16 != 0 || Cin < 16 || Cin > 256) return 0;
    if (OW != ST_OW || W != 2 * ST_OW || OH % ST_TH != 0 || H != 2 * OH) return 0;
    if ((long)B * H * W * Cin * 2 >= (1L << 30) || (long)B * OH * OW * Cout * 2 >= (1L << 31)) return 0;
    if ((long)B * (OH / ST_TH) < 256) return 0;                 // (fewer tiles than compute units: the gather kernel's grid fills the chip better)
    return 1;
}

// x [B,H,W,Cin] bf16, wp = danet_conv_pack_weights(..., mode 0, chunk 16), y [B,OH,OW,Cout] bf16; bn_sums: optional [BN_NCOPY][2][Cout] fused
// BatchNorm statistics (pre-zeroed).  Enable / disable at run time: danet_conv_stem_set (A-B timing, tests).
extern "C" int danet_conv_stem_forward(const void* x, const void* wp, void* y, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                       float* bn_sums, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(x && wp && y, "conv_stem_forward: null pointer");
    DANET_CHECK_ARG(danet_conv_stem_ok(B, H, W, Cin, OH, OW, Cout, ST_R, ST_R, 2, 3, 1, 1), "conv_stem_forward: unsupported problem (see danet_conv_stem_ok)");
    StemP p{};
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wp; p.y = y; p.stats = bn_sums;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.OH = OH;
    p.nslab = Cin / 16; p.strips = OH / ST_TH; p.ntiles = B * p.strips;
    p.x_bytes = (int)((long)B * H * W * Cin * 2); p.y_bytes = (int)((long)B * OH * OW * Cout * 2);
    static bool attr_set = false;
    static int cus = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr_set = true;
    }
    const int grid = p.ntiles < cus ? p.ntiles : cus;
    hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)grid), dim3(256), (size_t)ST_LDS, (hipStream_t)stream, p);
    DANET_CHECK_LAUNCH("conv_stem_kernel");
    return DANET_OK;
}

long danet_conv::conv_stem_knob(long enable) {
    const long prev = g_stem_on ? 1 : 0;
    if (enable >= 0) g_stem_on = enable != 0;
    return prev;
}
