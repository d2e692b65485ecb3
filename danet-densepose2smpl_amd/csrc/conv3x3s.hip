// 3x3 / stride-1 / pad-1 convolution (forward and data gradient), streamed: the successor of conv3x3.hip's tile kernel
// for the HRNet branch layers (/root/reference/models/module/hr_module.py:15-179, res_module.py:27-56).
//
// conv3x3.hip runs a tile as load -> MFMA -> store, serially, and all workgroups are in the same phase at once: memory
// and matrix cores take turns (16 % of the bf16 MFMA rate in round 2).  Here a workgroup has FIVE waves:
//   * wave 4 is a loader.  It copies the input of the NEXT stage into the other half of a two-deep LDS ring with
//     `buffer_load_dwordx4 ... lds` (LDS-DMA: no staging registers, no ds_write pass; lanes whose source is outside the
//     image get an out-of-range offset and the hardware writes zeros, so halo columns, halo rows outside the image and
//     padding cells are zero-filled by the same instructions -- measured in tools/experiments/lds_dma.hip: one loader
//     wave per workgroup with one 38 KB tile in flight streams at the chip's full fabric rate, 6.9 TB/s).  Its own
//     vmcnt queue holds nothing but these copies, which is what a prefetch issued by the MFMA waves could not have
//     (their weight loads would queue behind it: in-order return).
//   * waves 0-3 run the k-loop of conv3x3.hip on the CURRENT stage: pixel fragments from the LDS tile by tap offsets,
//     weight fragments through a register ring from global memory, 16x16x32 bf16 MFMAs.
// A stage is a halo tile restricted to a range of input channels; a tile whose input does not fit one ring slot
// (96 / 192 / 384 channels) is a sequence of stages that share the accumulators -- the (tap, channel) k-steps of the
// UNCHANGED weight packing are visited stage by stage through a table.  The stage sequence of a workgroup runs across
// tiles and across the problems of a multi-problem launch (the four HRNet branches): while the last stage of one
// branch's tile is computed, the first stage of the next branch's tile is already arriving.
// One s_barrier per stage hands a ring slot from the loader to the MFMA waves and the other one back.
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int OOB = 0x7fffffff;
constexpr int S3_MAXP = 4, S3_MAXST = 4;
constexpr int S3_TAB = 128;                       // table entries (k-steps of a problem)
constexpr int S3_SCR = S3_TAB * 16;               // statistics scratch behind the table
constexpr int S3_FIXED = 4096;
constexpr int S3_LDS = 81920;                     // two workgroups per CU
constexpr int S3_BUF = (S3_LDS - S3_FIXED) / 2;   // one ring slot: 38 KB
constexpr int S3_D = 3;                           // weight-fragment ring: k-steps in flight
constexpr int S3_THREADS = 320;

struct S3Prob {
    const bf16_t* x; const bf16_t* w; void* y; const float* bias; float* stats; const bf16_t* addend;
    int B, H, W, Cin, Cout;
    int x_bytes, y_bytes;
    int TH, NI, Wp, Sp, tiles_h, nnb, nks, nc16;
    int tile0, ntiles;
    unsigned char flip, relu, out_fp32, swz, has_idle, kw, nt, nst;
    unsigned char st_c0[S3_MAXST], st_nc[S3_MAXST];      // first 16-channel block / number of blocks of every stage
    unsigned short st_j0[S3_MAXST + 1];                  // table entries of stage s: [st_j0[s], st_j0[s + 1])
    unsigned short ninstr;                               // 1 KB copy instructions per stage
};
struct S3Launch { S3Prob p[S3_MAXP]; int n; int total; int* dbg; };     // dbg: optional [blocks][16] timestamps (tools/c3s_bench.py)

__device__ inline unsigned udiv24(unsigned n, unsigned d, float rcp) {      // n < 2^24
    unsigned q = (unsigned)((float)n * rcp);
    const int r = (int)(n - q * d);
    if (r < 0) --q; else if (r >= (int)d) ++q;
    return q;
}
__device__ inline void lds_barrier() {           // orders LDS traffic only: global stores keep draining across it
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

__device__ inline void tile_coords(const S3Prob& p, int tt, int& img0, int& y0, int& nb) {
    int pt;
    if (p.swz) {                               // tt = (pt_hi * nnb + nb) * 8 + pt_lo: the N-blocks of a pixel tile share an XCD's L2
        const int lo = tt & 7, rest = tt >> 3;
        const int hi = rest / p.nnb;
        nb = rest - hi * p.nnb; pt = hi * 8 + lo;
    } else {
        pt = tt / p.nnb; nb = tt - pt * p.nnb;
    }
    const int bi = pt / p.tiles_h, tb = pt - bi * p.tiles_h;
    img0 = bi * p.NI; y0 = tb * p.TH;
}
__device__ inline int first_tile(const S3Prob& p, int bid, int nblk) {      // ids congruent to bid modulo nblk over the launch's list
    int tau = bid - p.tile0 % nblk;
    if (tau < 0) tau += nblk;
    return tau;
}
// statistics leave the registers after a tile when the workgroup's next tile of the problem has another channel block
__device__ inline bool flush_after(const S3Prob& p, int tau, int nblk) {
    if (!p.stats) return false;
    if (tau + nblk >= p.ntiles) return true;
    int i0, y0, nb0, nb1;
    tile_coords(p, tau, i0, y0, nb0);
    tile_coords(p, tau + nblk, i0, y0, nb1);
    return nb0 != nb1;
}

// ---- loader wave ------------------------------------------------------------------------------------------------------
// Stage s of tile tau.  The LDS image is [slab][row][column][chunk] in 16-byte cells (Sp chunks per pixel, Wp = W + 2
// columns, TH + 2 rows per slab); every row is copied by its own sequence of 1 KB instructions (the last one partial:
// exec-masked), so lane l of a row's i-th instruction always holds the same (column, chunk) -- its source offset relative
// to the row is computed ONCE per stage (S3_NIR registers) and an instruction costs two scalar operations: a loader that
// decoded (row, column, chunk) per instruction took ~250 cycles per KB and was the bottleneck of the whole kernel.
// Cells outside the image (halo columns, rows above / below the image, padding chunks) carry an out-of-range offset:
// the hardware writes zeros.
constexpr int S3_NIR = 8;                         // copy instructions per row at most (Wp * Sp <= 512 cells)

// (one straight-line body per instruction count: with a run-time count every copy sat in its own basic block behind a
// taken branch, ~130 cycles per KB)
template <int NIR>
__device__ __forceinline__ void s3_rows(const S3Prob& p, int s, int img0, int y0, unsigned char* buf, int lane, int* dbg)
{
    const int Sp = p.Sp, Wp = p.Wp, W = p.W, H = p.H, TH2 = p.TH + 2, NI = p.NI;
    const int S_s = 2 * p.st_nc[s], ch0 = p.st_c0[s] * 32;
    const int pixb = p.Cin * 2, rowb = W * pixb;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const int rowcells = Wp * Sp;
    int voff[NIR];
    {
        const float rc = 1.0f / (float)Sp;
#pragma unroll
        for (int i = 0; i < NIR; ++i) {
            const int q = i * 64 + lane;
            const int c = (int)udiv24((unsigned)q, (unsigned)Sp, rc), ch = q - c * Sp;
            const bool valid = ch < S_s && (unsigned)(c - 1) < (unsigned)W;
            voff[i] = valid ? (c - 1) * pixb + ch0 + ch * 16 : OOB;
        }
    }
    const bool in_tail = lane < rowcells - (NIR - 1) * 64;          // active lanes of a row's last instruction
    if (dbg && lane == 0) dbg[13] = (int)clock64();
    for (int sl = 0; sl < NI; ++sl) {
        unsigned char* dst = buf + sl * TH2 * rowcells * 16;
        int yy = y0 - 1;
        int soff_row = ((img0 + sl) * H + yy) * rowb;
        for (int rr = 0; rr < TH2; ++rr, ++yy, soff_row += rowb, dst += rowcells * 16) {
            // a row outside the image: every lane out of range (offset + scalar offset beyond the tensor, no 32-bit overflow)
            const int soff = (unsigned)yy < (unsigned)H ? soff_row : 0x40000000;
#pragma unroll
            for (int i = 0; i < NIR - 1; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(dst + i * 1024), 16, voff[i], soff, 0, 0);
            if (in_tail) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(dst + (NIR - 1) * 1024), 16, voff[NIR - 1], soff, 0, 0);
        }
    }
}

__device__ inline void s3_issue(const S3Prob& p, int tau, int s, unsigned char* buf, int lane, int* dbg = nullptr)
{
    int img0, y0, nb;
    tile_coords(p, tau, img0, y0, nb);
    switch ((p.Wp * p.Sp + 63) >> 6) {
        case 1: s3_rows<1>(p, s, img0, y0, buf, lane, dbg); break;
        case 2: s3_rows<2>(p, s, img0, y0, buf, lane, dbg); break;
        case 3: s3_rows<3>(p, s, img0, y0, buf, lane, dbg); break;
        case 4: s3_rows<4>(p, s, img0, y0, buf, lane, dbg); break;
        case 5: s3_rows<5>(p, s, img0, y0, buf, lane, dbg); break;
        case 6: s3_rows<6>(p, s, img0, y0, buf, lane, dbg); break;
        case 7: s3_rows<7>(p, s, img0, y0, buf, lane, dbg); break;
        default: s3_rows<8>(p, s, img0, y0, buf, lane, dbg); break;
    }
}

struct Pos { int ii, tau, s; bool valid, first; };

__device__ inline Pos pos_first(const S3Launch& L, int bid, int nblk, int rot, int ii0) {
    Pos q{ii0, 0, 0, false, true};
    for (; q.ii < L.n; ++q.ii) {
        const S3Prob& p = L.p[(q.ii + rot) % L.n];
        q.tau = first_tile(p, bid, nblk);
        if (q.tau < p.ntiles) { q.valid = true; return q; }
    }
    return q;
}
__device__ inline Pos pos_next(const S3Launch& L, int bid, int nblk, int rot, const Pos& c) {
    const S3Prob& p = L.p[(c.ii + rot) % L.n];
    Pos q = c;
    q.first = false;
    if (c.s + 1 < p.nst) { q.s = c.s + 1; return q; }
    q.s = 0;
    if (c.tau + nblk < p.ntiles) { q.tau = c.tau + nblk; return q; }
    return pos_first(L, bid, nblk, rot, c.ii + 1);
}

__device__ __forceinline__ void s3_loader(const S3Launch& L, int bid, int nblk, int rot, unsigned char* smem, int lane)
{
    unsigned char* const ring = smem + S3_FIXED;
    Pos cur = pos_first(L, bid, nblk, rot, 0);
    if (!cur.valid) return;
    int* const dbg = L.dbg ? L.dbg + bid * 16 : nullptr;
    if (dbg && lane == 0) dbg[8] = (int)clock64();
    s3_issue(L.p[(cur.ii + rot) % L.n], cur.tau, cur.s, ring, lane, dbg);
    if (dbg && lane == 0) dbg[9] = (int)clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg && lane == 0) dbg[10] = (int)clock64();
    int g = 0;
    while (cur.valid) {
        const S3Prob& p = L.p[(cur.ii + rot) % L.n];
        const Pos nxt = pos_next(L, bid, nblk, rot, cur);
        if (cur.first) __builtin_amdgcn_s_barrier();                        // the MFMA waves' tap table
        if (nxt.valid) s3_issue(L.p[(nxt.ii + rot) % L.n], nxt.tau, nxt.s, ring + ((g + 1) & 1) * S3_BUF, lane);
        if (cur.s + 1 == p.nst) {                                           // the tile's last stage: K-split exchange, statistics
            if (p.kw > 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
            if (flush_after(p, cur.tau, nblk)) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the next stage has landed
        __builtin_amdgcn_s_barrier();                                       // ... and the current one has been consumed
        cur = nxt;
        ++g;
        if (dbg && lane == 0 && g == 1) dbg[11] = (int)clock64();
    }
    if (dbg && lane == 0) dbg[12] = (int)clock64();
}

// ---- MFMA waves -------------------------------------------------------------------------------------------------------
template <int NT, int KW>
__device__ __forceinline__ void s3_body(const S3Prob& p, const int bid, const int nblk, int& g, unsigned char* smem, int* dbg)
{
    constexpr int MT = 4, PW = 4 / KW, MO = MT / KW, D = S3_D;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave % PW, kw = wave / PW;
    i32x4* const sTab = reinterpret_cast<i32x4*>(smem);
    float* const sScr = reinterpret_cast<float*>(smem + S3_SCR);
    unsigned char* const ring = smem + S3_FIXED;
    const int Wp = p.Wp, Sp = p.Sp, W = p.W, H = p.H, TH = p.TH, NI = p.NI, nst = p.nst;

    // ---- once per problem: tap table, per-lane fragment addresses (the previous problem's readers passed a barrier) ------
    int sj[S3_MAXST + 1];                          // table entries of stage s: [sj[s], sj[s + 1]) (scalars: statically indexed reads)
#pragma unroll
    for (int k = 0; k <= S3_MAXST; ++k) sj[k] = p.st_j0[k < nst ? k : nst];
    // this wave's k-steps of stage s: an equal share of the stage's table entries
    auto range_of = [&](int s, int w, int& jb, int& je) {
        int a = sj[0], b = sj[1];
#pragma unroll
        for (int k = 1; k < S3_MAXST; ++k) if (s == k) { a = sj[k]; b = sj[k + 1]; }
        const int c = (b - a + KW - 1) / KW;
        jb = a + w * c; je = min(b, jb + c);
    };
    auto my_range = [&](int s, int& jb, int& je) { range_of(s, kw, jb, je); };
    const int nent = sj[S3_MAXST];
    if (t < nent) {
        int s = 0;
#pragma unroll
        for (int k = 1; k < S3_MAXST; ++k) if (k < nst && t >= sj[k]) s = k;
        const int jl = t - p.st_j0[s];
        const int nc16 = p.nc16;
        i32x4 e = {0, 0, 0, 0};
        auto tapoff = [&](int tap) {
            const int r = (tap * 11) >> 5, sx = tap - 3 * r;          // tap / 3 for tap < 9
            const int off = ((r - 1) * Wp + (sx - 1)) * Sp * 16;
            return p.flip ? -off : off;
        };
        if (nst == 1) {                               // (tap, 16-channel block) order; a k-step may straddle two taps when nc16 is odd
            const float rc_n = 1.0f / (float)nc16;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int h = 2 * jl + half;
                if (h > 9 * nc16 - 1) h = 9 * nc16 - 1;                 // zero-weight tail of the last k-step: any valid cell
                const int tap = (int)udiv24((unsigned)h, (unsigned)nc16, rc_n), c16 = h - tap * nc16;
                const int off = tapoff(tap) + c16 * 32;
                if (half == 0) e.x = off; else e.y = off;
            }
            e.z = jl * 1024;
        } else {                                      // an even number of blocks per stage: both halves of a k-step share the tap
            const int hn = p.st_nc[s] >> 1;
            const int tap = jl / hn, cl = 2 * (jl - tap * hn);
            e.x = tapoff(tap) + cl * 32; e.y = e.x + 32;
            e.z = ((tap * nc16 + p.st_c0[s] + cl) >> 1) * 1024;
        }
        // successor in the owning wave's k-step sequence (its share of this stage, then its share of the next one; the
        // tile's last k-step points at itself): the weight prefetch follows these links without a branch
        {
            int a = p.st_j0[s], b = p.st_j0[s + 1];
            const int c = (b - a + KW - 1) / KW;
            const int w = (t - a) / c;
            int jb_, je_;
            range_of(s, w, jb_, je_);
            int nx = t;
            if (t + 1 < je_) nx = t + 1;
            else if (s + 1 < nst) { int nb2, ne2; range_of(s + 1, w, nb2, ne2); nx = nb2; }
            e.w = nx;
        }
        sTab[t] = e;
    }
    const int thw = TH * W, npix = NI * thw;
    const int osz = p.out_fp32 ? 4 : 2;
    int lanebase[MT], outoff[MT];
    {
        const float rc_thw = 1.0f / (float)thw, rc_w = 1.0f / (float)W;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int j = (pw * MT + mt) * 16 + li;
            const bool valid = j < npix;
            const int jc = valid ? j : 0;
            const int sl = (int)udiv24((unsigned)jc, (unsigned)thw, rc_thw), rem = jc - sl * thw;
            const int r = (int)udiv24((unsigned)rem, (unsigned)W, rc_w), xx = rem - r * W;
            lanebase[mt] = ((sl * (TH + 2) + r + 1) * Wp + xx + 1) * Sp * 16 + (lg & 1) * 16;
            outoff[mt] = valid ? (((sl * H + r) * W + xx) * p.Cout + lg * 4) * osz : OOB;
        }
    }
    lds_barrier();                                 // the tap table is in place
    if (dbg && t == 0 && g == 0) dbg[1] = (int)clock64();
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int nks = p.nks;
    const int wlane = lane * 16;

    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }

    for (int tau = first_tile(p, bid, nblk); tau < p.ntiles; tau += nblk) {
        int img0, y0, nb;
        tile_coords(p, tau, img0, y0, nb);
        const int n0 = nb * (16 * NT);
        const bf16_t* wblk = p.w + (size_t)(n0 / 16) * (size_t)nks * 512;
        bf16x8 A[D][NT];
        // Weight-fragment loads are inline asm: the compiler's own vmcnt bookkeeping drains the whole ring at every loop
        // header (vmcnt(0) once per D k-steps, measured in the disassembly).  Invisible to it, they are counted by hand: the
        // ring is a FIFO, so whenever a slot is used exactly NT * (D - 1) younger fragment loads exist -- s_waitcnt
        // vmcnt(NT * (D - 1)) (ring_wait) is enough, and any compiler-issued memory operation in between only makes it more
        // conservative.  The varying part of the address travels in the vector offset (a VALU-written scalar offset would
        // need wait states the compiler does not insert inside asm).
        const unsigned long long wa = reinterpret_cast<unsigned long long>(wblk);
        const i32x4 wdesc = {__builtin_amdgcn_readfirstlane((int)(unsigned)wa), __builtin_amdgcn_readfirstlane((int)((wa >> 32) & 0xffffu)),
                             __builtin_amdgcn_readfirstlane(NT * nks * 1024), 0x00020000};      // raw buffer: stride 0, num_records in bytes
        int wso[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wso[nt] = __builtin_amdgcn_readfirstlane(nt * nks * 1024);
        auto load_a = [&](int voff, bf16x8* a) {                   // voff: lane * 16 + byte offset of the k-step's fragment of row block 0
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a[nt]) : "v"(voff), "s"(wdesc), "s"(wso[nt]) : "memory");
        };
        auto ring_wait = [&](bf16x8* a) {
            if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[0]) : "n"(1 * (D - 1)));
            if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(2 * (D - 1)));
            if constexpr (NT == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(3 * (D - 1)));
        };
        // The ring runs over this wave's k-steps of the whole tile (stage after stage): k-step i sits in slot i % D and, once
        // used, the slot is refilled with k-step i + D.  The prefetch walks the table's successor links D k-steps ahead of
        // the MFMAs; the entry of the NEXT refill is read one k-step early.
        int pj;
        { int je0; my_range(0, pj, je0); }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const i32x4 q = sTab[pj];
            load_a(wlane + q.z, A[d]);
            pj = __builtin_amdgcn_readfirstlane(q.w);
        }
        i32x4 qn = sTab[pj];                          // the next refill's table entry
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (dbg && t == 0 && g == 0) dbg[2] = (int)clock64();
        int r0 = 0;                                   // ring slot of the next k-step
        for (int s = 0; s < nst; ++s) {
            const unsigned char* const sX = ring + (g & 1) * S3_BUF;
            int j, je;
            my_range(s, j, je);
            i32x4 e = sTab[j];
            // one k-step on ring slot `a`: MT pixel fragments from the LDS tile, MT * NT MFMAs, the slot refilled
            auto kstep = [&](const int jj, bf16x8* a) {
                const int koff = lg >= 2 ? e.y : e.x;
                bf16x8 b[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + koff);
                e = sTab[min(jj + 1, je - 1)];
                const int voffn = wlane + qn.z;
                qn = sTab[__builtin_amdgcn_readfirstlane(qn.w)];
                ring_wait(a);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], b[mt], acc[mt][nt], 0, 0, 0);
                load_a(voffn, a);
                __builtin_amdgcn_sched_barrier(0);          // k-steps stay in order: every slot's loads are D k-steps ahead of their use
            };
            // head: up to D - 1 k-steps that bring the ring back to slot 0; groups of D without any condition inside (a
            // guarded step makes the compiler wait for ALL outstanding weight loads at the loop header); tail
#pragma unroll
            for (int d = 1; d < D; ++d)
                if (r0 != 0 && r0 <= d && j < je) { kstep(j, A[d]); ++j; }
            for (; j + D <= je; j += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) kstep(j + d, A[d]);
            }
            r0 = 0;
#pragma unroll
            for (int d = 0; d < D - 1; ++d)
                if (j < je) { kstep(j, A[d]); ++j; r0 = d + 1; }
            if (s + 1 < nst) { lds_barrier(); ++g; }          // slot consumed; the next stage's slot is complete
        }

        const bool stamp = dbg && t == 0 && g + 1 == nst;       // the workgroup's first tile
        if (stamp) dbg[3] = (int)clock64();
        // ---- K-split: partial sums meet in the consumed slot; wave kw finishes accumulator tiles [kw*MO, (kw+1)*MO) ------
        unsigned char* const sR = ring + (g & 1) * S3_BUF;
        if constexpr (KW > 1) {
            lds_barrier();                                          // every wave is done reading the tile
            unsigned char* const myred = sR + (size_t)wave * ((MT - MO) * NT * 1024) + lane * 16;
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                if (q != kw) {                                      // (uniform per wave)
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        const int f = (q < kw ? q : q - 1) * MO + m;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            *reinterpret_cast<f32x4*>(myred + (f * NT + nt) * 1024) = acc[q * MO + m][nt];
                    }
                }
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                if (q != kw) {                                      // contributions of wave (pw, q) to my tiles
                    const int src = q * PW + pw;
                    const int f0 = (kw < q ? kw : kw - 1) * MO;
                    const unsigned char* const rd = sR + (size_t)src * ((MT - MO) * NT * 1024) + lane * 16;
#pragma unroll
                    for (int m = 0; m < MO; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(rd + ((f0 + m) * NT + nt) * 1024);
#pragma unroll
                            for (int qq = 0; qq < KW; ++qq) if (qq == kw) acc[qq * MO + m][nt] += v;   // static index
                        }
                }
            }
        }

        if (stamp) dbg[4] = (int)clock64();
        // ---- epilogue on the wave's own tiles ------------------------------------------------------------------------------
        const int tile_out = ((img0 * H + y0) * W) * p.Cout * osz;
#pragma unroll
        for (int qq = 0; qq < KW; ++qq) {
            if (qq == kw) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cl = n0 + nt * 16 + lg * 4;
                    const bool cok = cl < p.Cout;
                    const int so = tile_out + (n0 + nt * 16) * osz;
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) {
                        const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.Cout * 4, 0x00020000);
                        bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(br, cok ? cl * 4 : OOB, 0, 0));
                    }
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        const int mt = qq * MO + m;
                        const int off = (cok && outoff[mt] != OOB) ? outoff[mt] : OOB;
                        f32x4 v = acc[mt][nt] + bv;
                        if (p.addend) {
                            const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.addend), 0, p.y_bytes, 0x00020000);
                            const i32x2 aq = __builtin_amdgcn_raw_buffer_load_b64(ar, off, so, 0);
                            v[0] += __uint_as_float((unsigned)aq.x << 16); v[1] += __uint_as_float((unsigned)aq.x & 0xffff0000u);
                            v[2] += __uint_as_float((unsigned)aq.y << 16); v[3] += __uint_as_float((unsigned)aq.y & 0xffff0000u);
                        }
                        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        if (p.out_fp32) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yr, off, so, 0);
                        } else {
                            const i32x2 pk = {(int)f2bf_pk(v[0], v[1]), (int)f2bf_pk(v[2], v[3])};
                            __builtin_amdgcn_raw_buffer_store_b64(pk, yr, off, so, 0);
                            if (p.stats) {           // BatchNorm statistics from the fp32 accumulators (see conv3x3.hip)
                                if (p.has_idle) {
                                    const float msk = off != OOB ? 1.f : 0.f;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { const float q = v[r] * msk; s1[nt][r] += q; s2[nt][r] = fmaf(q, q, s2[nt][r]); }
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { s1[nt][r] += v[r]; s2[nt][r] = fmaf(v[r], v[r], s2[nt][r]); }
                                }
                            }
                        }
                    }
                }
            }
        }
        if (stamp) dbg[5] = (int)clock64();
        // ---- statistics: 16 pixel lanes (DPP) -> 4 waves (scratch) -> one atomic per channel into replica bid % BN_NCOPY -------
        if (flush_after(p, tau, nblk)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row_sum16(s1[nt][r]), b = row_sum16(s2[nt][r]);
                    if (li == 0) { sScr[(wave * 2 + 0) * (NT * 16) + nt * 16 + lg * 4 + r] = a; sScr[(wave * 2 + 1) * (NT * 16) + nt * 16 + lg * 4 + r] = b; }
                    s1[nt][r] = 0.f; s2[nt][r] = 0.f;
                }
            lds_barrier();
            if (t < 2 * NT * 16) {
                const int which = t / (NT * 16), c = t - which * (NT * 16);
                const float v = (sScr[(0 * 2 + which) * (NT * 16) + c] + sScr[(1 * 2 + which) * (NT * 16) + c]) +
                                (sScr[(2 * 2 + which) * (NT * 16) + c] + sScr[(3 * 2 + which) * (NT * 16) + c]);
                if (n0 + c < p.Cout) atomicAdd(p.stats + ((size_t)(bid % BN_NCOPY) * 2 + which) * p.Cout + n0 + c, v);
            }
        }
        lds_barrier();                                              // the last stage's slot is consumed
        if (stamp) dbg[6] = (int)clock64();
        ++g;
    }
}

template <int NT>
__global__ __launch_bounds__(S3_THREADS, 3) void conv3x3_stream_kernel(S3Launch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s3_smem[];
    const int bid = blockIdx.x, nblk = gridDim.x;
    // every workgroup visits the problems in its own rotation (the two workgroups of a CU, ids 256 apart, are one problem apart)
    const int rot = (bid + bid / 256) % L.n;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 4) { s3_loader(L, bid, nblk, rot, s3_smem, threadIdx.x & 63); return; }
    int* const dbg = L.dbg ? L.dbg + bid * 16 : nullptr;
    if (dbg && threadIdx.x == 0) {
        dbg[0] = (int)clock64();
        dbg[15] = (int)((__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xff00u) | ((__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u) << 16));   // CU: HW_ID cu / sh / se, XCC_ID
    }
    int g = 0;
    for (int ii = 0; ii < L.n; ++ii) {
        const S3Prob& p = L.p[(ii + rot) % L.n];
        if (first_tile(p, bid, nblk) >= p.ntiles) continue;
        switch (p.kw) {
            case 1: s3_body<NT, 1>(p, bid, nblk, g, s3_smem, dbg); break;
            case 2: s3_body<NT, 2>(p, bid, nblk, g, s3_smem, dbg); break;
            default: s3_body<NT, 4>(p, bid, nblk, g, s3_smem, dbg); break;
        }
    }
    if (dbg && threadIdx.x == 0) dbg[7] = (int)clock64();
}

bool g_s3_on = getenv("DANET_NO_C3_STREAM") == nullptr;
int g_s3_blocks = getenv("DANET_C3S_BLOCKS") ? atoi(getenv("DANET_C3S_BLOCKS")) : 512;
int g_s3_kw = getenv("DANET_C3S_KW") ? atoi(getenv("DANET_C3S_KW")) : 0;          // forced K split (tests, A-B timing); 0: the planner's choice

// Tile plan of q for the K split KW; returns false when the tiling cannot run.
bool s3_plan_one(const ConvP& p, S3Prob& q, int NT, int KW) {
    constexpr int MT = 4;
    const int PW = 4 / KW, TP = PW * MT * 16;
    const int H = p.OH, W = p.OW;
    int NI = 1, TH = H;
    if (H * W <= TP) {
        for (int n = TP / (H * W); n >= 1; --n) if (p.B % n == 0) { NI = n; break; }
    } else {
        TH = 0;
        for (int h = TP / W; h >= 1; --h) if (H % h == 0) { TH = h; break; }
        if (TH == 0) return false;
    }
    if ((long)NI * TH * W * 4 < (long)TP * 3) return false;             // < 75 % of the register tile in use
    const int nc16 = p.Cin / 16, nrows = NI * (TH + 2), Wp = W + 2;
    const int nks = p.Kp / 32;
    if (nks > S3_TAB) return false;
    if (KW > 1 && 4 * (MT - MT / KW) * NT * 1024 > S3_BUF) return false;
    int nst = 0, per = 0, Sp = 0, ninstr = 0;
    for (int n = 1; n <= S3_MAXST; ++n) {
        int c = (nc16 + n - 1) / n;
        if (n > 1) { if (nc16 % 2) return false; c += c & 1; }
        int sp = 2 * c;
        while (sp % 4 != 2) ++sp;
        const long cells = (long)nrows * Wp * sp;
        const int ni = (int)((cells + 63) / 64);
        if (Wp * sp > S3_NIR * 64) continue;                             // (a row is at most S3_NIR copy instructions)
        if (cells * 16 <= S3_BUF) { nst = (nc16 + c - 1) / c; per = c; Sp = sp; ninstr = ni; break; }
    }
    if (nst == 0) return false;
    q.nst = (unsigned char)nst;
    int j0 = 0;
    for (int s = 0; s < nst; ++s) {
        const int c0 = s * per, nc = nc16 - c0 < per ? nc16 - c0 : per;
        q.st_c0[s] = (unsigned char)c0; q.st_nc[s] = (unsigned char)nc;
        q.st_j0[s] = (unsigned short)j0;
        const int ne = nst == 1 ? nks : 9 * nc / 2;
        if (ne - (KW - 1) * ((ne + KW - 1) / KW) < S3_D) return false;  // every wave's share of a stage fills the fragment ring
        j0 += ne;
    }
    q.st_j0[nst] = (unsigned short)j0;
    if (j0 != nks) return false;
    q.TH = TH; q.NI = NI; q.Wp = Wp; q.Sp = Sp; q.ninstr = (unsigned short)ninstr;
    q.tiles_h = H / TH; q.nnb = p.Cout_pad / (16 * NT); q.nks = nks; q.nc16 = nc16;
    const int npt = (p.B / NI) * q.tiles_h;
    q.ntiles = npt * q.nnb;
    q.kw = (unsigned char)KW; q.nt = (unsigned char)NT;
    q.swz = (npt % 8 == 0 && q.nnb > 1) ? 1 : 0;
    q.has_idle = (NI * TH * W != TP || p.Cout != p.Cout_pad) ? 1 : 0;
    return true;
}

bool s3_shape_ok(const ConvP& p) {
    if (!g_s3_on) return false;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.groups != 1) return false;
    if (p.H != p.OH || p.W != p.OW || p.Cin % 16 != 0 || p.Cout % 4 != 0) return false;
    if (p.x_bytes >= (1L << 30) || p.y_bytes >= (1L << 31)) return false;       // (rows outside the image add 2^30 to their offsets: see s3_issue)
    if ((long)p.Cout_pad * p.Kp * 2 >= (1L << 31)) return false;
    if ((long)p.B * p.H * p.W >= (1L << 24)) return false;
    if (p.bn_red) return false;                     // (the fused BatchNorm-backward reduction stays on conv3x3.hip)
    return danet_conv_nt(p.Cout) <= 3;
}

bool s3_plan(const ConvP& p, S3Prob& q, int nprob) {
    const int NT = danet_conv_nt(p.Cout);
    static const int cand[3] = {1, 2, 4};
    const int want = (2 * 256 + nprob - 1) / nprob;
    int best = -1;
    S3Prob tmp = q;
    for (int c = 0; c < 3; ++c) {
        S3Prob t = q;
        if (g_s3_kw && cand[c] != g_s3_kw) continue;
        if (!s3_plan_one(p, t, NT, cand[c])) continue;
        if (t.ntiles >= want) { q = t; return true; }
        if (t.ntiles > best) { best = t.ntiles; tmp = t; }
    }
    if (best < 0) return false;
    q = tmp;
    return true;
}

template <int NT>
void s3_launch_nt(const S3Launch& L, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_stream_kernel<NT>), dim3((unsigned)grid), dim3(S3_THREADS), (size_t)S3_LDS, st, L);
}

}  // namespace

namespace danet_conv {

// n (<= 4) problems in one launch of the streamed kernel.  0 on launch, -1 when the set cannot run here (nothing is
// launched then); dry = true only answers that question.
int conv3x3s_launch(const ConvP* ps, int n, void* stream, bool dry) {
    if (n < 1 || n > S3_MAXP) return -1;
    S3Launch L{};
    L.n = n;
    int tile0 = 0, NT = 0;
    for (int i = 0; i < n; ++i) {
        const ConvP& p = ps[i];
        if (!s3_shape_ok(p)) return -1;
        S3Prob& q = L.p[i];
        q.x = p.x; q.w = p.w; q.y = p.y; q.bias = p.bias; q.stats = p.stats; q.addend = p.addend;
        q.B = p.B; q.H = p.OH; q.W = p.OW; q.Cin = p.Cin; q.Cout = p.Cout;
        q.flip = p.transposed ? 1 : 0; q.relu = p.relu ? 1 : 0; q.out_fp32 = p.out_fp32 ? 1 : 0;
        q.x_bytes = (int)p.x_bytes; q.y_bytes = (int)p.y_bytes;
        if (!s3_plan(p, q, n)) return -1;
        if (NT == 0) NT = q.nt; else if (NT != q.nt) return -1;
        q.tile0 = tile0;
        tile0 += q.ntiles;
    }
    L.total = tile0;
    L.dbg = conv3x3_debug_buffer();
    if (dry) return 0;
    const int grid = L.total < g_s3_blocks ? L.total : g_s3_blocks;
    hipStream_t st = (hipStream_t)stream;
    switch (NT) {
        case 1: s3_launch_nt<1>(L, grid, st); return 0;
        case 2: s3_launch_nt<2>(L, grid, st); return 0;
        case 3: s3_launch_nt<3>(L, grid, st); return 0;
        default: return -1;
    }
}

}  // namespace danet_conv

// enable: 0 / 1 (-1 keeps); blocks: workgroup cap of a launch (<= 0 keeps); kw: forced K split 1 / 2 / 4 (0: the planner's
// choice, < 0 keeps).  Returns the previous `enable`.
extern "C" int danet_conv3x3_stream_set(int enable, int blocks, int kw) {
    const int prev = g_s3_on ? 1 : 0;
    if (enable >= 0) g_s3_on = enable != 0;
    if (blocks > 0) g_s3_blocks = blocks;
    if (kw >= 0) g_s3_kw = kw;
    return prev;
}
// What the streamed kernel would do with a problem: KW * 100 + stages * 10 + NT (0: not taken).
extern "C" int danet_conv3x3_stream_plan(int B, int H, int W, int Cin, int Cout, int nprob) {
    ConvP p{};
    p.B = B; p.H = p.OH = H; p.W = p.OW = W; p.Cin = Cin; p.Cout = Cout; p.R = p.S = 3; p.stride = p.pad = p.dil = p.groups = 1;
    p.K = 9 * Cin; p.Kp = (p.K + 31) / 32 * 32;
    const int nt = danet_conv_nt(Cout);
    p.Cout_pad = (Cout + 16 * nt - 1) / (16 * nt) * (16 * nt);
    p.x_bytes = (long)B * H * W * Cin * 2; p.y_bytes = (long)B * H * W * Cout * 2;
    if (!s3_shape_ok(p)) return 0;
    S3Prob q{};
    if (!s3_plan(p, q, nprob)) return 0;
    return q.kw * 100 + q.nst * 10 + q.nt;
}
