// 3x3 / stride-1 / pad-1 convolution (forward and data gradient), streamed: the successor of conv3x3.hip's tile kernel
// for the HRNet branch layers (/root/reference/models/module/hr_module.py:15-179, res_module.py:27-56).
//
// conv3x3.hip runs a tile as load -> MFMA -> store, serially, and all workgroups are in the same phase at once: memory
// and matrix cores take turns (16 % of the bf16 MFMA rate in round 2).  Here the input of the NEXT stage travels while the
// current one is computed:
//   * a workgroup (four waves, two workgroups per CU) owns a two-slot LDS ring.  Right after the barrier that ends a
//     stage every wave copies its share of the rows of the stage AFTER the one now starting into the slot that just became
//     free, with `buffer_load_dwordx4 ... lds` (LDS-DMA: no staging registers, no ds_write pass; lanes whose source lies
//     outside the image carry an out-of-range offset and the hardware writes zeros, so halo columns, rows outside the image
//     and padding chunks are zero-filled by the same instructions; tools/experiments/lds_dma*.hip: 16 cycles per 1 KB
//     copy to issue, the chip's full fabric rate with one tile in flight per workgroup).
//   * vmcnt returns in order, so a weight-fragment load issued after those copies completes after them (observed: tools/experiments/
//     dma_order.hip, 0 violations in 260 k wave-rounds, out-of-range copies included; S3_STRICT_COPIES builds the vmcnt(0) variant).  The fragment
//     ring is therefore deep (D k-steps = ~2 k cycles of MFMA work ahead) and refilled BEFORE the copies at a tile's start:
//     by the time a k-step needs a fragment requested after the copies, they have landed anyway.  (A first version used a
//     fifth, loader wave with its own vmcnt queue: 5-wave workgroups at 168 VGPRs do not co-reside two per CU -- measured
//     one -- and half the chip's MFMA issue slots sat empty.)
//   * fragment loads and copies are inline asm, counted by hand (s_waitcnt vmcnt(NT * (D - 1)) before a slot's use): the
//     compiler's own bookkeeping drained the whole ring at every loop header.
//   * what a workgroup needs once per problem is kept off its serial path (measured: ~7 k cycles of table / address set-up
//     per tile in the first version, as much as the k-loop): the tap table is built ONCE per problem signature (on the
//     host, uploaded into a caller-provided cache) and arrives by LDS-DMA together with the problem's first stage; per-lane pixel
//     addresses use shifts when the sizes are powers of two; the launch descriptor is read with scalar loads.
// A stage is a halo tile restricted to a range of input channels; a tile whose input does not fit one ring slot
// (96 / 192 / 384 channels) is a sequence of stages that share the accumulators -- the (tap, channel) k-steps of the
// UNCHANGED weight packing are visited stage by stage through the table.  The stage sequence of a workgroup runs across
// tiles and across the problems of a multi-problem launch (the four HRNet branches): while the last stage of one
// branch's tile is computed, the first stage of the next branch's tile is already arriving.
#include "common.h"
#include "conv_common.h"
#include "grid_barrier.h"
#include <type_traits>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int OOB = 0x7fffffff;
constexpr int S3_MAXP = 4, S3_MAXST = 4;
constexpr int S3_TAB = 112;                       // table entries (k-steps of a problem)
constexpr int S3_TABB = S3_TAB * 16;              // bytes of one table slot (two slots: the next problem's table arrives early)
constexpr int S3_SCR = 2 * S3_TABB;               // statistics scratch behind the tables (4 waves x 2 x 48 floats)
constexpr int S3_FIXED = S3_SCR + 1536;
constexpr int S3_LDS = 81920;                     // two workgroups per CU
constexpr int S3_BUF = (S3_LDS - S3_FIXED) / 2;   // one ring slot: 37.5 KB
constexpr int S3_D = 4;                           // weight-fragment ring: k-steps in flight (even: ring slot d uses pixel-fragment buffer d & 1)
#ifndef DANET_S3_PLAIN_EPI
#define DANET_S3_PLAIN_EPI 1
#endif
constexpr bool S3_PLAIN_EPI = DANET_S3_PLAIN_EPI != 0;      // specialised epilogue for the option-free case (build knob)
#ifndef DANET_S3_WRAP
#define DANET_S3_WRAP 0
#endif
constexpr bool S3_WRAP = DANET_S3_WRAP != 0;     // the ring carries over from a tile to the next (build knob; measured neutral: 40.1 vs 40.3 us on the four-branch launch, so off)
#ifndef S3_STEP_OVER
#define S3_STEP_OVER 0         // build knob: ring waits step over the stage copies queued in front of them (ring_wait).  Correct, but the four-
#endif                         // branch launch takes 40.9 us with it against 39.4: the CU's other workgroup already covers that stall.  Off.
#ifndef S3_STRICT_COPIES
#define S3_STRICT_COPIES 0     // build knob: a stage's slot is published after vmcnt(0) instead of on the ring waits' in-order argument
#endif                         // (38.2-40.4 us against 38.6-39.4: no need -- tools/experiments/dma_order.hip finds loads in issue order).  Off.
#ifndef DANET_S3_WIDE_STORES
#define DANET_S3_WIDE_STORES 0
#endif
constexpr bool S3_WIDE_STORES = DANET_S3_WIDE_STORES != 0;  // 16-byte epilogue stores through v_permlane16_swap (build knob; measured neutral on the four-branch launch: 39.5 / 39.6 / 40.4 / 38.2 us with, 40.9 / 37.2 / 39.2 / 37.4 without -- the epilogue is not store-issue bound --, so off)
constexpr int S3_THREADS = 256;

struct S3Prob {
    const bf16_t* x; const bf16_t* w; void* y; const float* bias; float* stats; const bf16_t* addend;
    const i32x4* tab;                                    // the problem's tap table (device cache, see s3_table_for)
    int B, H, W, Cin, Cout;                              // Cin, Cout: channels PER GROUP
    int groups, nnbg, pixb_in, Cout_tot;                 // grouped layers: N-blocks per group, bytes per input pixel, output channels of all groups
    float rc_nnbg;
    int x_bytes, y_bytes;
    int TH, NI, Wp, Sp, tiles_h, nnb, nks, nc16;
    int tile0, ntiles;
    // The problem's tiles are dealt round-robin to the nwg workgroups wg0, wg0 + 1, ... (modulo the grid): workgroup wg0 + d takes tiles
    // d, d + nwg, ...  (filled at launch: s3_assign).  Rounds 2-4 dealt the launch's concatenated tile list over ALL workgroups, which gave
    // a workgroup of the four-branch launch 3 or 5 units of work (a 48-channel tile = 1) where 4 is the average, 5 / 3 / 1 in the
    // three-branch launches, 3 / 1 in the two-branch ones; now every problem gets a share of the workgroups in proportion to its work.
    int wg0, nwg;
    float rc_nnb, rc_th;                                 // 1 / nnb, 1 / tiles_h (division-free tile coordinates)
    int lw, lthw;                                        // log2(W), log2(TH * W) when both are powers of two, else -1
    int flip, relu, out_fp32, swz, has_idle, kw, nt, nst, nent;
    // per stage s, one byte each (words, not arrays: the device reads them with run-time s): first 16-channel block, number
    // of blocks, END of the stage's table entries (stage s owns entries [j0(s - 1), j0(s)), j0(-1) = 0)
    unsigned st_c0w, st_ncw, st_j0w;
};
__host__ __device__ inline int st_c0(const S3Prob& p, int s) { return (int)((p.st_c0w >> (8 * s)) & 255u); }
__host__ __device__ inline int st_nc(const S3Prob& p, int s) { return (int)((p.st_ncw >> (8 * s)) & 255u); }
__host__ __device__ inline int st_end(const S3Prob& p, int s) { return (int)((p.st_j0w >> (8 * s)) & 255u); }
__host__ __device__ inline int st_begin(const S3Prob& p, int s) { return s == 0 ? 0 : st_end(p, s - 1); }
static_assert(S3_TAB < 256 && sizeof(S3Prob) % 4 == 0, "S3Prob");
// The debug stamps are GLOBAL-address-space stores: through a generic pointer they are FLAT instructions, and a pending FLAT access
// makes the compiler's waitcnt pass force lgkmcnt(0) at every LDS dependency for the rest of the kernel (it never sees a vmcnt
// wait here -- those are hand-counted inline asm -- so the FLAT access stays "pending" for it): the pixel-fragment prefetch of the
// k-loop was serialised by exactly that.
typedef __attribute__((address_space(1))) int* dbg_ptr;
struct S3Launch { S3Prob p[S3_MAXP]; int n; int total; int* dbg; };     // dbg: optional [blocks][16] timestamps (tools/c3s_diag.py)
// The BatchNorm that follows problem i, applied by the same launch (s3_bn_tail; conv_common.h BnApply): all problems of the launch or none.
struct S3Bn {
    const bf16_t* res; bf16_t* out; const float* gamma; const float* beta; float* running_mean; float* running_var; float* saved; unsigned char* mask;
    float inv_count, unbias; int relu, pad_;
};
static_assert(sizeof(S3Bn) % 4 == 0, "S3Bn");
struct S3LaunchBn { S3Launch c; S3Bn bn[S3_MAXP]; unsigned* bar; float momentum, eps; };

// The kernel argument is indexed with run-time problem numbers.  Done on the by-value argument the compiler copies the
// whole structure to scratch memory (600 bytes per lane, every access a counted vector load that drains the fragment
// ring); read through the kernel-argument segment as constant memory the accesses are scalar loads.
typedef const __attribute__((address_space(4))) S3Launch* S3LaunchK;
__device__ inline S3LaunchK s3_args() { return (S3LaunchK)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ inline S3Prob desc_prob(int idx) {
    union { S3Prob v; int w[sizeof(S3Prob) / 4]; } u;
    const __attribute__((address_space(4))) int* const src = (const __attribute__((address_space(4))) int*)&s3_args()->p[idx];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(S3Prob) / 4); ++i) u.w[i] = src[i];
    return u.v;
}
#define S3_FIELD(idx, field) (s3_args()->p[idx].field)
// (the BatchNorm launch: the same kernel-argument segment, longer -- S3Launch is its first member)
typedef const __attribute__((address_space(4))) S3LaunchBn* S3LaunchBnK;
__device__ inline S3LaunchBnK s3_bn_args() { return (S3LaunchBnK)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ inline S3Bn desc_bn(int idx) {
    union { S3Bn v; int w[sizeof(S3Bn) / 4]; } u;
    const __attribute__((address_space(4))) int* const src = (const __attribute__((address_space(4))) int*)&s3_bn_args()->bn[idx];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(S3Bn) / 4); ++i) u.w[i] = src[i];
    return u.v;
}

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

// ---- the tap table of a problem (built once per signature, s3_table_for) ---------------------------------------------------
// Entry j = one k-step (32 K values = two 16-channel blocks): {x, y} = LDS byte offsets of its two halves relative to a
// lane's pixel cell, z = byte offset of its weight fragment (row block 0) in the packed operand, w = the next k-step of
// the wave that owns entry j (its share of this stage, then of the following stages; the last one points back at the wave's
// first k-step: the ring's refills past the end of a tile fetch the first fragments of the next tile, see s3_problem).
struct TabKey { int Wp, Sp, nc16, flip, kw, nst; unsigned c0w, ncw, j0w; };
struct TabEntry { int x, y, z, w; };
static_assert(sizeof(TabEntry) == sizeof(i32x4), "TabEntry");

// Host side: the table is a pure function of the key (<= 112 entries), computed here and uploaded once per key.
void s3_table_host(const TabKey& k, TabEntry* out /* [S3_TAB] */)
{
    const int nst = k.nst, KW = k.kw, nc16 = k.nc16;
    auto sbeg = [&](int s) { return s == 0 ? 0 : (int)((k.j0w >> (8 * (s - 1))) & 255u); };
    auto send = [&](int s) { return (int)((k.j0w >> (8 * s)) & 255u); };
    auto range_of = [&](int s, int w, int& jb, int& je) {
        const int a = sbeg(s), b = send(s);
        const int c = (b - a + KW - 1) / KW;
        jb = std::min(b, a + w * c); je = std::min(b, jb + c);
    };
    auto tapoff = [&](int tap) {
        const int r = tap / 3, sx = tap - 3 * r;
        const int off = ((r - 1) * k.Wp + (sx - 1)) * k.Sp * 16;
        return k.flip ? -off : off;
    };
    const int nent = send(nst - 1);
    for (int t = 0; t < S3_TAB; ++t) {
        TabEntry e{0, 0, 0, 0};
        if (t < nent) {
            int s = 0;
            while (s + 1 < nst && t >= send(s)) ++s;
            const int jl = t - sbeg(s);
            if (nst == 1) {                           // (tap, 16-channel block) order; a k-step may straddle two taps when nc16 is odd
                for (int half = 0; half < 2; ++half) {
                    int h = 2 * jl + half;
                    if (h > 9 * nc16 - 1) h = 9 * nc16 - 1;             // zero-weight tail of the last k-step: any valid cell
                    const int tap = h / nc16, c16 = h - tap * nc16;
                    const int off = tapoff(tap) + c16 * 32;
                    if (half == 0) e.x = off; else e.y = off;
                }
                e.z = jl * 1024;
            } else {                                  // an even number of blocks per stage: both halves of a k-step share the tap
                const int hn = (int)((k.ncw >> (8 * s)) & 255u) >> 1;
                const int tap = jl / hn, cl = 2 * (jl - tap * hn);
                e.x = tapoff(tap) + cl * 32; e.y = e.x + 32;
                e.z = ((tap * nc16 + (int)((k.c0w >> (8 * s)) & 255u) + cl) >> 1) * 1024;
            }
            const int a = sbeg(s), b = send(s);
            const int c = (b - a + KW - 1) / KW;
            const int w = (t - a) / c;
            int jb, je;
            range_of(s, w, jb, je);
            int nx = -1;
            if (t + 1 < je) nx = t + 1;
            else for (int q = nst - 1; q > s; --q) { int nb2, ne2; range_of(q, w, nb2, ne2); if (nb2 < ne2) nx = nb2; }
            if (nx < 0)                                   // the wave's last k-step of the tile: on to its first one (of the NEXT tile)
                for (int q = nst - 1; q >= 0; --q) { int nb2, ne2; range_of(q, w, nb2, ne2); if (nb2 < ne2) nx = nb2; }
            e.w = nx < 0 ? t : nx;
        }
        out[t] = e;
    }
}

// Cache of tap tables in CALLER-PROVIDED device memory (danet_conv3x3_stream_tables: one workspace per device, registered by
// the host side before the first launch; without one the streamed kernel refuses and conv3x3.hip's tile kernel runs).  A
// table is a pure function of its key: computed on the host and uploaded with a synchronous copy the first time a key is
// seen, so it is complete before ANY stream can launch a kernel that reads it (a builder kernel on the launching stream
// would be unordered against other streams and only recorded, not run, under a capture).  New keys are refused while the
// launching stream is capturing (a synchronous copy is illegal there): the caller falls back for that launch.
struct TabPool { i32x4* base = nullptr; int cap = 0, used = 0; std::map<std::tuple<int, int, int, int, int, int, unsigned, unsigned, unsigned>, int> index; };
std::mutex g_tab_mutex;
std::map<int, TabPool> g_tab_pools;

const i32x4* s3_table_for(const S3Prob& q, hipStream_t st, bool dry)
{
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto pit = g_tab_pools.find(dev);
    if (pit == g_tab_pools.end() || !pit->second.base) return nullptr;       // no workspace registered for this device
    TabPool& pool = pit->second;
    const auto key = std::make_tuple(q.Wp, q.Sp, q.nc16, q.flip, q.kw, q.nst, q.st_c0w, q.st_ncw, q.st_j0w);
    auto it = pool.index.find(key);
    if (it != pool.index.end()) return pool.base + (size_t)it->second * S3_TAB;
    if (dry) return reinterpret_cast<const i32x4*>(1);                       // (a dry run only asks whether the problem qualifies)
    if (pool.used >= pool.cap) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    TabKey k{q.Wp, q.Sp, q.nc16, q.flip, q.kw, q.nst, q.st_c0w, q.st_ncw, q.st_j0w};
    TabEntry host[S3_TAB];
    s3_table_host(k, host);
    const int slot = pool.used;
    if (hipMemcpy(pool.base + (size_t)slot * S3_TAB, host, S3_TABB, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ++pool.used;
    pool.index[key] = slot;
    return pool.base + (size_t)slot * S3_TAB;
}

// ---- tile bookkeeping --------------------------------------------------------------------------------------------------
__device__ inline void tile_coords(const S3Prob& p, int tt, int& img0, int& y0, int& nb) {
    int pt;
    if (p.swz) {                               // tt = (pt_hi * nnb + nb) * 8 + pt_lo: the N-blocks of a pixel tile share an XCD's L2
        const int lo = tt & 7, rest = tt >> 3;
        const int hi = (int)udiv24((unsigned)rest, (unsigned)p.nnb, p.rc_nnb);
        nb = rest - hi * p.nnb; pt = hi * 8 + lo;
    } else {
        pt = (int)udiv24((unsigned)tt, (unsigned)p.nnb, p.rc_nnb); nb = tt - pt * p.nnb;
    }
    const int bi = (int)udiv24((unsigned)pt, (unsigned)p.tiles_h, p.rc_th), tb = pt - bi * p.tiles_h;
    img0 = __builtin_amdgcn_readfirstlane(bi * p.NI); y0 = __builtin_amdgcn_readfirstlane(tb * p.TH);
    nb = __builtin_amdgcn_readfirstlane(nb);
}
__device__ inline int first_tile_of(int wg0, int nwg, int bid, int nblk) {      // the workgroup's first tile of a problem (>= ntiles: none)
    int d = bid - wg0;
    if (d < 0) d += nblk;
    return d < nwg ? d : 0x3fffffff;
}
// statistics leave the registers after a tile when the workgroup's next tile of the problem has another channel block
__device__ inline bool flush_after(const S3Prob& p, int tau, int nblk) {
    if (!p.stats) return false;
    (void)nblk;
    if (tau + p.nwg >= p.ntiles) return true;
    if (p.nnb == 1) return false;
    int i0, y0, nb0, nb1;
    tile_coords(p, tau, i0, y0, nb0);
    tile_coords(p, tau + p.nwg, i0, y0, nb1);
    return nb0 != nb1;
}

// ---- stage copies ------------------------------------------------------------------------------------------------------
// Stage s of tile tau.  The LDS image is [slab][row][column][chunk] in 16-byte cells (Sp chunks per pixel, Wp = W + 2
// columns, TH + 2 rows per slab); every row is copied by its own sequence of 1 KB instructions (the last one partial:
// exec-masked), so lane l of a row's i-th instruction always holds the same (column, chunk): its source offset relative
// to the row is computed once per stage (NIR registers, one division and then increments) and an instruction costs three
// scalar operations.  (A first version decoded (row, column, chunk) per instruction: ~250 cycles per KB; with a run-time
// instruction count every copy sat in its own basic block behind a taken branch: ~130.)  Wave w copies rows w, w + 4, ...
// Cells outside the image (halo columns, rows above / below the image, padding chunks) carry an out-of-range offset: the
// hardware writes zeros.  With two workgroups of four waves copying at once an instruction takes ~100 cycles: the CU's
// 64 B / clk vector-memory path, not the issue, is the limit (84 KB per CU and tile pair = 1.3 k cycles).
constexpr int S3_NIR = 8;                         // copy instructions per row at most (Wp * Sp <= 512 cells)

__device__ __forceinline__ void dma16(unsigned lds_addr, int voff, const i32x4& desc, int soff) {
    // M0 = LDS base of the 1 KB piece (written in the statement that uses it: the compiler does not preserve it)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ inline i32x4 raw_desc(const void* base, int bytes) {           // raw buffer: stride 0, num_records in bytes
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000};
}

template <int NIR>
__device__ __forceinline__ int s3_rows(const S3Prob& p, int s, int img0, int y0, int gofs, unsigned char* buf, int wave, int lane)
{
    const int Sp = p.Sp, Wp = p.Wp, W = p.W, H = p.H, TH2 = p.TH + 2, NI = p.NI;
    const int S_s = 2 * st_nc(p, s), ch0 = st_c0(p, s) * 32 + gofs;        // gofs: byte offset of the tile's group within a pixel
    const int pixb = p.pixb_in, rowb = W * pixb;
    const i32x4 desc = raw_desc(p.x, p.x_bytes);
    const int rowcells = Wp * Sp;
    int voff[NIR];
    {
        // cell q = i * 64 + lane of a row = (column c, chunk ch): one division, then steps of 64 cells
        int c = (int)udiv24((unsigned)lane, (unsigned)Sp, 1.0f / (float)Sp), ch = lane - c * Sp;
        const int dc = 64 / Sp, dch = 64 - dc * Sp;
#pragma unroll
        for (int i = 0; i < NIR; ++i) {
            const bool valid = ch < S_s && (unsigned)(c - 1) < (unsigned)W;
            voff[i] = valid ? (c - 1) * pixb + ch * 16 : OOB;          // (the stage's channel offset travels in the scalar offset)
            ch += dch; c += dc;
            if (ch >= Sp) { ch -= Sp; ++c; }
        }
    }
    const bool in_tail = lane < rowcells - (NIR - 1) * 64;          // active lanes of a row's last instruction
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr_t)buf;
    const int nrows = NI * TH2;
    int sl = 0, rr = wave;
    for (int ra = wave; ra < nrows; ra += 4, rr += 4) {
        while (rr >= TH2) { rr -= TH2; ++sl; }
        const int yy = y0 - 1 + rr;
        // a row outside the image: every lane out of range (offset + scalar offset beyond the tensor, no 32-bit overflow)
        const int soff = __builtin_amdgcn_readfirstlane((unsigned)yy < (unsigned)H ? ((img0 + sl) * H + yy) * rowb + ch0 : 0x40000000);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(ra * rowcells * 16));
#pragma unroll
        for (int i = 0; i < NIR - 1; ++i) dma16(dst + i * 1024, voff[i], desc, soff);
        if (in_tail) dma16(dst + (NIR - 1) * 1024, voff[NIR - 1], desc, soff);
    }
    return nrows > wave ? ((nrows - wave + 3) >> 2) * NIR : 0;        // copy instructions this wave has just queued (wave-uniform)
}

// with_tab: the problem's tap table travels with this stage (wave 0 copies it to LDS address tab_dst: S3_TAB entries = 1 KB + 768 B)
__device__ inline int s3_issue(const S3Prob& p, int tau, int s, unsigned char* buf, bool with_tab, unsigned tab_dst, int wave, int lane)
{
    int extra = 0;
    if (with_tab && wave == 0) {
        const i32x4 tdesc = raw_desc(p.tab, S3_TABB);
        dma16(tab_dst, lane * 16, tdesc, 0);
        if (lane < (S3_TABB - 1024) / 16) dma16(tab_dst + 1024, lane * 16, tdesc, 1024);
        extra = 2;
    }
    int img0, y0, nb;
    tile_coords(p, tau, img0, y0, nb);
    const int gofs = p.groups > 1 ? __builtin_amdgcn_readfirstlane((int)udiv24((unsigned)nb, (unsigned)p.nnbg, p.rc_nnbg) * p.Cin * 2) : 0;
    switch ((p.Wp * p.Sp + 63) >> 6) {
        case 1: return extra + s3_rows<1>(p, s, img0, y0, gofs, buf, wave, lane);
        case 2: return extra + s3_rows<2>(p, s, img0, y0, gofs, buf, wave, lane);
        case 3: return extra + s3_rows<3>(p, s, img0, y0, gofs, buf, wave, lane);
        case 4: return extra + s3_rows<4>(p, s, img0, y0, gofs, buf, wave, lane);
        case 5: return extra + s3_rows<5>(p, s, img0, y0, gofs, buf, wave, lane);
        case 6: return extra + s3_rows<6>(p, s, img0, y0, gofs, buf, wave, lane);
        case 7: return extra + s3_rows<7>(p, s, img0, y0, gofs, buf, wave, lane);
        default: return extra + s3_rows<8>(p, s, img0, y0, gofs, buf, wave, lane);
    }
}

// The workgroup's stage sequence: problems in its rotation, its tiles of a problem, the stages of a tile.
// newprob: the stage is the first one of a problem visit (its table travels with it).
struct Pos { int ii, tau, s; bool valid, newprob; };

__device__ inline int wrap_idx(int i, int n) { return i >= n ? i - n : i; }          // (i < 2 n)
__device__ inline Pos pos_first(int nprob, int bid, int nblk, int rot, int ii0) {
    Pos q{ii0, 0, 0, false, true};
    for (; q.ii < nprob; ++q.ii) {
        const int idx = wrap_idx(q.ii + rot, nprob);
        q.tau = first_tile_of(S3_FIELD(idx, wg0), S3_FIELD(idx, nwg), bid, nblk);
        if (q.tau < S3_FIELD(idx, ntiles)) { q.valid = true; return q; }
    }
    return q;
}
__device__ inline Pos pos_next(int nprob, int bid, int nblk, int rot, const Pos& c) {
    const int idx = wrap_idx(c.ii + rot, nprob);
    Pos q = c;
    q.newprob = false;
    if (c.s + 1 < S3_FIELD(idx, nst)) { q.s = c.s + 1; return q; }
    q.s = 0;
    if (c.tau + S3_FIELD(idx, nwg) < S3_FIELD(idx, ntiles)) { q.tau = c.tau + S3_FIELD(idx, nwg); return q; }
    return pos_first(nprob, bid, nblk, rot, c.ii + 1);
}
// np: problem visits started so far (the table slot of the problem being computed is (np - 1) & 1, a new one's np & 1)
// returns the number of copy instructions this wave queued (0: no further stage)
__device__ inline int issue_pos(int nprob, int rot, const Pos& q, unsigned char* smem, unsigned char* slot, int np, int wave, int lane) {
    if (q.valid) {
        const S3Prob p = desc_prob(wrap_idx(q.ii + rot, nprob));
        const unsigned tab_dst = (unsigned)(unsigned long long)(lds_ptr_t)(smem + (np & 1) * S3_TABB);
        return s3_issue(p, q.tau, q.s, slot, q.newprob, tab_dst, wave, lane);
    }
    return 0;
}

// ---- a tile's end: K-split exchange, epilogue, statistics ---------------------------------------------------------------
// PLAIN: no bias, addend, ReLU, fp32 output or idle lanes -- the common case (every BasicBlock convolution), instantiated without the
// per-value tests of those options (12 accumulator tiles x 5 uniform branches per tile otherwise).
template <int NT, int KW, bool PLAIN>
__device__ __forceinline__ void s3_finish(const S3Prob& p, f32x4 (*acc)[NT], float (*s1)[4], float (*s2)[4], const int* outoff, unsigned char* sR,
                                          float* sScr, int img0, int y0, int n0, int climit, bool flush, int bid, dbg_ptr dbg_stamp)
{
    constexpr int MT = 4, PW = 4 / KW, MO = MT / KW;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave % PW, kw = wave / PW;
    const int H = p.H, W = p.W;
    const bool has_bias = PLAIN ? false : p.bias != nullptr, has_add = PLAIN ? false : p.addend != nullptr;
    const bool relu = PLAIN ? false : (bool)p.relu, out_fp32 = PLAIN ? false : (bool)p.out_fp32, has_idle = PLAIN ? false : (bool)p.has_idle;
    const int osz = out_fp32 ? 4 : 2;
    // ---- K-split: partial sums meet in the consumed slot; wave kw finishes accumulator tiles [kw*MO, (kw+1)*MO) ----------
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
    if (dbg_stamp) dbg_stamp[4] = (int)clock64();
    // ---- epilogue on the wave's own tiles ----------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    // n0: first output channel of the tile's block among ALL groups' channels, climit: end of its group's channels
    const int tile_out = ((img0 * H + y0) * W) * p.Cout_tot * osz;
    auto add_stats = [&](int nt, f32x4 v) {                 // BatchNorm statistics from the fp32 accumulators (see conv3x3.hip), two values per instruction
        f32x2_ lo = {v[0], v[1]}, hi = {v[2], v[3]};
        f32x2_& a0 = *reinterpret_cast<f32x2_*>(&s1[nt][0]); f32x2_& a1 = *reinterpret_cast<f32x2_*>(&s1[nt][2]);
        f32x2_& q0 = *reinterpret_cast<f32x2_*>(&s2[nt][0]); f32x2_& q1 = *reinterpret_cast<f32x2_*>(&s2[nt][2]);
        a0 += lo; a1 += hi;
        q0 = __builtin_elementwise_fma(lo, lo, q0); q1 = __builtin_elementwise_fma(hi, hi, q1);
    };
    // PLAIN, two channel blocks at a time: v_permlane16_swap trades lane rows 1 <-> 0 and 3 <-> 2 between the packed registers of
    // blocks a and b, after which a lane owns EIGHT consecutive channels (rows 0 / 2: block a's channels 0-7 / 8-15, rows 1 / 3:
    // block b's) and stores 16 bytes -- 8 instead of 12 store instructions per pixel fragment row at NT = 3 (csrc/conv_pw.hip measured
    // the same exchange: 8-byte pieces 32 bytes apart cost the write path twice the instructions for the same lines).
    constexpr int NPAIR = (PLAIN && S3_WIDE_STORES) ? NT / 2 : 0;
#pragma unroll
    for (int qq = 0; qq < KW; ++qq) {
        if (qq == kw) {
            if constexpr (NPAIR > 0) {
#pragma unroll
                for (int np = 0; np < NPAIR; ++np)
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        const int mt = qq * MO + m;
                        const f32x4 va = acc[mt][2 * np], vb = acc[mt][2 * np + 1];
                        if (p.stats) { add_stats(2 * np, va); add_stats(2 * np + 1, vb); }
                        const auto sx = __builtin_amdgcn_permlane16_swap(f2bf_pk(va[0], va[1]), f2bf_pk(vb[0], vb[1]), false, false);
                        const auto sy = __builtin_amdgcn_permlane16_swap(f2bf_pk(va[2], va[3]), f2bf_pk(vb[2], vb[3]), false, false);
                        const i32x4 q = {(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
                        // outoff holds the lane's 4-channel run (lg * 8 bytes); the 8-channel run starts at block (lg & 1), channel (lg >> 1) * 8
                        const int off = outoff[mt] != OOB ? outoff[mt] - lg * 8 + (lg & 1) * 32 + (lg >> 1) * 16 : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(q, yr, off, tile_out + (n0 + 2 * np * 16) * 2, 0);
                    }
            }
#pragma unroll
            for (int nt = 2 * NPAIR; nt < NT; ++nt) {
                const int cl = n0 + nt * 16 + lg * 4;
                const bool cok = cl < climit;
                const int so = tile_out + (n0 + nt * 16) * osz;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (has_bias) {
                    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.Cout_tot * 4, 0x00020000);
                    bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(br, cok ? cl * 4 : OOB, 0, 0));
                }
#pragma unroll
                for (int m = 0; m < MO; ++m) {
                    const int mt = qq * MO + m;
                    const int off = (cok && outoff[mt] != OOB) ? outoff[mt] : OOB;
                    f32x4 v = acc[mt][nt];
                    if (has_bias) v += bv;
                    if (has_add) {
                        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.addend), 0, p.y_bytes, 0x00020000);
                        const i32x2 aq = __builtin_amdgcn_raw_buffer_load_b64(ar, off, so, 0);
                        v[0] += __uint_as_float((unsigned)aq.x << 16); v[1] += __uint_as_float((unsigned)aq.x & 0xffff0000u);
                        v[2] += __uint_as_float((unsigned)aq.y << 16); v[3] += __uint_as_float((unsigned)aq.y & 0xffff0000u);
                    }
                    if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (out_fp32) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yr, off, so, 0);
                    } else {
                        const i32x2 pk = {(int)f2bf_pk(v[0], v[1]), (int)f2bf_pk(v[2], v[3])};
                        __builtin_amdgcn_raw_buffer_store_b64(pk, yr, off, so, 0);
                        if (p.stats) {
                            if (has_idle) { const float msk = off != OOB ? 1.f : 0.f; v *= msk; }
                            add_stats(nt, v);
                        }
                    }
                }
            }
        }
    }
    if (dbg_stamp) dbg_stamp[5] = (int)clock64();
    // ---- statistics: 16 pixel lanes (DPP) -> 4 waves (scratch) -> one atomic per channel into replica bid % BN_NCOPY -----------
    if (flush) {
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
            // (a GLOBAL atomic: see dbg_ptr -- a FLAT one would serialise every later LDS wait of the kernel)
            if (n0 + c < climit)
                bn_acc_add(p.stats, bid, which, p.Cout_tot, n0 + c, v);
        }
    }
}

// ---- one problem of the launch: every tile of this workgroup -----------------------------------------------------------
// g: stages completed so far (ring slot of the current stage = g & 1); np: problem visits started (this one included).
// On entry the problem's first stage and its table are complete in their slots and published by a barrier; nothing
// later has been requested.
template <int NT>
__device__ __forceinline__ void s3_problem(const int nprob, const int ii, const int rot, const int bid, const int nblk, int& g, const int np,
                                           unsigned char* smem, dbg_ptr dbg, int (&dsum)[3])
{
    constexpr int MT = 4, D = S3_D;
    const S3Prob p = desc_prob(wrap_idx(ii + rot, nprob));
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int KW = p.kw, lkw = KW >> 1;                        // KW = 1, 2, 4 -> log2 = 0, 1, 2
    const int PW = 4 >> lkw;
    const int pw = wave & (PW - 1), kw = wave >> (2 - lkw);
    const i32x4* const sTab = reinterpret_cast<const i32x4*>(smem + ((np - 1) & 1) * S3_TABB);
    const unsigned char* const tabB = reinterpret_cast<const unsigned char*>(sTab) + ((threadIdx.x & 32) ? 4 : 0);     // a lane's LDS-offset word of an entry
    float* const sScr = reinterpret_cast<float*>(smem + S3_SCR);
    unsigned char* const ring = smem + S3_FIXED;
    const int Wp = p.Wp, Sp = p.Sp, W = p.W, H = p.H, TH = p.TH, NI = p.NI, nst = p.nst;
    if (dbg && t == 0 && g == 0) dbg[8] = (int)clock64();

    // wave w's k-steps of stage s: an equal share of the stage's table entries
    auto range_of = [&](int s, int w, int& jb, int& je) {
        const int a = st_begin(p, s), b = st_end(p, s);
        const int c = (b - a + KW - 1) >> lkw;
        jb = min(b, a + w * c); je = min(b, jb + c);
    };
    const int thw = TH * W, npix = NI * thw;
    const int osz = p.out_fp32 ? 4 : 2;
    int lanebase[MT], outoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int j = (pw * MT + mt) * 16 + li;
        const bool valid = j < npix;
        const int jc = valid ? j : 0;
        int sl, r, xx;
        if (p.lw >= 0) {
            sl = jc >> p.lthw; const int rem = jc & (thw - 1);
            r = rem >> p.lw; xx = rem & (W - 1);
        } else {
            sl = (int)udiv24((unsigned)jc, (unsigned)thw, 1.0f / (float)thw); const int rem = jc - sl * thw;
            r = (int)udiv24((unsigned)rem, (unsigned)W, 1.0f / (float)W); xx = rem - r * W;
        }
        lanebase[mt] = ((sl * (TH + 2) + r + 1) * Wp + xx + 1) * Sp * 16 + (lg & 1) * 16;
        outoff[mt] = valid ? (((sl * H + r) * W + xx) * p.Cout_tot + lg * 4) * osz : OOB;
    }
    if (dbg && t == 0 && g == 0) dbg[11] = (int)clock64();
    const int nks = p.nks;
    const int wlane = lane * 16;

    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }

    // the first stage in which this wave has k-steps, and those
    int jb0 = 0, je0 = 0;
#pragma unroll
    for (int k = S3_MAXST - 1; k >= 0; --k) if (k < nst) { int a, b; range_of(k, kw, a, b); if (a < b) { jb0 = a; je0 = b; } }

    Pos cur{ii, first_tile_of(p.wg0, p.nwg, bid, nblk), 0, true, true};
    // The ring lives across the tiles of a problem visit: the refills issued during a tile's last D k-steps follow the table's
    // wrap-around links and fetch the FIRST D k-steps of the next tile (same weights whenever the workgroup's next tile has the
    // same channel block -- always for the grids used: the tile stride is a multiple of the channel-block count), so that tile
    // starts with its fragments already in flight behind the previous epilogue instead of waiting ~2.7 k cycles for a refill.
    static_assert(D % 2 == 0, "ring slot d pairs with pixel-fragment buffer d & 1");
    bf16x8 A[D][NT];
    bf16x8 Bq[2][MT];
    i32x2 qn{};                                       // {z, w} of the table entry of the next refill
    int r0 = 0;                                       // ring slot of the next k-step
    bool primed = false;                              // the ring already holds (or is fetching) this tile's first k-steps
    for (; cur.ii == ii && cur.valid;) {
        const int tau = cur.tau;
        int img0, y0, nb;
        tile_coords(p, tau, img0, y0, nb);
        const int n0 = nb * (16 * NT);                               // row of the packed weights ([group][rows_pad / 16][k-step] fragments)
        [[maybe_unused]] int since = D, cq = 0;                       // ring waits since this wave last queued stage copies, and how many (ring_wait)
        const bf16_t* wblk = p.w + (size_t)(n0 / 16) * (size_t)nks * 512;
        int cb = n0, clim = p.Cout;                                  // the block's first output channel / the end of its group's channels
        if (p.groups > 1) {
            const int gq = __builtin_amdgcn_readfirstlane((int)udiv24((unsigned)nb, (unsigned)p.nnbg, p.rc_nnbg));
            cb = gq * p.Cout + (nb - gq * p.nnbg) * (16 * NT); clim = (gq + 1) * p.Cout;
        }
        // Weight-fragment loads are inline asm: the compiler's own vmcnt bookkeeping drains the whole ring at every loop
        // header (vmcnt(0) once per D k-steps, measured in the disassembly).  Invisible to it, they are counted by hand: the
        // ring is a FIFO, so whenever a slot is used exactly NT * (D - 1) younger fragment loads exist -- s_waitcnt
        // vmcnt(NT * (D - 1)) (ring_wait) is enough, and any other memory operation in between only makes it more
        // conservative.  The varying part of the address travels in the vector offset (a VALU-written scalar offset would
        // need wait states the compiler does not insert inside asm).
        const i32x4 wdesc = raw_desc(wblk, NT * nks * 1024);
        int wso[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wso[nt] = __builtin_amdgcn_readfirstlane(nt * nks * 1024);
        auto load_a = [&](int voff, bf16x8* a) {                   // voff: lane * 16 + byte offset of the k-step's fragment of row block 0
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a[nt]) : "v"(voff), "s"(wdesc), "s"(wso[nt]) : "memory");
        };
        // `since` ring waits ago this wave queued `cq` stage copies.  The slot a k-step uses was requested D k-steps earlier: for the first D
        // k-steps after the copies that is BEFORE them, and everything younger -- NT (D - 1) fragment loads and the copies -- may still be
        // in flight (loads return in issue order: tools/experiments/dma_order.hip): stepping over the copies instead of sitting them out
        // (S3_STEP_OVER; the count is rounded down to a multiple of 7: a smaller allowance only waits longer).  Every alternative is ONE
        // s_waitcnt behind a scalar branch -- the k-step itself exists once (a duplicated ring turn is what broke the stem kernel's
        // first version: DESIGN.md 3.1).
        // (The alternatives are operand-free waits followed by ONE statement that hands the slot's registers to the compiler: alternatives
        // that each carried the registers as "+v" operands made them phi-joined values, and the copies the compiler may then insert read
        // registers whose loads are still in flight -- wrong results, measured.)
        auto ring_wait = [&](bf16x8* a) {
            constexpr int N0 = NT * (D - 1);
            static_assert(N0 + 28 <= 63, "vmcnt");
#if S3_STEP_OVER
            if (since < D) {
                ++since;
                if (cq >= 28) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0 + 28) : "memory");
                else if (cq >= 21) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0 + 21) : "memory");
                else if (cq >= 14) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0 + 14) : "memory");
                else if (cq >= 7) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0 + 7) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0) : "memory");
            } else
#endif
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0) : "memory");
            if constexpr (NT == 1) asm volatile("" : "+v"(a[0]));
            if constexpr (NT == 2) asm volatile("" : "+v"(a[0]), "+v"(a[1]));
            if constexpr (NT == 3) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
        };
        // The ring runs over this wave's k-steps of the whole tile (stage after stage): k-step i sits in slot i % D and, once
        // used, the slot is refilled with k-step i + D.  The prefetch walks the table's successor links D k-steps ahead of
        // the MFMAs; the entry of the NEXT refill is read one k-step early.
        if (primed) {
            // (nothing to do: slots r0, r0 + 1, ... hold k-steps 0, 1, ... of this tile and qn the entry of the next refill)
        } else if (jb0 + D < je0) {                   // (the common case: the first D k-steps are consecutive entries)
            r0 = 0;
            i32x4 q[D];
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] = sTab[jb0 + d];
            qn = i32x2{sTab[jb0 + D].z, sTab[jb0 + D].w};
#pragma unroll
            for (int d = 0; d < D; ++d) load_a(wlane + q[d].z, A[d]);
        } else {
            r0 = 0;
            int pj = jb0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const i32x4 q = sTab[pj];
                load_a(wlane + q.z, A[d]);
                pj = __builtin_amdgcn_readfirstlane(q.w);
            }
            qn = i32x2{sTab[pj].z, sTab[pj].w};
        }
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (dbg && t == 0 && g == 0) dbg[12] = (int)clock64();

        for (int s = 0; s < nst; ++s) {
            // the stage after this one starts travelling into the slot the last barrier freed (after the ring's own loads
            // of this point: see the header)
            const int tq0 = dbg ? (int)clock64() : 0;      // (debug launches only: per-workgroup sums over ALL stages -- copy issue, k-steps, tile end)
            const Pos nxt = pos_next(nprob, bid, nblk, rot, cur);
            cq = issue_pos(nprob, rot, nxt, smem, ring + ((g + 1) & 1) * S3_BUF, np, wave, lane);
            since = 0;
            if (dbg && t == 0 && g == 0) dbg[2] = (int)clock64();
            const int tq1 = dbg ? (int)clock64() : 0;
            const unsigned char* const sX = ring + (g & 1) * S3_BUF;
            int j, je;
            range_of(s, kw, j, je);
            const int nsteps = je - j;
            // Pixel fragments are double-buffered in registers: k-step jj issues the LDS reads of k-step jj + 1 BEFORE its own MFMAs
            // (a wave alone on its SIMD measured ~590 cycles per k-step against 192 of MFMA work: the ds_read latency of every
            // k-step sat in front of its MFMAs).  Ring slot d pairs with buffer d & 1 (D is even), so all indices stay static.
            // The first k-step of a stage reads its fragments unpipelined (the slot was published by the last barrier); the
            // last one pre-reads a clamped entry nobody uses.
            // Table reads are slim and come FIRST in a k-step (a lane reads the one LDS offset it needs: .x for lanes 0-31, .y for
            // 32-63; the refill entry only its {z, w} half): LDS returns in order, so the MFMAs' wait for the current fragments
            // and the next k-step's wait for its table words both leave the four younger fragment reads in flight.
            int ek = *reinterpret_cast<const int*>(tabB + min(j, p.nent - 1) * 16);
            {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { Bq[0][mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + ek); Bq[1][mt] = Bq[0][mt]; }
                ek = *reinterpret_cast<const int*>(tabB + min(j + 1, max(je - 1, 0)) * 16);
            }
            // One k-step on ring slot `a`: MT * NT MFMAs with everything else of the k-step slotted BETWEEN them, one small group
            // per MFMA (sched_barrier(0) pins the order).  A wave issues in order: with the ~28 other instructions of a k-step
            // in front of its MFMA block a wave alone on its SIMD needed ~500 cycles per k-step for 192 cycles of MFMA work;
            // behind an MFMA (4 cycles to issue, 16 in the pipe) three more instructions issue for free.  Slots, by MFMA index
            // i = nt * MT + mt: i < MT: the NEXT k-step's pixel fragment i (LDS); after the last MFMA that reads a[nt]: that
            // fragment's refill; i = MT, MT + 1 (NT = 1: MT - 1): the table words of the k-step after the next.
            auto kstep = [&](const int jj, bf16x8* a, const bf16x8* bc, bf16x8* bn) {
                const int koff = ek;
                int voffn = 0;
                auto table_next = [&](int part) {
                    if (part == 0) {
                        voffn = wlane + qn.x;
                        qn = *reinterpret_cast<const i32x2*>(reinterpret_cast<const unsigned char*>(sTab) + __builtin_amdgcn_readfirstlane(qn.y) * 16 + 8);
                    } else {
                        ek = *reinterpret_cast<const int*>(tabB + min(jj + 2, je - 1) * 16);
                    }
                };
                ring_wait(a);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], bc[mt], acc[mt][nt], 0, 0, 0);
                        if (nt == 0) bn[mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + koff);
                        if (nt == 0 && mt == MT - 1) table_next(0);
                        if (NT == 1 ? (mt == MT - 1) : (nt == 1 && mt == 0)) table_next(1);
                        if (mt == MT - 1)
                            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a[nt]) : "v"(voffn), "s"(wdesc), "s"(wso[nt]) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
            };
            // head: up to D - 1 k-steps that bring the ring back to slot 0; groups of D without any condition inside; tail
#pragma unroll
            for (int d = 1; d < D; ++d)
                if (r0 == d && j < je) { kstep(j, A[d], Bq[d & 1], Bq[(d + 1) & 1]); ++j; r0 = (d + 1) % D; }
            if (r0 == 0) {
                for (; j + D <= je; j += D) {
#pragma unroll
                    for (int d = 0; d < D; ++d) kstep(j + d, A[d], Bq[d & 1], Bq[(d + 1) & 1]);
                }
#pragma unroll
                for (int d = 0; d < D - 1; ++d)
                    if (r0 == d && j < je) { kstep(j, A[d], Bq[d & 1], Bq[(d + 1) & 1]); ++j; r0 = d + 1; }
            }
            cur = nxt;
            // the copies requested at this stage's start are older than the refills of its k-steps: D - 1 k-steps later a
            // ring_wait has covered them; a shorter stage waits explicitly
            if (dbg) { const int tq2 = (int)clock64(); dsum[0] += tq1 - tq0; dsum[1] += tq2 - tq1; }
            if (nsteps < D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if S3_STRICT_COPIES
            // (build knob, off) publish the slot after a full wait instead of on the ring waits' in-order argument; the ring's slots stay
            // valid, their latest refills are simply waited for here instead of D - 1 k-steps later
            if (s + 1 < nst) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]));
                    if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]));
                    if constexpr (NT == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]));
                }
            }
#endif
            if (s + 1 < nst) { lds_barrier(); ++g; }          // slot consumed; the next stage's slot is complete
        }
        // The ring's last refills are still in flight.  When the workgroup's next tile belongs to this problem and has the same
        // channel block they are that tile's first k-steps: the slots simply stay live across the epilogue (loop-carried, so the
        // compiler keeps the registers).  Otherwise nobody needs them, but the compiler knows nothing of them and would hand
        // their destination registers to the epilogue: every slot stays live up to a full wait.
        primed = false;
        if (S3_WRAP && cur.valid && cur.ii == ii) {
            int i2, y2, nb2;
            tile_coords(p, cur.tau, i2, y2, nb2);
            primed = nb2 == nb;
        }
        if (!primed) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]));
                if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]));
                if constexpr (NT == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[d][0]), "+v"(A[d][1]), "+v"(A[d][2]));
            }
        }
        const dbg_ptr stamp = (dbg && t == 0 && g + 1 == nst) ? dbg : nullptr;       // the workgroup's first tile
        if (stamp) stamp[3] = (int)clock64();
        const int tq3 = dbg ? (int)clock64() : 0;
        unsigned char* const sR = ring + (g & 1) * S3_BUF;
        const bool flush = flush_after(p, tau, nblk);
        if (S3_PLAIN_EPI && !p.bias && !p.addend && !p.relu && !p.out_fp32 && !p.has_idle) {
            switch (KW) {
                case 1: s3_finish<NT, 1, true>(p, acc, s1, s2, outoff, sR, sScr, img0, y0, cb, clim, flush, bid, stamp); break;
                case 2: s3_finish<NT, 2, true>(p, acc, s1, s2, outoff, sR, sScr, img0, y0, cb, clim, flush, bid, stamp); break;
                default: s3_finish<NT, 4, true>(p, acc, s1, s2, outoff, sR, sScr, img0, y0, cb, clim, flush, bid, stamp); break;
            }
        } else {
            switch (KW) {
                case 1: s3_finish<NT, 1, false>(p, acc, s1, s2, outoff, sR, sScr, img0, y0, cb, clim, flush, bid, stamp); break;
                case 2: s3_finish<NT, 2, false>(p, acc, s1, s2, outoff, sR, sScr, img0, y0, cb, clim, flush, bid, stamp); break;
                default: s3_finish<NT, 4, false>(p, acc, s1, s2, outoff, sR, sScr, img0, y0, cb, clim, flush, bid, stamp); break;
            }
        }
        lds_barrier();                                              // the last stage's slot is consumed
        if (stamp) stamp[6] = (int)clock64();
        if (dbg) dsum[2] += (int)clock64() - tq3;
        ++g;
    }
}

// ---- the BatchNorm that follows, applied by the same launch (conv -> train-mode BatchNorm -> [+ residual] -> [ReLU]) ----------
// /root/reference/models/module/res_module.py:39-56 (BasicBlock: conv1 -> bn1 -> relu, conv2 -> bn2 -> + residual -> relu), the HRNet
// branch layers of hr_module.py:155-177.  After a workgroup's last tile: every statistic of the launch has been added to the
// replicas (agent-scope atomics, waited for), a grid-wide barrier (grid_barrier.h: no fences -- what crosses it are those atomics, read
// back with agent-scope loads), then every workgroup walks ITS OWN tiles again -- the rounded bf16 outputs it stored minutes of
// cycles ago sit in its XCD's L2 (a workgroup reads only what it wrote itself: plain loads are coherent) -- and writes
// out = [relu](fmaf(y, sc, sh) [+ res]) and the ReLU gate bytes exactly as norm_act.hip's bn_apply_body would in a launch of its
// own: same operand rounding, same expressions (the contraction of the variance is pinned to what that kernel compiles to: fma(-mean,
// mean, E[x^2]); the running statistics are four rounded products and two rounded sums there).  The tile that holds a channel
// block's first pixels (pixel tile 0) also writes mean / invstd for the backward pass and updates the running statistics: once per
// channel.  The launch must be co-resident (<= 2 workgroups per compute unit, nothing else on the device): the caller's contract,
// as for the one-pass BatchNorm backward, with the same bounded spin and error word.
template <typename T> __device__ inline T ld_acc_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NT>
__device__ __forceinline__ void s3_bn_tail(const int nprob, const int rot, const int bid, const int nblk, unsigned char* smem, dbg_ptr dbg)
{
    constexpr int NC = 16 * NT, CG = 4 * NT, U = 4 * NT;                // channels / 4-channel groups of a channel block; items per lane of a full tile
    constexpr int CMAX = 384, RMAX = BN_NCOPY / DANET_BN_NCOPY_DIV;     // channels per problem (checked at launch); replicas at most (bn_ncopy)
    static_assert(S3_MAXP * 2 * CMAX * 4 <= S3_BUF, "s3_bn_tail scratch");
    const int t = threadIdx.x;
    float* const sTab = reinterpret_cast<float*>(smem + S3_FIXED);        // [problem visit][scale | shift][CMAX] (the ring is free now)
    const float momentum = s3_bn_args()->momentum, eps = s3_bn_args()->eps;
    // ---- every channel's scale and shift of the problems this workgroup has tiles of: ONE round of replica loads per problem ------
    // (a workgroup usually stays with one problem: s3_assign.)  Lane = channel: its 2 x ncopy accumulators are requested at once and
    // added in replica order in double precision, as reduce_replicas does (norm_act.hip).
    for (int ii = 0; ii < nprob; ++ii) {
        const int idx = wrap_idx(ii + rot, nprob);
        const int tau0 = first_tile_of(S3_FIELD(idx, wg0), S3_FIELD(idx, nwg), bid, nblk);
        if (tau0 >= S3_FIELD(idx, ntiles)) continue;
        const S3Prob p = desc_prob(idx);
        const S3Bn b = desc_bn(idx);
        const int C = p.Cout_tot;
        const bn_acc_t* const acc = reinterpret_cast<const bn_acc_t*>(p.stats);
        const int ncopy = bn_ncopy(C);
        for (int c = t; c < C; c += 256) {
            bn_acc_t v1[RMAX], v2[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                v1[r] = r < ncopy ? ld_acc_agent(acc + (size_t)r * 2 * C + c) : (bn_acc_t)0;
                v2[r] = r < ncopy ? ld_acc_agent(acc + (size_t)r * 2 * C + C + c) : (bn_acc_t)0;
            }
            bn_acc_t s1 = 0, s2 = 0;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) { s1 += v1[r]; s2 += v2[r]; }
            float sc, sh, mean, var, invstd;
            {
#pragma clang fp contract(off)
                mean = (float)s1 * b.inv_count;
                const float ex2 = (float)s2 * b.inv_count;
                var = fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f);
                invstd = rsqrtf(var + eps);
                sc = invstd * b.gamma[c];
                sh = __builtin_fmaf(-mean, sc, b.beta[c]);
            }
            sTab[(ii * 2 + 0) * CMAX + c] = sc; sTab[(ii * 2 + 1) * CMAX + c] = sh;
            // the workgroup that holds pixel tile 0 of the channel's block writes the saved statistics and the running averages: once per channel
            const int nb = c / NC, tt0 = p.swz ? nb * 8 : nb;
            int d = bid - p.wg0; if (d < 0) d += nblk;
            if (tt0 % p.nwg == d) {
#pragma clang fp contract(off)
                b.saved[c] = mean; b.saved[C + c] = invstd;
                if (b.running_mean) {
                    const float keep = 1.f - momentum;
                    const float a0 = keep * b.running_mean[c], a1 = momentum * mean;
                    b.running_mean[c] = a0 + a1;
                    const float q0 = keep * b.running_var[c], q1 = (momentum * var) * b.unbias;
                    b.running_var[c] = q0 + q1;
                }
            }
        }
    }
    __syncthreads();
    if (dbg && t == 0) dbg[5] = (int)clock64();
    // ---- the workgroup's own tiles again: a tile's loads (U outputs + U residual pieces per lane) are all in flight before the first
    // one is used, and the NEXT tile's are requested before this tile is computed (two register sets, the tile loop unrolled by two)
    for (int ii = 0; ii < nprob; ++ii) {
        const int idx = wrap_idx(ii + rot, nprob);
        const int tau0 = first_tile_of(S3_FIELD(idx, wg0), S3_FIELD(idx, nwg), bid, nblk);
        if (tau0 >= S3_FIELD(idx, ntiles)) continue;
        const S3Prob p = desc_prob(idx);
        const S3Bn b = desc_bn(idx);
        const int Ctot = p.Cout_tot, clim = p.Cout;                     // (groups == 1: checked at launch)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(b.res ? b.res : (const bf16_t*)p.y), 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(b.out, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(b.mask ? (void*)b.mask : (void*)p.y, 0, b.mask ? p.y_bytes >> 3 : 0, 0x00020000);
        const bool has_res = b.res != nullptr, has_mask = b.mask != nullptr, relu = b.relu != 0;
        const int thw = p.TH * p.W, npix = p.NI * thw, hw = p.H * p.W, pixb = Ctot * 2;
        const float rc_thw = 1.0f / (float)thw;
        const int nitems = npix * CG;
        const float* const tab = sTab + ii * 2 * CMAX;
        // a lane's items of a tile: (pixel, channel group) does not depend on the tile -- offsets relative to the tile's first pixel and
        // channel block, once per problem
        int rel[U], cgo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = t + u * 256;
            const int pix = i / CG, cg = i - pix * CG;
            int sl, rem;
            if (p.lw >= 0) { sl = pix >> p.lthw; rem = pix & (thw - 1); }
            else { sl = (int)udiv24((unsigned)pix, (unsigned)thw, rc_thw); rem = pix - sl * thw; }
            rel[u] = i < nitems ? (sl * hw + rem) * pixb + cg * 8 : OOB;
            cgo[u] = cg * 4;
        }
        auto issue = [&](int tau, int (&off)[U], i32x2 (&xq)[U], i32x2 (&rq)[U], int& cb) {
            int img0, y0, nb;
            tile_coords(p, tau, img0, y0, nb);
            cb = nb * NC;
            const int base = ((img0 * p.H + y0) * p.W) * pixb + cb * 2;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                off[u] = (rel[u] != OOB && cb + cgo[u] < clim) ? base + rel[u] : OOB;
                xq[u] = __builtin_amdgcn_raw_buffer_load_b64(xr, off[u], 0, 0);
            }
            if (has_res) {
#pragma unroll
                for (int u = 0; u < U; ++u) rq[u] = __builtin_amdgcn_raw_buffer_load_b64(rr, off[u], 0, 0);
            }
        };
        auto finish = [&](const int (&off)[U], const i32x2 (&xq)[U], const i32x2 (&rq)[U], int cb) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c0 = min(cb + cgo[u], CMAX - 4);                 // (idle lanes read a valid slot)
                const f32x4 sc = *reinterpret_cast<const f32x4*>(&tab[c0]), sh = *reinterpret_cast<const f32x4*>(&tab[CMAX + c0]);
                const float a[4] = {__uint_as_float((unsigned)xq[u].x << 16), __uint_as_float((unsigned)xq[u].x & 0xffff0000u),
                                    __uint_as_float((unsigned)xq[u].y << 16), __uint_as_float((unsigned)xq[u].y & 0xffff0000u)};
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                if (has_res) {
                    r[0] = __uint_as_float((unsigned)rq[u].x << 16); r[1] = __uint_as_float((unsigned)rq[u].x & 0xffff0000u);
                    r[2] = __uint_as_float((unsigned)rq[u].y << 16); r[3] = __uint_as_float((unsigned)rq[u].y & 0xffff0000u);
                }
                int mb = 0;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = __builtin_fmaf(a[j], sc[j], sh[j]);
                    if (has_res) v += r[j];
                    mb |= (v > 0.f ? 1 : 0) << j;
                    o[j] = relu ? fmaxf(v, 0.f) : v;
                }
                const i32x2 pk = {(int)f2bf_pk(o[0], o[1]), (int)f2bf_pk(o[2], o[3])};
                __builtin_amdgcn_raw_buffer_store_b64(pk, yr, off[u], 0, 0);
                if (has_mask) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)mb, mr, off[u] == OOB ? OOB : off[u] >> 3, 0, 0);
            }
        };
        int offA[U], offB[U], cbA = 0, cbB = 0;
        i32x2 xA[U], rA[U], xB[U], rB[U];
        int tau = tau0;
        issue(tau, offA, xA, rA, cbA);
        for (;;) {
            const int t1 = tau + p.nwg;
            if (t1 < p.ntiles) issue(t1, offB, xB, rB, cbB);
            finish(offA, xA, rA, cbA);
            if (t1 >= p.ntiles) break;
            const int t2 = t1 + p.nwg;
            if (t2 < p.ntiles) issue(t2, offA, xA, rA, cbA);
            finish(offB, xB, rB, cbB);
            if (t2 >= p.ntiles) break;
            tau = t2;
        }
    }
}

template <int NT, bool WITH_BN>
__device__ __forceinline__ void s3_kernel_body()
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s3_smem[];
    const int bid = blockIdx.x, nblk = gridDim.x;
    const int nprob = s3_args()->n;
    // every workgroup visits the problems in its own rotation (the two workgroups of a CU, ids 256 apart, are one problem apart)
    const unsigned rv = (unsigned)(bid + bid / 256);
    const int rot = __builtin_amdgcn_readfirstlane((int)(rv - (unsigned)nprob * udiv24(rv, (unsigned)nprob, 1.0f / (float)nprob)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int* const dbg0 = s3_args()->dbg;
    const dbg_ptr dbg = dbg0 ? (dbg_ptr)(dbg0 + bid * 16) : nullptr;
    if (dbg && threadIdx.x == 0) {
        dbg[0] = (int)clock64();
        dbg[15] = (int)((__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xff00u) | ((__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u) << 16));   // CU: HW_ID cu / sh / se, XCC_ID
    }
    const Pos first = pos_first(nprob, bid, nblk, rot, 0);
    if (!WITH_BN && !first.valid) return;
    if (first.valid) {
        issue_pos(nprob, rot, first, s3_smem, s3_smem + S3_FIXED, 0, wave, lane);      // the very first stage: nothing to overlap it with
        if (dbg && threadIdx.x == 0) dbg[9] = (int)clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                                                  // ... published
        if (dbg && threadIdx.x == 0) dbg[10] = (int)clock64();
        int g = 0, np = 0;
        int dsum[3] = {0, 0, 0};
        for (int ii = first.ii; ii < nprob; ++ii) {
            const int idx = wrap_idx(ii + rot, nprob);
            if (first_tile_of(S3_FIELD(idx, wg0), S3_FIELD(idx, nwg), bid, nblk) >= S3_FIELD(idx, ntiles)) continue;
            ++np;
            s3_problem<NT>(nprob, ii, rot, bid, nblk, g, np, s3_smem, dbg, dsum);
        }
        if (dbg && threadIdx.x == 0) { dbg[7] = (int)clock64(); dbg[1] = dsum[0]; dbg[13] = dsum[1]; dbg[14] = dsum[2]; }
    }
    if constexpr (WITH_BN) {
        // every output and every statistic of this workgroup has left (stores, atomics and the inline-asm loads the compiler does not count)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        danet::grid_barrier(s3_bn_args()->bar, (unsigned)nblk, 0x40000u + (unsigned)nblk);
        if (dbg && threadIdx.x == 0) dbg[8] = (int)clock64();
        s3_bn_tail<NT>(nprob, rot, bid, nblk, s3_smem, dbg);
        if (dbg && threadIdx.x == 0) dbg[12] = (int)clock64();
    }
}
template <int NT>
__global__ __launch_bounds__(S3_THREADS, 2) void conv3x3_stream_kernel(S3Launch L) { s3_kernel_body<NT, false>(); }
// the same kernel with the BatchNorm tail (a kernel of its own: the plain one keeps its code, and profiles tell the two apart)
template <int NT>
__global__ __launch_bounds__(S3_THREADS, 2) void conv3x3_stream_bn_kernel(S3LaunchBn L) { s3_kernel_body<NT, true>(); }

bool g_s3_on = getenv("DANET_NO_C3_STREAM") == nullptr;
int g_s3_blocks = getenv("DANET_C3S_BLOCKS") ? atoi(getenv("DANET_C3S_BLOCKS")) : 512;
int g_s3_kw = getenv("DANET_C3S_KW") ? atoi(getenv("DANET_C3S_KW")) : 0;          // forced K split (tests, A-B timing); 0: the planner's choice
int g_s3_want = getenv("DANET_C3S_WANT") ? atoi(getenv("DANET_C3S_WANT")) : 0;     // tiles per problem the planner aims for (0: 512 / problems of the launch)

int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

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
    const int nc16 = p.Cin_g / 16, nrows = NI * (TH + 2), Wp = W + 2;
    const int nks = p.Kp / 32;
    if (nks > S3_TAB) return false;
    if (KW > 1 && 4 * (MT - MT / KW) * NT * 1024 > S3_BUF) return false;
    int nst = 0, per = 0, Sp = 0;
    for (int n = 1; n <= S3_MAXST; ++n) {
        int c = (nc16 + n - 1) / n;
        if (n > 1) { if (nc16 % 2) return false; c += c & 1; }
        int sp = 2 * c;
        while (sp % 4 != 2) ++sp;
        const long cells = (long)nrows * Wp * sp;
        if (Wp * sp > S3_NIR * 64) continue;                             // (a row is at most S3_NIR copy instructions)
        if (cells * 16 <= S3_BUF) { nst = (nc16 + c - 1) / c; per = c; Sp = sp; break; }
    }
    if (nst == 0) return false;
    q.nst = nst;
    q.st_c0w = q.st_ncw = q.st_j0w = 0;
    int j0 = 0;
    for (int s = 0; s < nst; ++s) {
        const int c0 = s * per, nc = nc16 - c0 < per ? nc16 - c0 : per;
        const int ne = nst == 1 ? nks : 9 * nc / 2;
        j0 += ne;
        q.st_c0w |= (unsigned)c0 << (8 * s); q.st_ncw |= (unsigned)nc << (8 * s); q.st_j0w |= (unsigned)j0 << (8 * s);
    }
    if (j0 != nks) return false;
    q.nent = j0;
    q.TH = TH; q.NI = NI; q.Wp = Wp; q.Sp = Sp;
    q.tiles_h = H / TH; q.nnbg = p.Cout_pad / (16 * NT); q.nnb = q.nnbg * p.groups; q.nks = nks; q.nc16 = nc16;
    q.rc_nnbg = 1.0f / (float)q.nnbg;
    q.rc_nnb = 1.0f / (float)q.nnb; q.rc_th = 1.0f / (float)q.tiles_h;
    q.lw = ilog2_exact(W); q.lthw = ilog2_exact(TH * W);
    if (q.lw < 0 || q.lthw < 0) q.lw = q.lthw = -1;
    const int npt = (p.B / NI) * q.tiles_h;
    q.ntiles = npt * q.nnb;
    q.kw = KW; q.nt = NT;
    q.swz = (npt % 8 == 0 && q.nnb > 1) ? 1 : 0;
    q.has_idle = (NI * TH * W != TP || p.Cout_g != p.Cout_pad) ? 1 : 0;
    return true;
}

bool s3_shape_ok(const ConvP& p) {
    if (!g_s3_on) return false;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1) return false;
    // grouped layers (the 24-group partial-IUV head, /root/reference/models/danet/iuv_estimator.py:193-206): a tile = one group's
    // channels of a pixel tile; forward only (the data gradient's 24 channels per group are no multiple of 16)
    static const bool no_grouped = getenv("DANET_NO_C3S_GROUPED") != nullptr;      // A-B timing knob
    if (p.groups != 1 && (no_grouped || p.transposed || p.addend)) return false;
    if (p.H != p.OH || p.W != p.OW || p.Cin_g % 16 != 0 || p.Cout_g % 4 != 0) return false;
    if (p.x_bytes >= (1L << 30) || p.y_bytes >= (1L << 31)) return false;       // (rows outside the image add 2^30 to their offsets: see s3_rows)
    if ((long)p.groups * p.Cout_pad * p.Kp * 2 >= (1L << 31)) return false;
    if ((long)p.B * p.H * p.W >= (1L << 24)) return false;
    if (p.bn_red) return false;                     // (the fused BatchNorm-backward reduction stays on conv3x3.hip)
    return danet_conv_nt(p.Cout_g) <= 3;
}

// The first K split that gives a launch of nprob problems enough tiles, else the one with the most tiles: fewer, larger tiles
// with less K splitting win as long as every workgroup gets work (measured on the four HRNet branches, tools/c3s_bench.py:
// 39.2 us with 512 + 3 x 256 tiles, 42.6 us with the per-workgroup load balanced through smaller tiles, 43.1 us with 512 each).
bool s3_plan(const ConvP& p, S3Prob& q, int nprob) {
    const int NT = danet_conv_nt(p.Cout_g);
    static const int cand[3] = {1, 2, 4};
    const int want = g_s3_want > 0 ? g_s3_want : (2 * 256 + nprob - 1) / nprob;
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

// Which workgroups take which problem's tiles (S3Prob.wg0 / nwg).  A tile's cost ~ its k-steps per wave (nks / KW) plus a fixed
// share for staging, exchange and epilogue (g_s3_tile_cost k-step equivalents).  Every problem starts with one block of 32
// workgroups (32: a multiple of the 8 XCDs and of every channel-block count, which the kernel's weight prefetch across tiles and the
// XCD swizzle of tile_coords rely on); the remaining blocks go, one at a time, to the problem whose workgroups currently carry the
// most work -- exclusive ranges, so a workgroup stays with one problem (one tap table, one set of weights).  A launch with a
// single problem, or with fewer than 32 workgroups per problem, keeps the plain deal (the concatenated list over all workgroups).
// Measured at B = 32 (tools/c3s_balance.py, profiles/r05_c3s_balance.txt; us per launch, plain deal -> this one, outputs bit-identical):
// two branches 25.4 -> 21.8 forward + statistics, 22.8 -> 20.6 data gradient; four branches 38.5 -> 35.9 and 35.7 -> 33.8; THREE
// branches 31.5 -> 31.3 and 28.6 -> 29.9 -- there the plain deal already gives every workgroup one 48-channel tile plus one tile of
// a deeper branch, and keeps it: balance = 1 skips three-problem launches, 2 applies it to every launch (A-B knob).
int g_s3_balance = getenv("DANET_C3S_BALANCE") ? atoi(getenv("DANET_C3S_BALANCE")) : 1;
int g_s3_tile_cost = getenv("DANET_C3S_TILE_COST") ? atoi(getenv("DANET_C3S_TILE_COST")) : 16;
void s3_assign(S3Launch& L, int grid) {
    const int n = L.n;
    constexpr int BLK = 32;
    bool ok = g_s3_balance != 0 && n > 1 && (n != 3 || g_s3_balance == 2) && grid % BLK == 0 && grid / BLK >= n;
    int nwg[S3_MAXP];
    double cost[S3_MAXP];
    if (ok) {
        int left = grid / BLK - n;
        for (int i = 0; i < n; ++i) { nwg[i] = BLK; cost[i] = (double)L.p[i].nks / L.p[i].kw + g_s3_tile_cost; }
        auto load = [&](int i) { return (double)((L.p[i].ntiles + nwg[i] - 1) / nwg[i]) * cost[i]; };
        while (left > 0) {
            int worst = -1;
            for (int i = 0; i < n; ++i)
                if (nwg[i] < L.p[i].ntiles && (worst < 0 || load(i) > load(worst))) worst = i;
            if (worst < 0) break;                                  // every problem already has a workgroup per tile
            // (a block more only helps when it lowers the tiles-per-workgroup count: keep adding until it does, or give up on this problem)
            nwg[worst] += BLK; --left;
        }
        for (int i = 0; i < n; ++i) if (nwg[i] > L.p[i].ntiles) nwg[i] = (L.p[i].ntiles + 7) / 8 * 8 > nwg[i] ? nwg[i] : (L.p[i].ntiles + 7) / 8 * 8;
    }
    if (ok) {
        int w = 0;
        for (int i = 0; i < n; ++i) { L.p[i].wg0 = w; L.p[i].nwg = nwg[i]; w += nwg[i]; }
        if (w > grid) ok = false;
    }
    if (!ok)
        for (int i = 0; i < n; ++i) { L.p[i].wg0 = L.p[i].tile0 % grid; L.p[i].nwg = L.p[i].ntiles < grid ? L.p[i].ntiles : grid; }
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
template <int NT>
void s3_launch_nt(const S3LaunchBn& L, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream_bn_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_stream_bn_kernel<NT>), dim3((unsigned)grid), dim3(S3_THREADS), (size_t)S3_LDS, st, L);
}
template <typename LaunchT>
int s3_launch_any(int NT, const LaunchT& L, int grid, hipStream_t st) {
    switch (NT) {
        case 1: s3_launch_nt<1>(L, grid, st); return 0;
        case 2: s3_launch_nt<2>(L, grid, st); return 0;
        case 3: s3_launch_nt<3>(L, grid, st); return 0;
        default: return -1;
    }
}

}  // namespace

namespace danet_conv {

// n (<= 4) problems in one launch of the streamed kernel.  0 on launch, -1 when the set cannot run here (nothing is
// launched then); dry = true only answers that question.
int conv3x3s_launch(const ConvP* ps, int n, void* stream, bool dry) {
    if (n < 1 || n > S3_MAXP) return -1;
    S3Launch L{};
    L.n = n;
    hipStream_t st = (hipStream_t)stream;
    int tile0 = 0, NT = 0;
    // the BatchNorm tail: every problem of the launch or none; plain forward problems with statistics only
    const bool with_bn = ps[0].bna != nullptr;
    for (int i = 0; i < n; ++i) {
        const ConvP& p = ps[i];
        if ((p.bna != nullptr) != with_bn) return -1;
        if (with_bn) {
            const BnApply& a = *p.bna;
            if (!p.stats || p.groups != 1 || p.transposed || p.bias || p.addend || p.relu || p.out_fp32 || p.bn_red) return -1;
            if (p.Cout > 384 || p.Cout % 4) return -1;                                  // (s3_bn_tail's scale / shift table: CMAX channels per problem)
            if (!a.out || !a.gamma || !a.beta || !a.saved || !ps[0].bna->bar || (a.running_mean == nullptr) != (a.running_var == nullptr)) return -1;
            if (a.momentum != ps[0].bna->momentum || a.eps != ps[0].bna->eps) return -1;
        }
    }
    for (int i = 0; i < n; ++i) {
        const ConvP& p = ps[i];
        if (!s3_shape_ok(p)) return -1;
        S3Prob& q = L.p[i];
        q.x = p.x; q.w = p.w; q.y = p.y; q.bias = p.bias; q.stats = p.stats; q.addend = p.addend;
        q.B = p.B; q.H = p.OH; q.W = p.OW; q.Cin = p.Cin_g; q.Cout = p.Cout_g;
        q.groups = p.groups; q.pixb_in = p.Cin * 2; q.Cout_tot = p.Cout;
        q.flip = p.transposed ? 1 : 0; q.relu = p.relu ? 1 : 0; q.out_fp32 = p.out_fp32 ? 1 : 0;
        q.x_bytes = (int)p.x_bytes; q.y_bytes = (int)p.y_bytes;
    }
    for (int i = 0; i < n; ++i) {
        S3Prob& q = L.p[i];
        if (!s3_plan(ps[i], q, n)) return -1;
        if (NT == 0) NT = q.nt; else if (NT != q.nt) return -1;
        q.tile0 = tile0;
        tile0 += q.ntiles;
    }
    for (int i = 0; i < n; ++i) {                  // (tables last: a set that does not qualify builds none)
        L.p[i].tab = s3_table_for(L.p[i], st, dry);
        if (!L.p[i].tab) return -1;
    }
    L.total = tile0;
    L.dbg = conv3x3_debug_buffer();
    if (dry) return 0;
    static const bool verbose = getenv("DANET_C3S_VERBOSE") != nullptr;
    if (verbose) {
        fprintf(stderr, "[c3s] %d problems:", n);
        for (int i = 0; i < n; ++i) fprintf(stderr, " C%d@%dx%d kw%d st%d tiles%d", L.p[i].Cin, L.p[i].H, L.p[i].W, L.p[i].kw, L.p[i].nst, L.p[i].ntiles);
        fprintf(stderr, "\n");
    }
    const int grid = L.total < g_s3_blocks ? L.total : g_s3_blocks;
    s3_assign(L, grid);
    if (verbose) {
        fprintf(stderr, "[c3s] workgroups:");
        for (int i = 0; i < n; ++i) fprintf(stderr, " %d+%d", L.p[i].wg0, L.p[i].nwg);
        fprintf(stderr, " of %d\n", grid);
    }
    if (with_bn) {
        // the BatchNorm tail crosses a grid-wide barrier: every workgroup of the launch must be resident at once -- at most two per
        // compute unit (the kernel's launch bound and its LDS), which the workgroup cap (512 on the 256 compute units of an MI355X)
        // already says; a cap raised past that by a knob makes the set unfusable rather than a deadlock
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
        if (grid > 2 * cus) return -1;
        S3LaunchBn LB{};
        LB.c = L;
        for (int i = 0; i < n; ++i) {
            const BnApply& a = *ps[i].bna;
            S3Bn& b = LB.bn[i];
            b.res = a.res; b.out = a.out; b.gamma = a.gamma; b.beta = a.beta; b.running_mean = a.running_mean; b.running_var = a.running_var;
            b.saved = a.saved; b.mask = a.mask; b.relu = a.relu;
            const long M = (long)ps[i].B * ps[i].OH * ps[i].OW;
            b.inv_count = 1.0f / (float)M; b.unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;       // (norm_act.hip danet_bn_forward_multi)
        }
        LB.bar = ps[0].bna->bar; LB.momentum = ps[0].bna->momentum; LB.eps = ps[0].bna->eps;
        return s3_launch_any(NT, LB, grid, st);
    }
    return s3_launch_any(NT, L, grid, st);
}

}  // namespace danet_conv

// enable: 0 / 1 (-1 keeps); blocks: workgroup cap of a launch (<= 0 keeps); kw: forced K split 1 / 2 / 4 (0: the planner's
// choice, < 0 keeps); want_tiles: tiles per problem the planner aims for (0: 512 / problems of the launch, < 0 keeps).
// Returns the previous `enable`.
// The device memory the tap tables of the CURRENT device live in (caller-owned, must outlive every launch; the library
// never allocates device memory).  bytes / danet_conv3x3_stream_table_bytes() tables fit; registering again (another
// buffer, or NULL to withdraw it) forgets the cached tables.  Returns the number of tables the workspace holds.
extern "C" size_t danet_conv3x3_stream_table_bytes(void) { return (size_t)S3_TABB; }
extern "C" int danet_conv3x3_stream_tables(void* ws, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    TabPool& pool = g_tab_pools[dev];
    pool.base = (i32x4*)ws;
    pool.cap = ws ? (int)(bytes / S3_TABB) : 0;
    pool.used = 0;
    pool.index.clear();
    return pool.cap;
}
long danet_conv::conv3x3s_knob(int id, long v) {
    long prev = 0;
    switch (id) {
        case DANET_KNOB_C3S_ENABLE: prev = g_s3_on ? 1 : 0; if (v >= 0) g_s3_on = v != 0; break;
        case DANET_KNOB_C3S_BLOCKS: prev = g_s3_blocks; if (v > 0) g_s3_blocks = (int)v; break;
        case DANET_KNOB_C3S_KW: prev = g_s3_kw; if (v >= 0) g_s3_kw = (int)v; break;
        case DANET_KNOB_C3S_WANT: prev = g_s3_want; if (v >= 0) g_s3_want = (int)v; break;
        case DANET_KNOB_C3S_BALANCE: prev = g_s3_balance; if (v >= 0) g_s3_balance = (int)v; break;
        case DANET_KNOB_C3S_TILE_COST: prev = g_s3_tile_cost; if (v >= 0) g_s3_tile_cost = (int)v; break;
        default: break;
    }
    return prev;
}
// What the streamed kernel would do with a problem: KW * 100 + stages * 10 + NT (0: not taken).
extern "C" int danet_conv3x3_stream_plan(int B, int H, int W, int Cin, int Cout, int nprob) {
    ConvP p{};
    p.B = B; p.H = p.OH = H; p.W = p.OW = W; p.Cin = Cin; p.Cout = Cout; p.R = p.S = 3; p.stride = p.pad = p.dil = p.groups = 1;
    p.Cin_g = Cin; p.Cout_g = Cout;
    p.K = 9 * Cin; p.Kp = (p.K + 31) / 32 * 32;
    const int nt = danet_conv_nt(Cout);
    p.Cout_pad = (Cout + 16 * nt - 1) / (16 * nt) * (16 * nt);
    p.x_bytes = (long)B * H * W * Cin * 2; p.y_bytes = (long)B * H * W * Cout * 2;
    if (!s3_shape_ok(p)) return 0;
    S3Prob q{};
    if (!s3_plan(p, q, nprob)) return 0;
    return q.kw * 100 + q.nst * 10 + q.nt;
}
