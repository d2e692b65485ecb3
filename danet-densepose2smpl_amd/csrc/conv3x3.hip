// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) on an LDS-resident halo tile -- the layers that
// hold 89.6 % of the backbone's MACs (every BasicBlock / Bottleneck conv of /root/reference/models/module/
// hr_module.py:15-179 and res_module.py:27-97).
//
// Why a second kernel: conv_fast.hip gathers every B (pixel) fragment from global memory, so each activation
// byte crosses the L1 / texture path nine times (once per tap) and every wave re-reads the weights; at 64 B/clk of
// L1 that path, not the matrix cores, bounded the kernel (9.8 % of the bf16 MFMA rate in round 1).  Here
//   * a persistent workgroup (256 threads) walks a list of tiles; a tile is TH full rows of one image (or NI
//     whole small images) x 16*NT output channels,
//   * the tile's input rows plus a one-pixel halo are copied ONCE into LDS as [slab][row][col][chunk] with 16-byte
//     channel chunks and a per-pixel stride of Sp chunks, Sp = 2 (mod 4), which makes every ds_read_b128 B-fragment
//     read bank-conflict free; zero padding = LDS cells that are never written (pad columns) or written with
//     zeros (rows outside the image), so the k-loop has no masks, selects or bounds logic at all,
//   * a tap is an LDS address offset: per k-step one uniform table read, one select (only needed when Cin is an
//     odd multiple of 16 and a k-step straddles two taps), MT adds, MT ds_read_b128, NT weight-fragment loads
//     (global, fragment-major, through a register ring) and MT*NT MFMAs,
//   * the four waves split the tile's pixels (PW ways) and the K range (KW ways, PW*KW = 4); K-split partial
//     sums meet in LDS.  Small-M layers (192 ch @16x16, 384 ch @8x8) thus still run 64..128-pixel x 48-channel
//     register tiles per wave -- the shape that keeps weight traffic per MFMA low -- on 256 workgroups,
//   * per-lane addresses, the tap table and the BatchNorm-statistics accumulators live across tiles (prologue
//     and cross-lane reductions are paid once per workgroup, not once per tile).
// Weights are the same fragment-major packed operand as conv_fast.hip (mode 0 forward, mode 1 data gradient);
// the data gradient only mirrors the tap offsets.
#include "common.h"
#include "conv_common.h"
#include <type_traits>

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int OOB = 0x7fffffff;
constexpr int TAB_BYTES = 1024;           // tap table: up to 128 k-steps x {lo, hi} offsets
constexpr int NSUB = 2;                   // staging: a row's W*S chunks are covered by NSUB passes of 256 threads
constexpr int PADN = 4;                   // pad-column cells re-zeroed per thread (K-split tiles only)
constexpr int C3_MAXP = 4;

struct C3Prob {
    const bf16_t* x; const bf16_t* w; void* y; const float* bias; float* stats;
    const bf16_t* bn_x; const bf16_t* bn_y; const float* bn_saved; float* bn_red; int bn_gate;
    const bf16_t* addend;     // optional bf16 tensor shaped like y, added before the output is rounded (residual-branch gradient)
    int B, H, W, Cin, Cout, Cout_pad;
    int flip, relu, out_fp32;
    int TH, NI, Wp, S, Sp, tiles_h, npt, nnb, nks, nc16;
    int cfg;                  // MT*100 + NT*10 + KW
    int tile0, ntiles;        // this problem's range in the launch's tile list
    int x_bytes, y_bytes;
    int swz;                  // tile order keeps the N-blocks of a pixel tile on one XCD
    int has_idle;             // some fragment lanes of a tile map to no pixel (NI * TH * W < PW * MT * 16) or channel (Cout < Cout_pad)
    int* dbg;                 // optional [blocks][8] phase timestamps of each workgroup's last tile (tools/c3_bench.py)
};

struct C3Launch { C3Prob p[C3_MAXP]; int n; int total; int stagger; };

__device__ inline unsigned udiv24(unsigned n, unsigned d, float rcp) {      // n < 2^24
    unsigned q = (unsigned)((float)n * rcp);
    const int r = (int)(n - q * d);
    if (r < 0) --q; else if (r >= (int)d) ++q;
    return q;
}

// Workgroup barrier that orders LDS traffic only: global stores of the epilogue keep draining across it
// (__syncthreads() would wait for them: its fence covers every address space).
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
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return v;
}

// s1/s2[nt][r] (per-lane partial sums of channels n0 + nt*16 + lg*4 + r) -> 16 pixel lanes (DPP) -> 4 waves (LDS
// scratch `sc`, >= 4*2*NT*16 floats, free at this point) -> one atomic per channel into replica rep % BN_NCOPY.
template <int NT>
__device__ inline void flush_channel_sums(float (*s1)[4], float (*s2)[4], float* sc, float* __restrict__ dst, int Ctot,
                                          int n0, int t, int li, int lg, int wave, int rep)
{
    lds_barrier();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = row_sum16(s1[nt][r]), b = row_sum16(s2[nt][r]);
            if (li == 0) { sc[(wave * 2 + 0) * (NT * 16) + nt * 16 + lg * 4 + r] = a; sc[(wave * 2 + 1) * (NT * 16) + nt * 16 + lg * 4 + r] = b; }
            s1[nt][r] = 0.f; s2[nt][r] = 0.f;
        }
    lds_barrier();
    if (t < 2 * NT * 16) {
        const int which = t / (NT * 16), c = t - which * (NT * 16);
        const float v = (sc[(0 * 2 + which) * (NT * 16) + c] + sc[(1 * 2 + which) * (NT * 16) + c]) +
                        (sc[(2 * 2 + which) * (NT * 16) + c] + sc[(3 * 2 + which) * (NT * 16) + c]);
        if (n0 + c < Ctot) bn_acc_add(dst, rep, which, Ctot, n0 + c, v);
    }
    lds_barrier();                          // (the scratch overlays tile cells: the next staging pass rewrites them, pad columns included)
}

template <int MT, int NT, int KW>
__device__ __forceinline__ void c3_body(const C3Prob& p, const int bid, const int nblk, unsigned char* smem)
{
    constexpr int PW = 4 / KW;
    constexpr int MO = MT / KW;                    // accumulator tiles a wave finishes (stores, statistics) after the K-split reduction
    constexpr int D = 3;                           // weight-fragment ring: k-steps in flight
    constexpr int RB = 6;                          // staging: rows whose loads are in flight together (a TH = 4 tile is one batch)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave % PW, kw = wave / PW;

    i32x2* const sTab = reinterpret_cast<i32x2*>(smem);
    unsigned char* const sX = smem + TAB_BYTES;

    const int Wp = p.Wp, Sp = p.Sp, S = p.S, W = p.W, H = p.H, TH = p.TH, NI = p.NI;
    const int rowB = Wp * Sp * 16;                 // LDS bytes of one slab row
    const int nrows = NI * (TH + 2);

    if (p.dbg && t == 0) p.dbg[bid * 8 + 0] = (int)clock64();
    // ---- the first tile's input rows are requested before anything else: they travel during the prologue -----------
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    int st_g[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) st_g[sub] = sub * 256 + t < W * S ? (sub * 256 + t) * 16 : OOB;
    // first tile of this problem owned by this workgroup: ids congruent to bid modulo nblk over the launch's list
    int tau = bid - p.tile0 % nblk;
    if (tau < 0) tau += nblk;
    auto tile_coords = [&](int tt, int& img0_, int& y0_, int& nb_) {
        int pt;
        if (p.swz) {                               // tt = (pt_hi * nnb + nb) * 8 + pt_lo
            const int lo = tt & 7, rest = tt >> 3;
            const int hi = rest / p.nnb;
            nb_ = rest - hi * p.nnb; pt = hi * 8 + lo;
        } else {
            pt = tt / p.nnb; nb_ = tt - pt * p.nnb;
        }
        const int bi = pt / p.tiles_h, tb = pt - bi * p.tiles_h;
        img0_ = bi * NI; y0_ = tb * TH;
    };
    // rows [r0, r0 + RB) of slab sl of the tile at (img0, y0): loads into registers / registers into the LDS tile
    auto issue_rows = [&](int img0_, int y0_, int sl, int r0, i32x4 (*v)[NSUB], bool live) {
        const int gimg = ((img0_ + sl) * H) * W * p.Cin * 2;               // byte offset of the image (x_bytes < 2^31)
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int yy = y0_ - 1 + r0 + u;
            const bool rok = live && yy >= 0 && yy < H && r0 + u < TH + 2;
            const int grow = gimg + yy * W * p.Cin * 2;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
                v[u][sub] = rok ? __builtin_amdgcn_raw_buffer_load_b128(xr, st_g[sub], grow, 0) : i32x4{0, 0, 0, 0};
        }
    };
    const bool pre = NI == 1 && TH + 2 <= RB && tau < p.ntiles;        // (one batch of rows: the common case)
    i32x4 xv0[RB][NSUB];
    {
        int i0, y0_, nb_;
        tile_coords(min(tau, p.ntiles - 1), i0, y0_, nb_);
        issue_rows(i0, y0_, 0, 0, xv0, pre);
    }

    // ---- once per problem: tap table, per-lane fragment / staging addresses ----------------------------------------
    lds_barrier();                                 // a previous problem's readers of this LDS are done
    if (t < p.nks) {
        const float rc_n = 1.0f / (float)p.nc16;
        i32x2 e;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int h = 2 * t + half;
            if (h > 9 * p.nc16 - 1) h = 9 * p.nc16 - 1;          // zero-weight tail of the last k-step: any valid cell
            const int tap = (int)udiv24((unsigned)h, (unsigned)p.nc16, rc_n), c16 = h - tap * p.nc16;
            const int r = (tap * 11) >> 5, s = tap - 3 * r;      // tap / 3 for tap < 9
            int off = ((r - 1) * Wp + (s - 1)) * Sp * 16;
            if (p.flip) off = -off;
            off += c16 * 32;
            if (half == 0) e.x = off; else e.y = off;
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
    int st_l[NSUB];
    {
        const float rc_s = 1.0f / (float)S;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int q = sub * 256 + t;
            const bool ok = q < W * S;
            const int pix = (int)udiv24((unsigned)(ok ? q : 0), (unsigned)S, rc_s), c = (ok ? q : 0) - pix * S;
            st_l[sub] = ok ? ((1 + pix) * Sp + c) * 16 : -1;
        }
    }
    auto write_rows = [&](int sl, int r0, i32x4 (*v)[NSUB]) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            if (r0 + u < TH + 2) {
                unsigned char* const lrow = sX + (sl * (TH + 2) + r0 + u) * rowB;
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub)
                    if (st_l[sub] >= 0) *reinterpret_cast<i32x4*>(lrow + st_l[sub]) = v[u][sub];
            }
        }
    };
    // zero padding: the two pad columns of every slab row are (re)zeroed per tile -- staging never writes them, the
    // K-split reduction and the statistics scratch overlay them
    int padaddr[PADN];
    {
        const int npad = nrows * 2 * Sp;
        const float rc_2s = 1.0f / (float)(2 * Sp);
#pragma unroll
        for (int i = 0; i < PADN; ++i) {
            const int q = i * 256 + t;
            int a = -1;
            if (q < npad) {
                const int row = (int)udiv24((unsigned)q, (unsigned)(2 * Sp), rc_2s), rem = q - row * (2 * Sp);
                const int side = rem >= Sp ? 1 : 0, c = rem - side * Sp;
                a = row * rowB + (side ? (Wp - 1) : 0) * Sp * 16 + c * 16;
            }
            padaddr[i] = a;
        }
    }
    lds_barrier();                                 // the tap table is in place
    if (p.dbg && t == 0) p.dbg[bid * 8 + 1] = (int)clock64();

    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int nks = p.nks;
    const int nks_w = (nks + KW - 1) / KW;
    const int jbeg = kw * nks_w, jend = min(nks, jbeg + nks_w);
    const int wlane = lane * 16;

    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    int stat_nb = -1;
    const bool acc_stats = p.stats != nullptr || p.bn_red != nullptr;
    float* const stat_dst = p.stats ? p.stats : p.bn_red;

    auto do_tile = [&](const int tau, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int img0, y0, nb;
        tile_coords(tau, img0, y0, nb);
        const int n0 = nb * (16 * NT);

        if (acc_stats && stat_nb != nb) {
            if (stat_nb >= 0)
                flush_channel_sums<NT>(s1, s2, reinterpret_cast<float*>(sX), stat_dst, p.Cout, stat_nb * (16 * NT), t, li, lg, wave, bid);
            stat_nb = nb;
        }

        // the first weight fragments travel while the tile is staged
        const bf16_t* wblk = p.w + (size_t)(n0 / 16) * (size_t)nks * 512;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wblk), 0, NT * nks * 1024, 0x00020000);
        bf16x8 A[D][NT];
        auto load_a = [&](int j, bf16x8* a) {
            const int jc = min(j, nks - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                a[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, (nt * nks + jc) * 1024, 0));
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load_a(jbeg + d, A[d]);

        // ---- stage the tile: NI slabs of TH+2 rows, interior columns only (pad columns stay zero) -----------------
#pragma unroll
        for (int i = 0; i < PADN; ++i)
            if (padaddr[i] >= 0) *reinterpret_cast<i32x4*>(sX + padaddr[i]) = i32x4{0, 0, 0, 0};
        bool staged = false;
        if constexpr (FIRST) {
            if (pre) { write_rows(0, 0, xv0); staged = true; }
        }
        if (!staged) {
            for (int sl = 0; sl < NI; ++sl)
                for (int r0 = 0; r0 < TH + 2; r0 += RB) {                  // RB rows per batch: every load in flight at once
                    i32x4 v[RB][NSUB];
                    issue_rows(img0, y0, sl, r0, v, true);
                    write_rows(sl, r0, v);
                }
        }
        lds_barrier();
        if (p.dbg && t == 0) p.dbg[bid * 8 + 2] = (int)clock64();

        // ---- k-loop: weights through a register ring, pixels from the LDS tile ------------------------------------
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        i32x2 e = sTab[min(jbeg, nks - 1)];
        for (int j = jbeg; j < jend; j += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (j + d < jend) {
                    const int koff = lg >= 2 ? e.y : e.x;
                    bf16x8 b[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const bf16x8*>(sX + lanebase[mt] + koff);
                    e = sTab[min(j + d + 1, nks - 1)];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[d][nt], b[mt], acc[mt][nt], 0, 0, 0);
                    load_a(j + d + D, A[d]);
                }
            }
        }

        if (p.dbg && t == 0) p.dbg[bid * 8 + 3] = (int)clock64();
        // ---- K-split: partial sums meet in LDS; wave kw finishes accumulator tiles [kw*MO, (kw+1)*MO) --------------
        if constexpr (KW > 1) {
            lds_barrier();                                          // every wave is done reading the tile
            // slot of (source wave, foreign tile f): wave * (MT - MO) * NT + f * NT + nt, 1 KB each
            unsigned char* const myred = sX + (size_t)wave * ((MT - MO) * NT * 1024) + lane * 16;
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                if (q != kw) {                                      // (uniform per wave)
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        const int f = (q < kw ? q : q - 1) * MO + m;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            *reinterpret_cast<f32x4*>(myred + (f * NT + nt) * 1024) = acc[q * MO + m][nt];
                        }
                    }
                }
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                if (q != kw) {                                      // contributions of wave (pw, q) to my tiles
                    const int src = q * PW + pw;
                    const int f0 = (kw < q ? kw : kw - 1) * MO;     // my tiles' foreign index in wave q's slots
                    const unsigned char* const rd = sX + (size_t)src * ((MT - MO) * NT * 1024) + lane * 16;
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

        if (p.dbg && t == 0) p.dbg[bid * 8 + 4] = (int)clock64();
        // ---- epilogue on the wave's own tiles ----------------------------------------------------------------------
        const int tile_out = ((img0 * H + y0) * W) * p.Cout * osz;  // byte offset of the tile's first output pixel
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
                    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, invs = {0.f, 0.f, 0.f, 0.f};
                    if (p.bn_red) {
                        const __amdgpu_buffer_rsrc_t svr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bn_saved), 0, p.Cout * 8, 0x00020000);
                        mean = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(svr, cok ? cl * 4 : OOB, 0, 0));
                        invs = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(svr, cok ? (p.Cout + cl) * 4 : OOB, 0, 0));
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
                            if (p.stats) {
                                // BatchNorm statistics from the fp32 accumulators (two VALU per value): they differ from the
                                // statistics of the bf16-rounded tensor by the mean of zero-mean rounding errors, ~1e-5 relative
                                // over a layer's >= 2048 samples per channel.  Idle fragment lanes (tiles that do not fill the
                                // register tile) hold garbage and are masked out; the common full tiles skip the mask.
                                if (p.has_idle) {
                                    const float msk = off != OOB ? 1.f : 0.f;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { const float q = v[r] * msk; s1[nt][r] += q; s2[nt][r] = fmaf(q, q, s2[nt][r]); }
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { s1[nt][r] += v[r]; s2[nt][r] = fmaf(v[r], v[r], s2[nt][r]); }
                                }
                            } else if (p.bn_red) {
                                const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_x), 0, p.y_bytes, 0x00020000);
                                const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_y ? p.bn_y : p.bn_x), 0, p.y_bytes, 0x00020000);
                                const i32x2 xq = __builtin_amdgcn_raw_buffer_load_b64(bxr, off, so, 0);
                                const float xv[4] = {__uint_as_float((unsigned)xq.x << 16), __uint_as_float((unsigned)xq.x & 0xffff0000u),
                                                     __uint_as_float((unsigned)xq.y << 16), __uint_as_float((unsigned)xq.y & 0xffff0000u)};
                                // the ReLU gate as 4 bits: from the BatchNorm's output (8 bytes per lane) or its byte mask (1 byte)
                                int gm = 15;
                                if (p.bn_gate == 2) {
                                    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bn_y), 0, p.y_bytes >> 3, 0x00020000);
                                    gm = (int)__builtin_amdgcn_raw_buffer_load_b8(mr, off != OOB ? off >> 3 : OOB, so >> 3, 0);
                                } else if (p.bn_y) {
                                    const i32x2 yq = __builtin_amdgcn_raw_buffer_load_b64(byr, off, so, 0);
                                    gm = ((yq.x << 16) > 0 ? 1 : 0) | ((int)((unsigned)yq.x & 0xffff0000u) > 0 ? 2 : 0) |
                                         ((yq.y << 16) > 0 ? 4 : 0) | ((int)((unsigned)yq.y & 0xffff0000u) > 0 ? 8 : 0);
                                }
                                if (off == OOB) gm = 0;
                                const float gq[4] = {__uint_as_float((unsigned)pk.x << 16), __uint_as_float((unsigned)pk.x & 0xffff0000u),
                                                     __uint_as_float((unsigned)pk.y << 16), __uint_as_float((unsigned)pk.y & 0xffff0000u)};
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float gv = (gm >> r) & 1 ? gq[r] : 0.f;
                                    s1[nt][r] += gv; s2[nt][r] += gv * (xv[r] - mean[r]) * invs[r];
                                }
                            }
                        }
                    }
                }
            }
        }
        lds_barrier();                                              // the tile's LDS may be overwritten (stores keep draining)
        if (p.dbg && t == 0) p.dbg[bid * 8 + 5] = (int)clock64();
    };
    if (tau < p.ntiles) {
        do_tile(tau, std::true_type{});
        for (tau += nblk; tau < p.ntiles; tau += nblk) do_tile(tau, std::false_type{});
    }
    if (acc_stats && stat_nb >= 0)
        flush_channel_sums<NT>(s1, s2, reinterpret_cast<float*>(sX), stat_dst, p.Cout, stat_nb * (16 * NT), t, li, lg, wave, bid);
    if (p.dbg && t == 0) p.dbg[bid * 8 + 6] = (int)clock64();
}

// One persistent launch over up to 4 problems (HRNet branches in lockstep): the register tilings the lockstep
// launches use, in one kernel.
__global__ __launch_bounds__(256, 2) void conv3x3_tile_kernel(C3Launch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char c3_smem[];
    // one tile list over all problems, workgroup b takes tiles b, b + grid, ...: measured faster than giving every
    // problem its own range of workgroups (44 vs 51 us for the four HRNet branches at B = 32)
    const int bid = blockIdx.x, nblk = gridDim.x;
    // the second workgroup of every CU (ids >= 256) starts late by about half a tile period: from then on one workgroup of a
    // CU computes while the other loads / stores, instead of both being in the same phase all the time
    if (bid >= 256) for (int i = 0; i < L.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // ... and every workgroup visits the problems in its own rotation (the two workgroups of a CU, ids 256 apart, are one
    // problem apart): 45.5 -> 44.3 us
    const int rot = (bid + bid / 256) % L.n;
    for (int ii = 0; ii < L.n; ++ii) {
        const int i = (ii + rot) % L.n;
        const C3Prob& p = L.p[i];
        switch (p.cfg) {
#define C3_CASE(M_, N_, K_) case M_ * 100 + N_ * 10 + K_: c3_body<M_, N_, K_>(p, bid, nblk, c3_smem); break;
            C3_CASE(4, 3, 1) C3_CASE(4, 3, 2) C3_CASE(4, 3, 4)
#undef C3_CASE
            default: break;
        }
    }
}
inline bool multi_has(int cfg) { return cfg == 431 || cfg == 432 || cfg == 434; }

// One problem, one register tiling per kernel (each instance gets its own register allocation).
template <int MT, int NT, int KW>
__global__ __launch_bounds__(256, 2) void conv3x3_one_kernel(C3Launch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char c3_smem[];
    if (blockIdx.x >= 256) for (int i = 0; i < L.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    c3_body<MT, NT, KW>(L.p[0], blockIdx.x, gridDim.x, c3_smem);
}

template <int MT, int NT, int KW>
void launch_one(const C3Launch& L, int grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_one_kernel<MT, NT, KW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_one_kernel<MT, NT, KW>), dim3((unsigned)grid), dim3(256), lds, st, L);
}

// run-time knobs (A-B timing, tests): defaults from the environment, settable through danet_knob (DANET_KNOB_C3_*)
struct Forced { int mt, kw; };
Forced g_force = [] {
    Forced f{0, 0};
    if (const char* s = getenv("DANET_C3_FORCE")) { if (sscanf(s, "%d,%d", &f.mt, &f.kw) != 2) f = Forced{0, 0}; }
    return f;
}();
bool g_c3_on = getenv("DANET_NO_C3") == nullptr;                  // off: everything back on conv_fast.hip
int g_c3_blocks = getenv("DANET_C3_BLOCKS") ? atoi(getenv("DANET_C3_BLOCKS")) : 512;
int g_c3_want = getenv("DANET_C3_WANT") ? atoi(getenv("DANET_C3_WANT")) : 0;     // tiles per problem the planner aims for (0: 512 / problems)
Forced forced_cfg() { return g_force; }
int* g_dbg = nullptr;
int g_c3_stagger = getenv("DANET_C3_STAGGER") ? atoi(getenv("DANET_C3_STAGGER")) : 0;   // x 8128 cycles

constexpr int LDS_TWO = 81920;        // two workgroups per CU
constexpr int LDS_ONE = 160 * 1024;

// Fills the tile plan of q for register tiling (MT, KW); returns the dynamic LDS bytes or -1 when the tiling cannot run.
int plan_one(const ConvP& p, C3Prob& q, int NT, int MT, int KW) {
    const int PW = 4 / KW, TP = PW * MT * 16;
    const int H = p.OH, W = p.OW;                  // stride 1: gathered and written tensors have the same extent
    int NI = 1, TH = H;
    if (H * W <= TP) {
        NI = 1;
        for (int n = TP / (H * W); n >= 1; --n) if (p.B % n == 0) { NI = n; break; }
    } else {
        TH = 0;
        for (int h = TP / W; h >= 1; --h) if (H % h == 0) { TH = h; break; }
        if (TH == 0) return -1;
    }
    if ((long)NI * TH * W * 4 < (long)TP * 3) return -1;             // < 75 % of the register tile in use
    const int S = p.Cin / 8;
    int Sp = S;
    while (Sp % 4 != 2) ++Sp;
    const int nks = p.Kp / 32;
    if (nks > TAB_BYTES / 8 || nks < KW * 2) return -1;
    if (W * S > NSUB * 256) return -1;
    const int nrows = NI * (TH + 2);
    if (nrows * 2 * Sp > PADN * 256) return -1;
    const long tile = (long)nrows * (W + 2) * Sp * 16;
    const long red = KW > 1 ? (long)4 * (MT - MT / KW) * NT * 1024 : 0;
    const long scr = 4 * 2 * NT * 16 * 4;
    long lds = tile > red ? tile : red;
    if (scr > lds) lds = scr;
    lds += TAB_BYTES;
    if (lds > LDS_ONE) return -1;
    q.TH = TH; q.NI = NI; q.Wp = W + 2; q.S = S; q.Sp = Sp;
    q.tiles_h = H / TH; q.npt = (p.B / NI) * q.tiles_h; q.nnb = p.Cout_pad / (16 * NT);
    q.nks = nks; q.nc16 = p.Cin / 16;
    q.ntiles = q.npt * q.nnb;
    q.cfg = MT * 100 + NT * 10 + KW;
    q.swz = (q.npt % 8 == 0 && q.nnb > 1) ? 1 : 0;
    q.has_idle = (NI * TH * W != TP || p.Cout != p.Cout_pad) ? 1 : 0;
    return (int)lds;
}

bool shape_ok(const ConvP& p, bool vec8) {
    if (!g_c3_on || !vec8) return false;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.groups != 1) return false;
    if (p.H != p.OH || p.W != p.OW || p.Cin % 16 != 0 || p.Cout % 4 != 0) return false;
    if (p.x_bytes >= (1L << 31) || p.y_bytes >= (1L << 31)) return false;
    if ((long)p.Cout_pad * p.Kp * 2 >= (1L << 31)) return false;
    if ((long)p.B * p.H * p.W >= (1L << 24)) return false;
    return true;
}

// Tiling for a problem that shares its launch with nprob - 1 others: the first candidate that gives the launch
// enough tiles for two workgroups per CU with a tile that leaves room for two workgroups' LDS; otherwise the
// candidate with the most tiles.
int plan(const ConvP& p, C3Prob& q, int nprob) {
    const int NT = danet_conv_nt(p.Cout);
    // measured on MI355X (tools/c3_bench.py, B = 32): the 64-pixel x 48-channel wave tile beats the 128-pixel one at
    // these sizes (more, shorter workgroups hide the serial load -> compute -> store chain of a tile better)
    static const int cand[6][2] = {{4, 1}, {4, 2}, {4, 4}, {8, 1}, {8, 2}, {8, 4}};
    const Forced f = forced_cfg();
    const int want = g_c3_want > 0 ? g_c3_want : (2 * 256 + nprob - 1) / nprob;
    int best = -1, best_tiles = -1, best_lds = -1;
    C3Prob tmp = q;
    for (int c = 0; c < 6; ++c) {
        const int MT = cand[c][0], KW = cand[c][1];
        if (MT == 8 && NT != 3) continue;
        if (nprob > 1 && !multi_has(MT * 100 + NT * 10 + KW)) continue;      // tilings compiled into the multi-problem kernel
        if (f.mt && (MT != f.mt || KW != f.kw)) continue;
        C3Prob t = q;
        const int lds = plan_one(p, t, NT, MT, KW);
        if (lds < 0) continue;
        if (f.mt) { q = t; return lds; }
        const bool two = lds <= LDS_TWO;
        if (two && t.ntiles >= want) { q = t; return lds; }
        // keep the best fallback: prefer two-per-CU tiles, then more tiles, then the earlier tiling
        const int score = (two ? 1 << 24 : 0) + t.ntiles;
        if (score > best_tiles) { best_tiles = score; best = c; best_lds = lds; tmp = t; }
    }
    if (best < 0) return -1;
    q = tmp;
    return best_lds;
}

}  // namespace

namespace danet_conv {

int* conv3x3_debug_buffer() { return g_dbg; }

bool conv3x3_ok(const ConvP& p, bool vec8) {
    if (!shape_ok(p, vec8)) return false;
    C3Prob q{};
    return plan(p, q, 1) > 0;
}

// MT*100 + NT*10 + KW of the tiling a problem gets in a launch of nprob problems (0: not supported).
int conv3x3_config(const ConvP& p, bool vec8, int nprob) {
    if (!shape_ok(p, vec8)) return 0;
    C3Prob q{};
    return plan(p, q, nprob) > 0 ? q.cfg : 0;
}

// Launches n (<= 4) problems in one launch.  0 on launch, -1 when the set cannot run on this kernel (nothing is
// launched then); dry = true only answers that question.
bool conv3x3_stream_first() { return g_c3_on && g_force.mt == 0; }      // conv3x3_launch offers the problem set to the streamed kernel first

int conv3x3_launch(const ConvP* ps, int n, void* stream, bool dry) {
    if (n < 1 || n > C3_MAXP) return -1;
    if (g_c3_on && g_force.mt == 0 && conv3x3s_launch(ps, n, stream, dry) == 0) return 0;      // the streamed kernel takes what it can (no forced tiling)
    C3Launch L{};
    L.n = n;
    int lds_max = 0, tile0 = 0;
    for (int i = 0; i < n; ++i) {
        const ConvP& p = ps[i];
        C3Prob& q = L.p[i];
        q.x = p.x; q.w = p.w; q.y = p.y; q.bias = p.bias; q.stats = p.stats;
        q.bn_x = p.bn_x; q.bn_y = p.bn_y; q.bn_saved = p.bn_saved; q.bn_red = p.bn_red; q.bn_gate = p.bn_gate;
        q.addend = p.addend;
        q.B = p.B; q.H = p.OH; q.W = p.OW; q.Cin = p.Cin; q.Cout = p.Cout; q.Cout_pad = p.Cout_pad;
        q.flip = p.transposed; q.relu = p.relu; q.out_fp32 = p.out_fp32;
        q.x_bytes = (int)p.x_bytes; q.y_bytes = (int)p.y_bytes;
        q.dbg = g_dbg;
        const int lds = plan(p, q, n);
        if (lds < 0) return -1;
        q.tile0 = tile0;
        tile0 += q.ntiles;
        if (lds > lds_max) lds_max = lds;
    }
    L.total = tile0;
    L.stagger = g_c3_stagger;
    if (n > 1) for (int i = 0; i < n; ++i) if (!multi_has(L.p[i].cfg)) return -1;
    if (dry) {
        if (n == 1) { const int c = L.p[0].cfg, mt = c / 100, nt = (c / 10) % 10, kw = c % 10; return ((mt == 4 && nt >= 1 && nt <= 4) || (mt == 8 && nt == 3)) && (kw == 1 || kw == 2 || kw == 4) ? 0 : -1; }
        return 0;
    }
    const int cap = lds_max <= LDS_TWO ? g_c3_blocks : 256;
    const int grid = L.total < cap ? L.total : cap;
    hipStream_t st = (hipStream_t)stream;
    if (n == 1) {
        switch (L.p[0].cfg) {
#define C3_CASE(M_, N_, K_) case M_ * 100 + N_ * 10 + K_: launch_one<M_, N_, K_>(L, grid, (size_t)lds_max, st); return 0;
            C3_CASE(8, 3, 1) C3_CASE(8, 3, 2) C3_CASE(8, 3, 4)
            C3_CASE(4, 3, 1) C3_CASE(4, 3, 2) C3_CASE(4, 3, 4)
            C3_CASE(4, 4, 1) C3_CASE(4, 4, 2) C3_CASE(4, 4, 4)
            C3_CASE(4, 2, 1) C3_CASE(4, 2, 2) C3_CASE(4, 2, 4)
            C3_CASE(4, 1, 1) C3_CASE(4, 1, 2) C3_CASE(4, 1, 4)
#undef C3_CASE
            default: return -1;
        }
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ONE);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_tile_kernel, dim3((unsigned)grid), dim3(256), (size_t)lds_max, st, L);
    return 0;
}

}  // namespace danet_conv

// danet_knob ids of this file: enable 0/1; forced register tiling (mt, kw) of every problem (0 = planner's choice); workgroup cap
// of a launch (0 is not a cap: ignored); tiles per problem the planner aims for (0 = 512 / problems of the launch).
long danet_conv::conv3x3_knob(int id, long v) {
    long prev = 0;
    switch (id) {
        case DANET_KNOB_C3_ENABLE: prev = g_c3_on ? 1 : 0; if (v >= 0) g_c3_on = v != 0; break;
        case DANET_KNOB_C3_MT: prev = g_force.mt; if (v >= 0) g_force.mt = (int)v; break;
        case DANET_KNOB_C3_KW: prev = g_force.kw; if (v >= 0) g_force.kw = (int)v; break;
        case DANET_KNOB_C3_BLOCKS: prev = g_c3_blocks; if (v > 0) g_c3_blocks = (int)v; break;
        case DANET_KNOB_C3_WANT: prev = g_c3_want; if (v >= 0) g_c3_want = (int)v; break;
        default: break;
    }
    return prev;
}
// Profiling hook: device buffer of blocks*8 ints that receives every workgroup's phase timestamps (NULL: off).
extern "C" void danet_conv3x3_debug(int* dev_buf) { g_dbg = dev_buf; }

