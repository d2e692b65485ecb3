// Weight gradient of pointwise (1x1, stride 1) convolutions:  dW[cout][cin] = sum over pixels of dY[pixel][cout] * X[pixel][cin]
// -- the Bottleneck projections 64 <-> 256 (/root/reference/models/module/res_module.py:64-97), the limb regressor's 24 -> 64
// input projection over the 768 part crops, the narrow head bottlenecks.  The generic kernel (conv_wgrad.hip) transposes every
// 32-pixel chunk with v_perm while staging and adds its partial sums with float atomics: 27 TFLOP/s = 1.5 TB/s on the
// 24 -> 64 layer, whose whole cost is reading dY (403 MB) and X (151 MB) once.  Here
//   * a chunk of 32 pixels of dY and of X is copied into LDS AS IT LIES in memory (rows of an NHWC tensor are contiguous over
//     the pixel index: one fully coalesced block per chunk, 16 bytes per lane), double-buffered, the next chunk's loads in flight
//     behind the current chunk's MFMAs;
//   * MFMA fragments with the PIXEL index on the reduction axis come out of ds_read_b64_tr_b16 (the LDS transpose read, lane
//     mapping as in conv_wgrad3x3.hip); pixel rows are padded by 32 bytes so that the four pixels a 16-lane group touches fall
//     on different banks;
//   * the (cout block) x (cin block) accumulator tiles of the layer are split over the four waves (cout blocks first, then cin
//     blocks) and stay in registers over the workgroup's whole pixel range;
//   * partial sums leave with plain coalesced stores into [split][cout][cin] and are summed in a fixed order by a second kernel
//     that also applies beta: deterministic, no atomics (for a 1x1 layer the packed layout IS the torch layout).
// Several problems share a launch (the trainer queues weight gradients and flushes them in multi-problem launches).
#include "common.h"
#include "conv_common.h"

namespace {

using namespace danet_conv;

typedef __attribute__((ext_vector_type(2))) unsigned v2u;
constexpr int OOB = 0x7fffffff;
constexpr int PWG_KS = 32;                    // pixels per MFMA k-step; a chunk = NKC k-steps (narrow layers: 2, so that a barrier round moves >= ~10 KB)
constexpr int PWG_NPM = 16;                   // problems per launch
constexpr int PWG_MAXPIECES = 4;              // 16-byte pieces per thread and tensor: 32 pixels x <= 256 channels

__device__ inline v2u tr_read(unsigned lds_byte_addr) {
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr) : "memory");
    return r;
}

struct PwgP {
    const bf16_t* x; const bf16_t* dy; float* part;
    int M, K, N;                              // pixels, input channels, output channels
    int KB, NB;                               // sixteen-channel blocks
    int nsplit, ksplit;                       // waves along cout blocks x waves along cin blocks (= 4)
    int msplit, nchunks;
    int x_bytes, dy_bytes;
};
struct PwgMulti { PwgP p[PWG_NPM]; int start[PWG_NPM + 1]; int n; };

template <int NBW, int KBW, int NKC>
__global__ __launch_bounds__(256) void conv_pw_wgrad_kernel(PwgMulti mp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pi = 0;
    while (pi + 1 < mp.n && (int)blockIdx.x >= mp.start[pi + 1]) ++pi;
    const PwgP& p = mp.p[pi];
    const int bx = blockIdx.x - mp.start[pi];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    constexpr int PWG_CH = PWG_KS * NKC;
    const int N = p.N, K = p.K;
    const int PXY = p.NB * 32 + 32, PXX = p.KB * 32 + 32;         // pixel rows: whole sixteen-channel blocks + 32 bytes of padding (see the header)
    const int YB = PWG_CH * PXY, BUF = YB + PWG_CH * PXX;
    const int wn = wave % p.nsplit, wk = wave / p.nsplit;

    f32x4 acc[NBW][KBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < KBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: lane i of a 16-lane group points at pixel 8 lg + 4 h + (i >> 2), channels c0 + 4 (i & 3)
    unsigned aoff[NBW], boff[KBW];
    bool aon[NBW], bon[KBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = wn + i * p.nsplit;
        aon[i] = nb < p.NB;
        aoff[i] = (unsigned)((lg * 8 + (li >> 2)) * PXY + ((aon[i] ? nb : 0) * 16 + 4 * (li & 3)) * 2);
    }
#pragma unroll
    for (int j = 0; j < KBW; ++j) {
        const int kb = wk + j * p.ksplit;
        bon[j] = kb < p.KB;
        boff[j] = (unsigned)((lg * 8 + (li >> 2)) * PXX + ((bon[j] ? kb : 0) * 16 + 4 * (li & 3)) * 2);
    }

    const int per = (p.nchunks + p.msplit - 1) / p.msplit;
    const int c_begin = bx * per, c_end = min(p.nchunks, c_begin + per);
    // staging: piece pc of a tensor's chunk = (pixel q = pc / C8, 16-byte channel chunk c8 = pc % C8); blocks of sixteen channels are
    // staged whole (channels beyond the tensor's count as zeros), so C8 = 2 * (blocks)
    const int N8 = p.NB * 2, K8 = p.KB * 2;
    const int NPY = PWG_CH * N8, NPX = PWG_CH * K8;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    int yrel[PWG_MAXPIECES], xrel[PWG_MAXPIECES], ylds[PWG_MAXPIECES], xlds[PWG_MAXPIECES], yq[PWG_MAXPIECES], xq[PWG_MAXPIECES];
#pragma unroll
    for (int u = 0; u < PWG_MAXPIECES; ++u) {
        const int pc = t + u * 256;
        {
            const int q = pc / N8, c8 = pc - q * N8;
            const bool live = pc < NPY && c8 * 8 < N;
            yrel[u] = live ? (q * N + c8 * 8) * 2 : OOB;
            ylds[u] = pc < NPY ? q * PXY + c8 * 16 : -1;
            yq[u] = q;
        }
        {
            const int q = pc / K8, c8 = pc - q * K8;
            const bool live = pc < NPX && c8 * 8 < K;
            xrel[u] = live ? (q * K + c8 * 8) * 2 : OOB;
            xlds[u] = pc < NPX ? q * PXX + c8 * 16 : -1;
            xq[u] = q;
        }
    }
    uint4 ystage[PWG_MAXPIECES], xstage[PWG_MAXPIECES];
    auto fetch = [&](int ch) {
        const int m0 = ch * PWG_CH;
        const int yb = m0 * N * 2, xb = m0 * K * 2;
#pragma unroll
        for (int u = 0; u < PWG_MAXPIECES; ++u) {
            ystage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yr, (yrel[u] != OOB && m0 + yq[u] < p.M) ? yb + yrel[u] : OOB, 0, 0));
            xstage[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, (xrel[u] != OOB && m0 + xq[u] < p.M) ? xb + xrel[u] : OOB, 0, 0));
        }
    };
    auto commit = [&](int buf) {
        unsigned char* const base = smem + buf * BUF;
#pragma unroll
        for (int u = 0; u < PWG_MAXPIECES; ++u) {
            if (ylds[u] >= 0) *reinterpret_cast<uint4*>(base + ylds[u]) = ystage[u];
            if (xlds[u] >= 0) *reinterpret_cast<uint4*>(base + YB + xlds[u]) = xstage[u];
        }
    };
    if (c_begin < c_end) fetch(c_begin);
    int buf = 0;
    for (int ch = c_begin; ch < c_end; ++ch) {
        commit(buf);
        __syncthreads();                       // chunk `buf` complete; the reads of `buf ^ 1` (previous chunk) are finished as well
        if (ch + 1 < c_end) fetch(ch + 1);
        const unsigned ybase = (unsigned)(buf * BUF), xbase = ybase + (unsigned)YB;
#pragma unroll
        for (int ks = 0; ks < NKC; ++ks) {
            const unsigned yk = ybase + (unsigned)(ks * PWG_KS * PXY), xk = xbase + (unsigned)(ks * PWG_KS * PXX);
            v2u alo[NBW], ahi[NBW], blo[KBW], bhi[KBW];
#pragma unroll
            for (int i = 0; i < NBW; ++i) { alo[i] = tr_read(yk + aoff[i]); ahi[i] = tr_read(yk + aoff[i] + 4 * PXY); }
#pragma unroll
            for (int j = 0; j < KBW; ++j) { blo[j] = tr_read(xk + boff[j]); bhi[j] = tr_read(xk + boff[j] + 4 * PXX); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the compiler does not count LDS operations issued from inline asm)
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 a[NBW], b[KBW];
#pragma unroll
            for (int i = 0; i < NBW; ++i) { const uint4 raw = {alo[i].x, alo[i].y, ahi[i].x, ahi[i].y}; a[i] = __builtin_bit_cast(bf16x8, raw); }
#pragma unroll
            for (int j = 0; j < KBW; ++j) { const uint4 raw = {blo[j].x, blo[j].y, bhi[j].x, bhi[j].y}; b[j] = __builtin_bit_cast(bf16x8, raw); }
#pragma unroll
            for (int i = 0; i < NBW; ++i)
#pragma unroll
                for (int j = 0; j < KBW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }
    // partial dW of this workgroup: part[bx][cout][cin]; the lane holds rows (couts) 4 lg + r of column (cin) li of every tile
    float* const dst = p.part + (size_t)bx * N * K;
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < KBW; ++j) {
            const int nb = wn + i * p.nsplit, kb = wk + j * p.ksplit;
            const int cin = kb * 16 + li;
            if (!aon[i] || !bon[j] || cin >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = nb * 16 + lg * 4 + r;
                if (cout < N) dst[(size_t)cout * K + cin] = acc[i][j][r];
            }
        }
}

struct PwgRed { const float* part[PWG_NPM]; float* dw[PWG_NPM]; int msplit[PWG_NPM]; long start[PWG_NPM + 1]; int n; float beta; };

// dW = beta * dW + sum over the splits, in a fixed order: eight lanes per output element (lane s sums splits s, s + 8, ..., then a
// three-step butterfly): a 768-element layer with hundreds of splits is otherwise three workgroups of serial, dependent loads
__global__ __launch_bounds__(256) void conv_pw_wgrad_reduce_kernel(PwgRed rp)
{
    const long gidx = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const bool live = gidx < rp.start[rp.n];
    const long gi = live ? gidx : rp.start[rp.n] - 1;
    int i = 0;
    while (i + 1 < rp.n && gi >= rp.start[i + 1]) ++i;
    const long idx = gi - rp.start[i], total = rp.start[i + 1] - rp.start[i];
    const float* part = rp.part[i];
    const int ms = rp.msplit[i];
    float s = 0.f;
    for (int sp = sub; sp < ms; sp += 32) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = sp + 8 * u < ms ? part[(size_t)(sp + 8 * u) * total + idx] : 0.f;
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (live && sub == 0) {
        float* dw = rp.dw[i];
        dw[idx] = rp.beta != 0.f ? dw[idx] * rp.beta + s : s;
    }
}

bool g_pwg_on = getenv("DANET_NO_PW_WGRAD") == nullptr;           // A-B timing knob (danet_conv_pw_wgrad_set)

struct PwgPlan { int NB, KB, nsplit, ksplit, nbw, kbw, nbw_t, kbw_t, nkc; };

bool pwg_plan(int N, int K, PwgPlan& pl) {
    pl.NB = (N + 15) / 16; pl.KB = (K + 15) / 16;
    if (pl.NB > 16 || pl.KB > 16) return false;
    pl.nsplit = pl.NB % 4 == 0 ? 4 : (pl.NB % 2 == 0 ? 2 : 1);
    pl.ksplit = 4 / pl.nsplit;
    pl.nbw = (pl.NB + pl.nsplit - 1) / pl.nsplit;
    pl.kbw = (pl.KB + pl.ksplit - 1) / pl.ksplit;
    pl.nbw_t = pl.nbw <= 1 ? 1 : (pl.nbw <= 2 ? 2 : 4);
    pl.kbw_t = pl.kbw <= 1 ? 1 : (pl.kbw <= 2 ? 2 : (pl.kbw <= 4 ? 4 : (pl.kbw <= 8 ? 8 : 16)));
    if (pl.nbw > 4 || pl.nbw_t * pl.kbw_t > 16) return false;
    // pixels per barrier round: 64 for layers of <= 8 sixteen-channel blocks in all (<= 288 bytes per staged pixel: <= 36 KB of LDS for
    // both buffers, and <= 4 staging pieces per thread and tensor), else 32
    pl.nkc = (pl.NB + pl.KB <= 8 && pl.NB <= 8 && pl.KB <= 8) ? 2 : 1;
    return true;
}

size_t pwg_lds(int N, int K, int nkc) { return (size_t)2 * PWG_KS * nkc * ((size_t)((N + 15) / 16 * 16) * 2 + 32 + (size_t)((K + 15) / 16 * 16) * 2 + 32); }

}  // namespace

namespace danet_conv {

// 1x1 / stride 1 / no padding / one group, channel counts in 8-channel granules and <= 256
bool conv_pw_wgrad_ok(const WgJob& j) {
    if (!g_pwg_on) return false;
    if (j.R != 1 || j.S != 1 || j.stride != 1 || j.pad != 0 || j.dil != 1 || j.groups != 1) return false;
    if (j.H != j.OH || j.W != j.OW || j.Cin % 8 != 0 || j.Cout % 8 != 0) return false;
    const long M = (long)j.B * j.OH * j.OW;
    if (M < 4096 || M * j.Cin * 2 >= (1L << 31) || M * j.Cout * 2 >= (1L << 31)) return false;
    PwgPlan pl;
    return pwg_plan(j.Cout, j.Cin, pl);
}

// Splits (workgroups) of a problem when `nprob` problems of `total_px` pixels in all share `target` workgroups.
static int pwg_msplit(const WgJob& j, long total_px, long target) {
    const long M = (long)j.B * j.OH * j.OW;
    PwgPlan pl;
    (void)pwg_plan(j.Cout, j.Cin, pl);
    const long nchunks = (M + PWG_KS * pl.nkc - 1) / (PWG_KS * pl.nkc);
    long ms = (long)((double)target * (double)M / (double)total_px + 0.5);
    if (ms > nchunks / 8) ms = nchunks / 8;                    // at least eight barrier rounds per workgroup (prologue, partial-sum traffic)
    if (ms < 1) ms = 1;
    return (int)ms;
}

// The problems idx[0..cnt) go out in launches of one kernel instance each (greedy, in order, at most PWG_NPM per launch): gpx[k] = the
// pixels of all problems that share problem k's launch.  `target` workgroups are dealt over the problems of ONE LAUNCH (round 5; rounds
// 3-4 dealt them over all problems of the call, so that each of its ~12 launches got a twelfth of the workgroups: 64 on 256 CUs).
static void pwg_group_pixels(const WgJob* jobs, const int* idx, int cnt, long* gpx) {
    static const bool per_call = getenv("DANET_PWG_PER_CALL") != nullptr;          // A-B timing knob: the rounds 3-4 deal
    if (per_call) {
        long total = 0;
        for (int k = 0; k < cnt; ++k) total += (long)jobs[idx[k]].B * jobs[idx[k]].OH * jobs[idx[k]].OW;
        for (int k = 0; k < cnt && k < 4096; ++k) gpx[k] = total;
        return;
    }
    bool done[4096];
    for (int k = 0; k < cnt && k < 4096; ++k) done[k] = false;
    for (int k0 = 0; k0 < cnt && k0 < 4096; ++k0) {
        if (done[k0]) continue;
        PwgPlan pl0;
        if (!pwg_plan(jobs[idx[k0]].Cout, jobs[idx[k0]].Cin, pl0)) { gpx[k0] = (long)jobs[idx[k0]].B * jobs[idx[k0]].OH * jobs[idx[k0]].OW; done[k0] = true; continue; }
        int members[64], nm = 0;
        long px = 0;
        for (int k = k0; k < cnt && k < 4096 && nm < PWG_NPM; ++k) {
            if (done[k]) continue;
            PwgPlan pl;
            if (!pwg_plan(jobs[idx[k]].Cout, jobs[idx[k]].Cin, pl) || pl.nbw_t != pl0.nbw_t || pl.kbw_t != pl0.kbw_t || pl.nkc != pl0.nkc) continue;
            done[k] = true;
            members[nm++] = k;
            px += (long)jobs[idx[k]].B * jobs[idx[k]].OH * jobs[idx[k]].OW;
        }
        for (int m = 0; m < nm; ++m) gpx[members[m]] = px;
    }
}

// Workspace floats the problems idx[0..cnt) need (partial sums [msplit][Cout][Cin] each).
size_t conv_pw_wgrad_ws_floats(const WgJob* jobs, const int* idx, int cnt, long target) {
    thread_local static long gpx_buf[4096];      // (per host thread: the library holds no shared mutable scratch)
    if (cnt > 4096) return 0;
    long* const gpx = gpx_buf;
    pwg_group_pixels(jobs, idx, cnt, gpx);
    long total_px = 0;
    for (int k = 0; k < cnt; ++k) total_px += (long)jobs[idx[k]].B * jobs[idx[k]].OH * jobs[idx[k]].OW;
    size_t need = 0;
    for (int k = 0; k < cnt; ++k) {
        const WgJob& j = jobs[idx[k]];
        need += ((size_t)pwg_msplit(j, gpx[k], target) * j.Cout * j.Cin + 15) / 16 * 16;
    }
    return need;
}

// Launches the problems idx[0..cnt) (all conv_pw_wgrad_ok) in groups that share an instantiation; ws: scratch of
// conv_pw_wgrad_ws_floats floats (need not be zeroed).  0 on success.
int conv_pw_wgrad_launch(const WgJob* jobs, const int* idx, int cnt, float* ws, float beta, long target, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    long total_px = 0;
    for (int k = 0; k < cnt; ++k) total_px += (long)jobs[idx[k]].B * jobs[idx[k]].OH * jobs[idx[k]].OW;
    bool done[4096];
    thread_local static long gpx_launch[4096];
    if (cnt > 4096) return -1;
    for (int k = 0; k < cnt; ++k) done[k] = false;
    pwg_group_pixels(jobs, idx, cnt, gpx_launch);
    size_t used = 0;
    for (int k0 = 0; k0 < cnt; ++k0) {
        if (done[k0]) continue;
        PwgPlan pl0;
        if (!pwg_plan(jobs[idx[k0]].Cout, jobs[idx[k0]].Cin, pl0)) return -1;
        PwgMulti mp; PwgRed rp;
        mp.n = 0; mp.start[0] = 0; rp.n = 0; rp.start[0] = 0; rp.beta = beta;
        size_t lds = 0;
        for (int k = k0; k < cnt && mp.n < PWG_NPM; ++k) {
            if (done[k]) continue;
            const WgJob& j = jobs[idx[k]];
            PwgPlan pl;
            if (!pwg_plan(j.Cout, j.Cin, pl) || pl.nbw_t != pl0.nbw_t || pl.kbw_t != pl0.kbw_t || pl.nkc != pl0.nkc) continue;
            done[k] = true;
            PwgP& p = mp.p[mp.n];
            const long M = (long)j.B * j.OH * j.OW;
            p.x = (const bf16_t*)j.x; p.dy = (const bf16_t*)j.dy; p.part = ws + used;
            p.M = (int)M; p.K = j.Cin; p.N = j.Cout; p.KB = pl.KB; p.NB = pl.NB; p.nsplit = pl.nsplit; p.ksplit = pl.ksplit;
            p.msplit = pwg_msplit(j, gpx_launch[k], target); p.nchunks = (int)((M + PWG_KS * pl.nkc - 1) / (PWG_KS * pl.nkc));
            p.x_bytes = (int)(M * j.Cin * 2); p.dy_bytes = (int)(M * j.Cout * 2);
            mp.start[mp.n + 1] = mp.start[mp.n] + p.msplit;
            rp.part[rp.n] = p.part; rp.dw[rp.n] = j.dw; rp.msplit[rp.n] = p.msplit;
            rp.start[rp.n + 1] = rp.start[rp.n] + (long)j.Cout * j.Cin;
            used += ((size_t)p.msplit * j.Cout * j.Cin + 15) / 16 * 16;
            const size_t l = pwg_lds(j.Cout, j.Cin, pl.nkc);
            if (l > lds) lds = l;
            ++mp.n; ++rp.n;
        }
        const dim3 grid((unsigned)mp.start[mp.n]);
#define PWG_CASE(A_, B_, C_) if (pl0.nbw_t == A_ && pl0.kbw_t == B_ && pl0.nkc == C_) { \
            static bool attr_set = false; \
            if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_wgrad_kernel<A_, B_, C_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_set = true; } \
            hipLaunchKernelGGL((conv_pw_wgrad_kernel<A_, B_, C_>), grid, dim3(256), lds, st, mp); } else
        PWG_CASE(1, 1, 2) PWG_CASE(1, 2, 2) PWG_CASE(1, 4, 2) PWG_CASE(1, 8, 2)
        PWG_CASE(2, 1, 2) PWG_CASE(2, 2, 2) PWG_CASE(2, 4, 2)
        PWG_CASE(4, 1, 2) PWG_CASE(4, 2, 2)
        PWG_CASE(1, 1, 1) PWG_CASE(1, 2, 1) PWG_CASE(1, 4, 1) PWG_CASE(1, 8, 1) PWG_CASE(1, 16, 1)
        PWG_CASE(2, 1, 1) PWG_CASE(2, 2, 1) PWG_CASE(2, 4, 1) PWG_CASE(2, 8, 1) PWG_CASE(4, 1, 1) PWG_CASE(4, 2, 1) PWG_CASE(4, 4, 1)
        return -1;
#undef PWG_CASE
        if (hipGetLastError() != hipSuccess) return -2;
        hipLaunchKernelGGL(conv_pw_wgrad_reduce_kernel, dim3((unsigned)danet::cdiv(rp.start[rp.n] * 8, 256)), dim3(256), 0, st, rp);
        if (hipGetLastError() != hipSuccess) return -2;
    }
    return 0;
}

}  // namespace danet_conv

// Run-time switch of the pointwise weight-gradient kernel (A-B timing, tests): enable 0 / 1 (-1 keeps); returns the previous setting.
long danet_conv::conv_pw_wgrad_knob(long enable) {
    const long prev = g_pwg_on ? 1 : 0;
    if (enable >= 0) g_pwg_on = enable != 0;
    return prev;
}
