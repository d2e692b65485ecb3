// The regressor's graph tail in ONE launch per direction (/root/reference/models/danet/smpl_regressor.py:846-900, GCN.py:12-92,
// utils/geometry.py:47-61): from the 24 limb features [B, 24, 128] to the four things the losses and the SMPL layer consume --
//   joint_rotation[0] = rot6d(head(pose_regressors[0], rot_feats) + mean_pose)
//   pos_init  = relu(BN24(r2p_A   (x) rot_feats (x) W0 + b0))            joint_position[0] = head(coord_regressors[0], pos_init)
//   h1..h3    = relu(BN24(norm_A  (x) h         (x) Wl + bl))  (128 -> 256 -> 256 -> 128),  norm_A = D^-1/2 (I + A_mask relu(edge)) D^-1/2
//   pos_ref   = pos_init + h3                                            joint_position[1] = head(coord_regressors[1], pos_ref)
//   rot_ref   = relu(BN24(p2r_A   (x) pos_ref   (x) W4 + b4))            smpl_pose = rot6d(head(pose_regressors[1], rot_ref) + mean_pose)
// As torch operations this was ~80 launches forward and ~125 backward of a few workgroups each (linear layers through the BLAS library,
// BatchNorm1d through MIOpen, element-wise glue), 0.39 + 0.65 ms of a stretch of the step where nothing else runs (round 6,
// tools/util_timeline.sh).  Here ONE workgroup owns ONE joint for the whole pass: BatchNorm1d(24) normalises per joint over (batch,
// channel), so its statistics are local; the grouped 1x1 heads are per joint; what crosses joints is only the adjacency mix, and
// there a grid-wide barrier (grid_barrier.h, the fenced form: plain stores cross it) separates the layers -- 4 barriers forward, 5
// backward, 24 workgroups.  Arithmetic: fp32 throughout (as the torch path), FMA on the vector units; the [32 x K] x [K x N] products
// keep the 32 batch rows of the joint in LDS (broadcast reads) and stream the weights from L2 (each workgroup reads every weight once).
// Weight gradients: per-joint partial products, summed over the joints in a fixed order after the last barrier (bit-reproducible).
#include "common.h"
#include "grid_barrier.h"

namespace {

constexpr int NJ = 24, BM = 32, NT = 1024, NLAY = 5, PADF = 4, MAXC = 256;      // 16 waves per workgroup: the loops are latency-bound, four waves per SIMD hide it
__host__ __device__ constexpr int lay_cin(int l) { return l == 2 || l == 3 ? 256 : 128; }
__host__ __device__ constexpr int lay_cout(int l) { return l == 1 || l == 2 ? 256 : 128; }

// (struct danet_gcn_tail_args of include/danet_hip.h; _lib.GcnTailArgs on the host side)
typedef danet_gcn_layer Layer;
typedef danet_gcn_tail_args TailArgs;

// the forward's workspace (kept for the backward): activations before / after BatchNorm + ReLU, the mixed inputs, the statistics, the 6-D poses, W^T
struct WsMap { size_t act[NLAY], ypre[NLAY], ax[NLAY], posref, stats, pose6[2], WT[NLAY], total; };
__host__ __device__ inline WsMap ws_map(int B) {
    WsMap m; size_t o = 0; const size_t R = (size_t)B * NJ;
    for (int l = 0; l < NLAY; ++l) { m.act[l] = o; o += R * lay_cout(l); m.ypre[l] = o; o += R * lay_cout(l); m.ax[l] = o; o += R * lay_cin(l); }
    m.posref = o; o += R * 128;
    m.stats = o; o += NLAY * NJ * 2;
    m.pose6[0] = o; o += R * 6; m.pose6[1] = o; o += R * 6;
    o = (o + 3) & ~(size_t)3;
    for (int l = 0; l < NLAY; ++l) { m.WT[l] = o; o += (size_t)lay_cin(l) * lay_cout(l); }
    m.total = o;
    return m;
}
// the backward's scratch: d(ax) per layer, per-joint partial weight / bias gradients, the rows of d norm_A
struct ScMap { size_t dax[NLAY], partW[NLAY], partb[NLAY], dA, total; };
__host__ __device__ inline ScMap sc_map(int B) {
    ScMap m; size_t o = 0; const size_t R = (size_t)B * NJ;
    for (int l = 0; l < NLAY; ++l) { m.dax[l] = o; o += R * lay_cin(l); }
    for (int l = 0; l < NLAY; ++l) { m.partW[l] = o; o += (size_t)NJ * lay_cin(l) * lay_cout(l); m.partb[l] = o; o += (size_t)NJ * lay_cout(l); }
    m.dA = o; o += NJ * NJ;
    m.total = o;
    return m;
}

__device__ long long g_tail_dbg[32];          // time stamps of workgroup 0 (tools: danet_gcn_tail_debug): forward 0 .. 15, backward 16 .. 31
#define TAIL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tail_dbg[i] = wall_clock64(); } while (0)

struct Small {                      // small LDS state of a workgroup
    float An[NJ * NJ];              // norm_A
    float red[NT / 64];
    float val[NJ + 4]; int idx[NJ + 4]; int nnz;      // (padded to a multiple of four with zero weights)
    float h[BM * 8];                // head outputs / their gradients [32][8]
    float dA[NJ];                   // this joint's row of d norm_A, summed over the three refinement layers
    float colsum[NT];               // bias-gradient partials [row group][channel]
};

__device__ inline float block_sum(float v, Small& s) {
    const int t = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((t & 63) == 0) s.red[t >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += s.red[i];
    return r;
}

// the non-zeros of a row (stride 1) or a column (stride 24) of an adjacency into s.idx / s.val, in index order: 24 lanes of the first wave,
// one load each (thread 0 walking the line alone paid 24 dependent round trips when the matrix is in global memory)
__device__ inline void sparse_line(const float* A, int stride, Small& s) {
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 64) {
        const float v = t < NJ ? A[t * stride] : 0.f;
        const unsigned long long m = __ballot(v != 0.f);
        const int pos = __popcll(m & ((1ull << t) - 1ull)), c = __popcll(m);
        if (v != 0.f) { s.idx[pos] = t; s.val[pos] = v; }
        if (t == 0) s.nnz = c;
        if (t >= c && t < ((c + 3) & ~3)) { s.idx[t] = 0; s.val[t] = 0.f; }        // padding: weight 0 on joint 0
    }
    __syncthreads();
}

// dst[b][k] (LDS, row stride C + PADF) (+)= sum_i val_i * src[(b * 24 + idx_i) * C + k]; rows >= B are zero.  Four channels per lane and
// four neighbours' loads in flight at a time (the loop is latency-bound: one load per trip cost ~1 us per trip)
template <int C, bool ACCUM>
__device__ inline void mix_rows(const float* src, int B, const Small& s, float* dst) {
    const int nnz4 = (s.nnz + 3) & ~3;
    constexpr int C4 = C / 4;
#pragma unroll 2
    for (int i = threadIdx.x; i < BM * C4; i += NT) {
        const int b = i / C4, k = (i % C4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) {
            const float* base = src + (size_t)b * NJ * C + k;
            for (int j = 0; j < nnz4; j += 4) {
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const float4*>(base + (size_t)s.idx[j + u] * C);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = s.val[j + u];
                    v.x = fmaf(a, q[u].x, v.x); v.y = fmaf(a, q[u].y, v.y); v.z = fmaf(a, q[u].z, v.z); v.w = fmaf(a, q[u].w, v.w);
                }
            }
        }
        float4* d = reinterpret_cast<float4*>(dst + b * (C + PADF) + k);
        if (ACCUM) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *d = v;
    }
}

typedef __attribute__((ext_vector_type(16))) float f32x16;

// sO[32][LD + PADF] = sA[32][K + PADF] x Wg[K][LD] on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an fmaf chain,
// bit for bit) -- with the FMA on the vector units the 32 rows of sA had to be re-read from LDS by every wave for every column (broadcast
// reads, 2.3x the LDS bandwidth the arithmetic could use: the 256 x 256 layer took 30 us where the FMAs need 14).  Wave w owns the
// 32-column tile w % (LD / 32) and one of 16 / (LD / 32) slices of K; the slices are added in a fixed order at the end.  Every element
// of Wg is the B operand of exactly ONE MFMA of ONE wave: the lanes fetch their operands straight from global memory, all of a wave's
// K / KG / 2 loads issued before the first MFMA (one round trip for the whole product; staged through LDS in chunks, each chunk's round
// trip was exposed: 16 chunks x ~1 us in the backward).  sO must not alias sA.
template <int K, int LD>
__device__ __forceinline__ void gemm_mfma(const float* sA, const float* __restrict__ Wg, float* sO) {
    constexpr int NCT = LD / 32, KG = (NT / 64) / NCT, KS = K / KG, NM = KS / 2;
    static_assert(NM >= 1 && NM <= 64 && KS % 2 == 0, "slicing");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, ct = wave % NCT, kg = wave / NCT, li = lane & 31, lk = lane >> 5;
    const float* wp = Wg + (size_t)(kg * KS + lk) * LD + ct * 32 + li;
    float bv[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) bv[m] = wp[(size_t)m * 2 * LD];
    __builtin_amdgcn_sched_barrier(0);          // (left alone the scheduler sinks each load next to its MFMA: one round trip per MFMA)
    const float* ap = sA + li * (K + PADF) + kg * KS + lk;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * m], bv[m], acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        if (kg == g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* o = sO + ((r & 3) + 8 * (r >> 2) + 4 * lk) * (LD + PADF) + ct * 32 + li;
                *o = g == 0 ? acc[r] : *o + acc[r];
            }
        }
        __syncthreads();
    }
}

// pW[CIN][COUT] (global) = sX^T sG: sX[32][CIN + PADF], sG[32][COUT + PADF]; a 32 x 32 tile per wave and trip, 16 MFMAs each
template <int CIN, int COUT>
__device__ __forceinline__ void wgrad_mfma(const float* sX, const float* sG, float* __restrict__ pW) {
    constexpr int NCT = COUT / 32, TILES = (CIN / 32) * NCT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lk = lane >> 5;
    for (int tile = wave; tile < TILES; tile += NT / 64) {
        const int kt = tile / NCT, ct = tile % NCT;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int bb = 0; bb < BM; bb += 2) {
            const float av = sX[(bb + lk) * (CIN + PADF) + kt * 32 + li];
            const float bv = sG[(bb + lk) * (COUT + PADF) + ct * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pW[(size_t)(kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * COUT + ct * 32 + li] = acc[r];
    }
}

__device__ inline void rot6d_fwd(const float* p, float* o) {
    // geometry.py:55-61: x viewed [3,2]; a1 = x[:,0], a2 = x[:,1]; F.normalize eps = 1e-12 (as csrc/geometry.hip rot6d_fwd_kernel)
    const float a1[3] = {p[0], p[2], p[4]}, a2[3] = {p[1], p[3], p[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[r * 3 + 0] = b1[r]; o[r * 3 + 1] = b2[r]; o[r * 3 + 2] = b3[r]; }
}
__device__ inline void rot6d_bwd(const float* p, const float* g, float* o) {
    const float a1[3] = {p[0], p[2], p[4]}, a2[3] = {p[1], p[3], p[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float gb1[3] = {g[0], g[3], g[6]}, gb2[3] = {g[1], g[4], g[7]};
    const float gb3[3] = {g[2], g[5], g[8]};
    gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1]; gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2]; gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
    gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1]; gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2]; gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
    const float t2 = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
    const float gu[3] = {(gb2[0] - b2[0] * t2) / n2, (gb2[1] - b2[1] * t2) / n2, (gb2[2] - b2[2] * t2) / n2};
    const float gub1 = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
    const float ga2[3] = {gu[0] - gub1 * b1[0], gu[1] - gub1 * b1[1], gu[2] - gub1 * b1[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) gb1[k] += -d * gu[k] - gub1 * a2[k];
    const float t1 = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
    const float ga1[3] = {(gb1[0] - b1[0] * t1) / n1, (gb1[1] - b1[1] * t1) / n1, (gb1[2] - b1[2] * t1) / n1};
    o[0] = ga1[0]; o[2] = ga1[1]; o[4] = ga1[2];
    o[1] = ga2[0]; o[3] = ga2[1]; o[5] = ga2[2];
}

// norm_A = d_i g_ij d_j, g = I + A_mask * relu(edge), d = (column sums of g)^-1/2 (utils/graph.py:232-261 on a tensor)
__device__ inline void build_norm_A(const float* A_mask, const float* edge, Small& s, float* g_out /* LDS [576] or NULL */) {
    __shared__ float sg[NJ * NJ];
    __shared__ float sd[NJ];
    const int t = threadIdx.x;
    for (int i = t; i < NJ * NJ; i += NT) sg[i] = ((i / NJ) == (i % NJ) ? 1.f : 0.f) + A_mask[i] * fmaxf(edge[i], 0.f);
    __syncthreads();
    if (t < NJ) {
        float c = 0.f;
        for (int i = 0; i < NJ; ++i) c += sg[i * NJ + t];
        sd[t] = c > 0.f ? 1.0f / sqrtf(c) : 0.f;
    }
    __syncthreads();
    for (int i = t; i < NJ * NJ; i += NT) { s.An[i] = sd[i / NJ] * sg[i] * sd[i % NJ]; if (g_out) g_out[i] = sg[i]; }
    if (g_out && t < NJ) g_out[NJ * NJ + t] = sd[t];
    __syncthreads();
}

// the grouped 1x1 head of joint n: sH[b][o] = bias[o] (+ add[o]) + sum_k sF[b][k] W[o][k]   (sF row stride 128 + PADF)
template <int OUT>
__device__ inline void head_fwd(const float* sF, const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ add, Small& s) {
    const int t = threadIdx.x;
    if (t < BM * OUT) {
        const int b = t / OUT, o = t % OUT;
        float v = bias[o] + (add ? add[o] : 0.f);
        for (int k = 0; k < 128; k += 4) {
            const float4 f = *reinterpret_cast<const float4*>(sF + b * (128 + PADF) + k);
            const float4 w = *reinterpret_cast<const float4*>(W + o * 128 + k);
            v = fmaf(f.x, w.x, v); v = fmaf(f.y, w.y, v); v = fmaf(f.z, w.z, v); v = fmaf(f.w, w.w, v);
        }
        s.h[b * 8 + o] = v;
    }
    __syncthreads();
}
// its backward: sG[b][c] (+)= sum_o sH[b][o] W[o][c];  gW[o][c] = sum_b sH[b][o] sF[b][c];  gbias[o] = sum_b sH[b][o]   (sH zero for rows >= B)
template <int OUT, bool ACCUM>
__device__ inline void head_bwd(const float* sF, const float* __restrict__ W, const Small& s, float* sG, float* __restrict__ gW, float* __restrict__ gbias) {
    constexpr int NG = NT / 128, RPG = BM / NG;
    const int t = threadIdx.x, c = t & 127, half = t >> 7;          // `half`: the row group (NG of them)
    float w[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) w[o] = W[o * 128 + c];
    for (int b = half * RPG; b < half * RPG + RPG; ++b) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < OUT; ++o) v = fmaf(s.h[b * 8 + o], w[o], v);
        if (ACCUM) sG[b * (128 + PADF) + c] += v; else sG[b * (128 + PADF) + c] = v;
    }
    if (half == 0) {
        float acc[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = 0.f;
        for (int b = 0; b < BM; ++b) {
            const float f = sF[b * (128 + PADF) + c];
#pragma unroll
            for (int o = 0; o < OUT; ++o) acc[o] = fmaf(s.h[b * 8 + o], f, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < OUT; ++o) gW[o * 128 + c] = acc[o];
    } else if (half == 1 && c < OUT) {
        float v = 0.f;
        for (int b = 0; b < BM; ++b) v += s.h[b * 8 + c];
        gbias[c] = v;
    }
    __syncthreads();
}

// One graph-convolution layer of joint n: mix (row of the adjacency) -> [32 x CIN] x [CIN x COUT] + bias -> BatchNorm over the joint's
// (batch, channel) values -> ReLU.  Leaves the activation in sOut (row stride COUT + PADF, rows >= B zero) and in the workspace.
template <int CIN, int COUT>
__device__ inline void layer_fwd(const TailArgs& a, const WsMap& w, int l, const float* Aline, const float* xin, int n, float* sIn, float* sOut, Small& s) {
    constexpr int RG = NT / COUT, RPT = BM / RG;
    const int t = threadIdx.x, B = a.B, c = t % COUT, r0 = (t / COUT) * RPT;
    sparse_line(Aline, 1, s);
    mix_rows<CIN, false>(xin, B, s, sIn);
    __syncthreads();
    {   // the mixed input, kept for the backward's weight gradient
        float* ax = a.ws + w.ax[l];
        for (int i = t; i < B * (CIN / 4); i += NT) {
            const int b = i / (CIN / 4), k = (i % (CIN / 4)) * 4;
            *reinterpret_cast<float4*>(ax + ((size_t)b * NJ + n) * CIN + k) = *reinterpret_cast<const float4*>(sIn + b * (CIN + PADF) + k);
        }
    }
    gemm_mfma<CIN, COUT>(sIn, a.L[l].W, sOut);
    float acc[RPT];
    const float bias = a.L[l].bias[c];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = sOut[(r0 + r) * (COUT + PADF) + c] + bias;
    float* ypre = a.ws + w.ypre[l];
    float* act = a.ws + w.act[l];
    const float cnt = (float)B * COUT;
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) if (r0 + r < B) { s1 += acc[r]; ypre[((size_t)(r0 + r) * NJ + n) * COUT + c] = acc[r]; }
    const float mean = block_sum(s1, s) / cnt;
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) if (r0 + r < B) { const float d = acc[r] - mean; s2 = fmaf(d, d, s2); }
    const float var = block_sum(s2, s) / cnt;
    const float invstd = 1.0f / sqrtf(var + a.eps);
    if (t == 0) {
        float* st = a.ws + w.stats + ((size_t)l * NJ + n) * 2;
        st[0] = mean; st[1] = invstd;
        if (a.L[l].running_mean) {
            a.L[l].running_mean[n] = (1.f - a.momentum) * a.L[l].running_mean[n] + a.momentum * mean;
            a.L[l].running_var[n] = (1.f - a.momentum) * a.L[l].running_var[n] + a.momentum * var * (cnt / fmaxf(cnt - 1.f, 1.f));
        }
    }
    const float ga = a.L[l].gamma[n], be = a.L[l].beta[n];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const bool valid = r0 + r < B;
        const float v = valid ? fmaxf(fmaf((acc[r] - mean) * invstd, ga, be), 0.f) : 0.f;
        sOut[(r0 + r) * (COUT + PADF) + c] = v;
        if (valid) act[((size_t)(r0 + r) * NJ + n) * COUT + c] = v;
    }
    __syncthreads();
}

constexpr int SM_FLOATS = 2 * BM * (MAXC + PADF);            // forward: sIn + sOut
constexpr int SM_FLOATS_BWD = 2 * BM * (MAXC + PADF) + BM * (128 + PADF);       // backward: sG + sX + sRes

__global__ __launch_bounds__(NT) void gcn_tail_fwd_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ Small s;
    float* sIn = smem;
    float* sOut = smem + BM * (MAXC + PADF);
    const int n = blockIdx.x, t = threadIdx.x, B = a.B;
    const WsMap w = ws_map(B);
    TAIL_STAMP(0);
    // W^T for the backward's data-gradient products (each workgroup a slice; eight loads in flight per lane)
    for (int l = 0; l < NLAY; ++l) {
        const int cin = lay_cin(l), cout = lay_cout(l), sz = cin * cout;
        float* WT = a.ws + w.WT[l];
        for (int e0 = n * NT + t; e0 < sz; e0 += NJ * NT * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * NJ * NT; v[u] = e < sz ? a.L[l].W[e] : 0.f; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * NJ * NT; if (e < sz) WT[(size_t)(e % cout) * cin + e / cout] = v[u]; }
        }
    }
    TAIL_STAMP(1);
    build_norm_A(a.A_mask, a.edge, s, nullptr);
    // ---- phase 0: pose head 0 on the raw features, layer 0 (rot -> pos), coordinate head 0
    for (int i = t; i < BM * 128; i += NT) {
        const int b = i >> 7, k = i & 127;
        sOut[b * (128 + PADF) + k] = b < B ? a.x[((size_t)b * NJ + n) * 128 + k] : 0.f;
    }
    __syncthreads();
    head_fwd<6>(sOut, a.Wp[0] + (size_t)n * 6 * 128, a.bp[0] + n * 6, a.mean_pose + n * 6, s);
    if (t < B) {
        float R[9];
        float* p6 = a.ws + w.pose6[0] + ((size_t)t * NJ + n) * 6;
#pragma unroll
        for (int o = 0; o < 6; ++o) p6[o] = s.h[t * 8 + o];
        rot6d_fwd(&s.h[t * 8], R);
#pragma unroll
        for (int e = 0; e < 9; ++e) a.jr0[(size_t)t * (NJ * 9) + n * 9 + e] = R[e];
    }
    layer_fwd<128, 128>(a, w, 0, a.A_r2p + n * NJ, a.x, n, sIn, sOut, s);
    head_fwd<3>(sOut, a.Wc[0] + (size_t)n * 3 * 128, a.bc[0] + n * 3, nullptr, s);
    if (t < B * 3) a.jp0[((size_t)(t / 3) * NJ + n) * 3 + t % 3] = s.h[(t / 3) * 8 + t % 3];
    TAIL_STAMP(2);
    danet::grid_barrier_fenced(a.bar, NJ, 257u);
    TAIL_STAMP(3);
    // ---- phases 1 - 3: the refinement layers on norm_A
    layer_fwd<128, 256>(a, w, 1, s.An + n * NJ, a.ws + w.act[0], n, sIn, sOut, s);
    TAIL_STAMP(4);
    danet::grid_barrier_fenced(a.bar, NJ, 258u);
    TAIL_STAMP(5);
    layer_fwd<256, 256>(a, w, 2, s.An + n * NJ, a.ws + w.act[1], n, sIn, sOut, s);
    TAIL_STAMP(6);
    danet::grid_barrier_fenced(a.bar, NJ, 259u);
    TAIL_STAMP(7);
    layer_fwd<256, 128>(a, w, 3, s.An + n * NJ, a.ws + w.act[2], n, sIn, sOut, s);
    {   // pos_ref = pos_init + h3 (this joint's rows), coordinate head 1
        const float* act0 = a.ws + w.act[0];
        float* pr = a.ws + w.posref;
        for (int i = t; i < BM * 128; i += NT) {
            const int b = i >> 7, k = i & 127;
            if (b < B) {
                const float v = sOut[b * (128 + PADF) + k] + act0[((size_t)b * NJ + n) * 128 + k];
                sOut[b * (128 + PADF) + k] = v;
                pr[((size_t)b * NJ + n) * 128 + k] = v;
            }
        }
        __syncthreads();
        head_fwd<3>(sOut, a.Wc[1] + (size_t)n * 3 * 128, a.bc[1] + n * 3, nullptr, s);
        if (t < B * 3) a.jp1[((size_t)(t / 3) * NJ + n) * 3 + t % 3] = s.h[(t / 3) * 8 + t % 3];
    }
    TAIL_STAMP(8);
    danet::grid_barrier_fenced(a.bar, NJ, 260u);
    TAIL_STAMP(9);
    // ---- phase 4: pos -> rot, pose head 1, rot6d
    layer_fwd<128, 128>(a, w, 4, a.A_p2r + n * NJ, a.ws + w.posref, n, sIn, sOut, s);
    head_fwd<6>(sOut, a.Wp[1] + (size_t)n * 6 * 128, a.bp[1] + n * 6, a.mean_pose + n * 6, s);
    if (t < B) {
        float R[9];
        float* p6 = a.ws + w.pose6[1] + ((size_t)t * NJ + n) * 6;
#pragma unroll
        for (int o = 0; o < 6; ++o) p6[o] = s.h[t * 8 + o];
        rot6d_fwd(&s.h[t * 8], R);
#pragma unroll
        for (int e = 0; e < 9; ++e) a.pose[(size_t)t * (NJ * 9) + n * 9 + e] = R[e];
    }
    TAIL_STAMP(10);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of one layer for joint n.  In: sG = d(activation) [32][COUT + PADF].  Out: d(ax) in sX [32][CIN + PADF] and in the scratch,
// this joint's partial d W / d bias in the scratch, d gamma / d beta of the joint, and (refinement layers) its row of d norm_A.
template <int CIN, int COUT>
__device__ inline void layer_bwd(const TailArgs& a, const WsMap& w, const ScMap& sc, int l, const float* Aline, const float* xin, int n, bool want_dA,
                                 float* sG, float* sX, Small& s)
{
    constexpr int RG = NT / COUT, RPT = BM / RG;
    const int t = threadIdx.x, B = a.B;
    {
        const int c = t % COUT, r0 = (t / COUT) * RPT;
        const float* ypre = a.ws + w.ypre[l];
        const float* act = a.ws + w.act[l];
        const float* st = a.ws + w.stats + ((size_t)l * NJ + n) * 2;
        const float mean = st[0], invstd = st[1], ga = a.L[l].gamma[n], cnt = (float)B * COUT;
        float gz[RPT], xh[RPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool valid = r0 + r < B;
            const size_t o = ((size_t)(r0 + r) * NJ + n) * COUT + c;
            const float y = valid ? ypre[o] : 0.f, on = valid ? act[o] : 0.f;
            xh[r] = (y - mean) * invstd;
            gz[r] = (valid && on > 0.f) ? sG[(r0 + r) * (COUT + PADF) + c] : 0.f;
            s1 += gz[r]; s2 = fmaf(gz[r], xh[r], s2);
        }
        s1 = block_sum(s1, s); s2 = block_sum(s2, s);
        if (t == 0) { a.gbeta[l][n] = s1; a.ggamma[l][n] = s2; }
        const float m1 = s1 / cnt, m2 = s2 / cnt, k0 = ga * invstd;
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const float gy = r0 + r < B ? k0 * (gz[r] - m1 - xh[r] * m2) : 0.f;
            sG[(r0 + r) * (COUT + PADF) + c] = gy;
            cs += gy;
        }
        s.colsum[t] = cs;
        __syncthreads();
        if (t < COUT) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g) v += s.colsum[g * COUT + c];
            a.scratch[sc.partb[l] + (size_t)n * COUT + c] = v;
        }
    }
    // ax as the forward left it (this joint's rows)
    {
        const float* ax = a.ws + w.ax[l];
        for (int i = t; i < BM * (CIN / 4); i += NT) {
            const int b = i / (CIN / 4), k = (i % (CIN / 4)) * 4;
            *reinterpret_cast<float4*>(sX + b * (CIN + PADF) + k) = b < B ? *reinterpret_cast<const float4*>(ax + ((size_t)b * NJ + n) * CIN + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (want_dA) sparse_line(Aline, 1, s);          // (the pattern of the row, for d norm_A below)
    __syncthreads();
    // partial d W [CIN][COUT] = ax^T gy
    wgrad_mfma<CIN, COUT>(sX, sG, a.scratch + sc.partW[l] + (size_t)n * CIN * COUT);
    __syncthreads();                                       // every wave is done with ax
    {   // d ax [32][CIN] = gy W^T
        gemm_mfma<COUT, CIN>(sG, a.ws + w.WT[l], sX);
        float* dax = a.scratch + sc.dax[l];
        for (int i = t; i < B * (CIN / 4); i += NT) {
            const int b = i / (CIN / 4), k = (i % (CIN / 4)) * 4;
            *reinterpret_cast<float4*>(dax + ((size_t)b * NJ + n) * CIN + k) = *reinterpret_cast<const float4*>(sX + b * (CIN + PADF) + k);
        }
    }
    if (want_dA) {      // d norm_A[n][m] += sum_{b,k} d ax[b][k] x[b][m][k] for the m of the row's pattern (s.idx still holds them)
        const int nnz = s.nnz;
        float acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
        const int nnz4 = (nnz + 3) & ~3;
        constexpr int C4 = CIN / 4;
        for (int i = t; i < B * C4; i += NT) {
            const int b = i / C4, k = (i % C4) * 4;
            const float4 d = *reinterpret_cast<const float4*>(sX + b * (CIN + PADF) + k);
            const float* base = xin + (size_t)b * NJ * CIN + k;
#pragma unroll
            for (int j = 0; j < NJ; j += 4) {
                if (j < nnz4) {
                    float4 q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const float4*>(base + (size_t)s.idx[j + u] * CIN);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[j + u] += (d.x * q[u].x + d.y * q[u].y) + (d.z * q[u].z + d.w * q[u].w);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float v = acc[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((t & 63) == 0) s.colsum[(t >> 6) * NJ + j] = v;
        }
        __syncthreads();
        if (t < nnz) {
            float v = 0.f;
            for (int wv = 0; wv < NT / 64; ++wv) v += s.colsum[wv * NJ + t];
            s.dA[s.idx[t]] += v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void gcn_tail_bwd_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ Small s;
    __shared__ float sgA[NJ * NJ + NJ];          // workgroup 0, last phase: g and d of build_norm_A
    float* sG = smem;
    float* sX = smem + BM * (MAXC + PADF);
    float* sRes = smem + 2 * BM * (MAXC + PADF);
    const int n = blockIdx.x, t = threadIdx.x, B = a.B;
    const WsMap w = ws_map(B);
    const ScMap sc = sc_map(B);
    TAIL_STAMP(16);
    build_norm_A(a.A_mask, a.edge, s, n == 0 ? sgA : nullptr);
    if (t < NJ) s.dA[t] = 0.f;
    // ---- layer 4 (pos -> rot) from the pose head
    for (int i = t; i < BM * 8; i += NT) s.h[i] = 0.f;
    __syncthreads();
    if (t < B && a.g_pose) rot6d_bwd(a.ws + w.pose6[1] + ((size_t)t * NJ + n) * 6, a.g_pose + (size_t)t * (NJ * 9) + n * 9, &s.h[t * 8]);
    for (int i = t; i < BM * 128; i += NT) {       // the head's input: this joint's rot_ref rows
        const int b = i >> 7, k = i & 127;
        sX[b * (128 + PADF) + k] = b < B ? a.ws[w.act[4] + ((size_t)b * NJ + n) * 128 + k] : 0.f;
    }
    __syncthreads();
    head_bwd<6, false>(sX, a.Wp[1] + (size_t)n * 6 * 128, s, sG, a.gWp[1] + (size_t)n * 6 * 128, a.gbp[1] + n * 6);
    layer_bwd<128, 128>(a, w, sc, 4, a.A_p2r + n * NJ, a.ws + w.posref, n, false, sG, sX, s);
    TAIL_STAMP(17);
    danet::grid_barrier_fenced(a.bar, NJ, 513u);
    TAIL_STAMP(18);
    // ---- d pos_ref = p2r_A^T d ax4 + coordinate head 1; layer 3
    sparse_line(a.A_p2r + n, NJ, s);
    mix_rows<128, false>(a.scratch + sc.dax[4], B, s, sG);
    for (int i = t; i < BM * 8; i += NT) s.h[i] = 0.f;
    __syncthreads();
    if (t < B * 3 && a.g_jp1) s.h[(t / 3) * 8 + t % 3] = a.g_jp1[((size_t)(t / 3) * NJ + n) * 3 + t % 3];
    for (int i = t; i < BM * 128; i += NT) {
        const int b = i >> 7, k = i & 127;
        sX[b * (128 + PADF) + k] = b < B ? a.ws[w.posref + ((size_t)b * NJ + n) * 128 + k] : 0.f;
    }
    __syncthreads();
    head_bwd<3, true>(sX, a.Wc[1] + (size_t)n * 3 * 128, s, sG, a.gWc[1] + (size_t)n * 3 * 128, a.gbc[1] + n * 3);
    for (int i = t; i < BM * 128; i += NT) sRes[(i >> 7) * (128 + PADF) + (i & 127)] = sG[(i >> 7) * (128 + PADF) + (i & 127)];
    __syncthreads();
    layer_bwd<256, 128>(a, w, sc, 3, s.An + n * NJ, a.ws + w.act[2], n, true, sG, sX, s);
    TAIL_STAMP(19);
    danet::grid_barrier_fenced(a.bar, NJ, 514u);
    TAIL_STAMP(20);
    // ---- layer 2
    sparse_line(s.An + n, NJ, s);
    mix_rows<256, false>(a.scratch + sc.dax[3], B, s, sG);
    __syncthreads();
    layer_bwd<256, 256>(a, w, sc, 2, s.An + n * NJ, a.ws + w.act[1], n, true, sG, sX, s);
    TAIL_STAMP(21);
    danet::grid_barrier_fenced(a.bar, NJ, 515u);
    TAIL_STAMP(22);
    // ---- layer 1
    sparse_line(s.An + n, NJ, s);
    mix_rows<256, false>(a.scratch + sc.dax[2], B, s, sG);
    __syncthreads();
    layer_bwd<128, 256>(a, w, sc, 1, s.An + n * NJ, a.ws + w.act[0], n, true, sG, sX, s);
    if (t < NJ) a.scratch[sc.dA + n * NJ + t] = s.dA[t];
    TAIL_STAMP(23);
    danet::grid_barrier_fenced(a.bar, NJ, 516u);
    TAIL_STAMP(24);
    // ---- d pos_init = (through the residual) + norm_A^T d ax1 + coordinate head 0; layer 0
    sparse_line(s.An + n, NJ, s);
    for (int i = t; i < BM * 128; i += NT) sG[(i >> 7) * (128 + PADF) + (i & 127)] = sRes[(i >> 7) * (128 + PADF) + (i & 127)];
    __syncthreads();
    mix_rows<128, true>(a.scratch + sc.dax[1], B, s, sG);
    for (int i = t; i < BM * 8; i += NT) s.h[i] = 0.f;
    __syncthreads();
    if (t < B * 3 && a.g_jp0) s.h[(t / 3) * 8 + t % 3] = a.g_jp0[((size_t)(t / 3) * NJ + n) * 3 + t % 3];
    for (int i = t; i < BM * 128; i += NT) {
        const int b = i >> 7, k = i & 127;
        sX[b * (128 + PADF) + k] = b < B ? a.ws[w.act[0] + ((size_t)b * NJ + n) * 128 + k] : 0.f;
    }
    __syncthreads();
    head_bwd<3, true>(sX, a.Wc[0] + (size_t)n * 3 * 128, s, sG, a.gWc[0] + (size_t)n * 3 * 128, a.gbc[0] + n * 3);
    layer_bwd<128, 128>(a, w, sc, 0, a.A_r2p + n * NJ, a.x, n, false, sG, sX, s);
    TAIL_STAMP(25);
    danet::grid_barrier_fenced(a.bar, NJ, 517u);
    TAIL_STAMP(26);
    // ---- d rot_feats = r2p_A^T d ax0 + pose head 0
    sparse_line(a.A_r2p + n, NJ, s);
    mix_rows<128, false>(a.scratch + sc.dax[0], B, s, sG);
    for (int i = t; i < BM * 8; i += NT) s.h[i] = 0.f;
    __syncthreads();
    if (t < B && a.g_jr0) rot6d_bwd(a.ws + w.pose6[0] + ((size_t)t * NJ + n) * 6, a.g_jr0 + (size_t)t * (NJ * 9) + n * 9, &s.h[t * 8]);
    for (int i = t; i < BM * 128; i += NT) {
        const int b = i >> 7, k = i & 127;
        sX[b * (128 + PADF) + k] = b < B ? a.x[((size_t)b * NJ + n) * 128 + k] : 0.f;
    }
    __syncthreads();
    head_bwd<6, true>(sX, a.Wp[0] + (size_t)n * 6 * 128, s, sG, a.gWp[0] + (size_t)n * 6 * 128, a.gbp[0] + n * 6);
    for (int i = t; i < B * 128; i += NT) a.gx[((size_t)(i >> 7) * NJ + n) * 128 + (i & 127)] = sG[(i >> 7) * (128 + PADF) + (i & 127)];
    TAIL_STAMP(27);
    // ---- the joints' partial weight / bias gradients, summed in joint order (each workgroup a slice of every layer)
    for (int l = 0; l < NLAY; ++l) {
        const int sz = lay_cin(l) * lay_cout(l), co = lay_cout(l);
        const float* pW = a.scratch + sc.partW[l];
        for (int e = n * NT + t; e < sz; e += NJ * NT) {
            float v[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] = pW[(size_t)j * sz + e];
            __builtin_amdgcn_sched_barrier(0);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc += v[j];
            a.gW[l][e] = acc;
        }
        const float* pb = a.scratch + sc.partb[l];
        for (int e = n * NT + t; e < co; e += NJ * NT) {
            float acc = 0.f;
            for (int j = 0; j < NJ; ++j) acc += pb[(size_t)j * co + e];
            a.gb[l][e] = acc;
        }
    }
    TAIL_STAMP(28);
    // ---- d edge_importance from d norm_A (workgroup 0): norm_A = d_i g_ij d_j, d_j = (sum_i g_ij)^-1/2, g = I + mask relu(edge)
    if (n == 0) {
        __shared__ float sGA[NJ * NJ];
        __shared__ float sdd[NJ];
        const float* g = sgA; const float* d = sgA + NJ * NJ;
        for (int i = t; i < NJ * NJ; i += NT) sGA[i] = a.scratch[sc.dA + i];
        __syncthreads();
        if (t < NJ) {        // dL/dd_k = sum_j G_kj g_kj d_j + sum_i G_ik d_i g_ik ;  dL/dDl_k = -1/2 d_k^3 dL/dd_k
            float v = 0.f;
            for (int j = 0; j < NJ; ++j) v += sGA[t * NJ + j] * g[t * NJ + j] * d[j] + sGA[j * NJ + t] * d[j] * g[j * NJ + t];
            sdd[t] = -0.5f * d[t] * d[t] * d[t] * v;
        }
        __syncthreads();
        for (int i = t; i < NJ * NJ; i += NT) {
            const int r = i / NJ, c = i % NJ;
            const float dg = sGA[i] * d[r] * d[c] + sdd[c];
            a.gedge[i] = a.edge[i] > 0.f ? dg * a.A_mask[i] : 0.f;
        }
    }
}

}  // namespace

// Floats of the forward's workspace (kept until the backward) and of the backward's scratch for a batch of B (<= 32) rows.
extern "C" size_t danet_gcn_tail_ws_floats(int B) { return ws_map(B).total; }
extern "C" size_t danet_gcn_tail_scratch_floats(int B) { return sc_map(B).total; }
extern "C" int danet_gcn_tail_max_batch(void) { return BM; }
// wall-clock stamps (100 MHz) of workgroup 0 of the last launches: [0..15] forward, [16..31] backward (tools/gcn_tail_bench.py)
extern "C" int danet_gcn_tail_debug(long long* out32) {
    DANET_ENTER();
    DANET_CHECK_ARG(out32, "gcn_tail_debug: NULL");
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tail_dbg), sizeof(long long) * 32, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "gcn_tail_debug: %s", hipGetErrorString(e));
    return DANET_OK;
}

static int tail_check(const TailArgs* a, const char* what) {
    DANET_CHECK_ARG(a && a->x && a->ws && a->bar && a->B >= 1 && a->B <= BM && a->A_r2p && a->A_p2r && a->A_mask && a->edge && a->mean_pose,
                    "%s: bad arguments (1 <= B <= %d)", what, BM);
    for (int l = 0; l < NLAY; ++l)
        DANET_CHECK_ARG(a->L[l].W && a->L[l].bias && a->L[l].gamma && a->L[l].beta, "%s: layer %d lacks a parameter", what, l);
    for (int i = 0; i < 2; ++i)
        DANET_CHECK_ARG(a->Wp[i] && a->bp[i] && a->Wc[i] && a->bc[i], "%s: head %d lacks a parameter", what, i);
    return DANET_OK;
}

extern "C" int danet_gcn_tail_forward(const void* args, void* stream)
{
    DANET_ENTER();
    const TailArgs* a = (const TailArgs*)args;
    if (int e = tail_check(a, "gcn_tail_forward")) return e;
    DANET_CHECK_ARG(a->jr0 && a->jp0 && a->jp1 && a->pose, "gcn_tail_forward: missing outputs");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_tail_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_FLOATS * 4); attr = true; }
    hipLaunchKernelGGL(gcn_tail_fwd_kernel, dim3(NJ), dim3(NT), SM_FLOATS * 4, (hipStream_t)stream, *a);
    DANET_CHECK_LAUNCH("gcn_tail_fwd_kernel");
    return DANET_OK;
}

extern "C" int danet_gcn_tail_backward(const void* args, void* stream)
{
    DANET_ENTER();
    const TailArgs* a = (const TailArgs*)args;
    if (int e = tail_check(a, "gcn_tail_backward")) return e;
    DANET_CHECK_ARG(a->gx && a->gedge && a->scratch, "gcn_tail_backward: missing outputs");
    for (int l = 0; l < NLAY; ++l)
        DANET_CHECK_ARG(a->gW[l] && a->gb[l] && a->ggamma[l] && a->gbeta[l], "gcn_tail_backward: layer %d lacks a gradient buffer", l);
    for (int i = 0; i < 2; ++i)
        DANET_CHECK_ARG(a->gWp[i] && a->gbp[i] && a->gWc[i] && a->gbc[i], "gcn_tail_backward: head %d lacks a gradient buffer", i);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_tail_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_FLOATS_BWD * 4); attr = true; }
    hipLaunchKernelGGL(gcn_tail_bwd_kernel, dim3(NJ), dim3(NT), SM_FLOATS_BWD * 4, (hipStream_t)stream, *a);
    DANET_CHECK_LAUNCH("gcn_tail_bwd_kernel");
    return DANET_OK;
}
