// Fused SMPL layer for gfx950: shape/pose blend-shapes, joint regression, 24-joint
// kinematic chain and linear-blend skinning (forward and backward).
//
// Replaces the arithmetic behind /root/reference/models/smpl.py:27-46 (smplx LBS).
//
// Decomposition (all fp32):
//   prep      one wave per batch item: rest joints J = J_template + J_shapedirs.beta (the
//             J_regressor contraction is folded into constants at model load), the 24-joint
//             chain, A_j = [Rg_j | Jp_j - Rg_j J_j], and the transposed pose feature pfT.
//   main      grid (vertex tiles of 64, batch groups of 8).  One lane per vertex COORDINATE
//             streams its posedirs column (coalesced 768-B rows); the pose feature of the 8
//             batch items is wave-uniform, so it is fetched with scalar loads and the inner
//             loop is 8 v_fmac per 4-byte load.  The 24 joint transforms of the 8 items sit in
//             LDS for the skinning pass.  posedirs (17 MB, the dominant HBM term) is read once
//             per batch group.
//   finalize  landmark picks + fixed-order sum of the extra-joint partials (deterministic).
// Backward mirrors this: main_bwd produces per-tile partials of dA, d(pose feature), d(beta);
// finalize_bwd reduces them in fixed order and back-propagates through the chain.
#include "common.h"
#include "grid_barrier.h"

namespace {

constexpr int NJ = 24;
constexpr int NPB = 207;           // pose-basis rows
constexpr int NPB_PAD = 208;
constexpr int TV = 64;             // vertices per tile
constexpr int TC = TV * 3;         // coordinates per tile
constexpr int NBG = 8;             // batch items per block
constexpr int WPAD = 25;           // padded row of the staged skin weights (bank spread)
constexpr int NB_MAX = 16;
constexpr int NE_MAX = 28;
constexpr int NL_MAX = 32;

// ctx layout per batch item (floats)
constexpr int CTX_A = 0;           // [24][12]
constexpr int CTX_J = 288;         // [24][3] rest joints
constexpr int CTX_JP = 360;        // [24][3] posed joints
constexpr int CTX_RG = 432;        // [24][9] global rotations
constexpr int CTX_STRIDE = 648;

// Forward main kernel: batch items per workgroup x vertices per tile = 512 (vertex, item) pairs.  Reading posedirs (17 MB) once
// for all 32 items of a batch (32 items x 16 vertices per workgroup) was built and measured in round 3: SLOWER than four batch
// groups of 8 (the re-reads hit the L2 / infinity cache; narrow tiles multiply the per-workgroup fixed cost: 37 KB of joint
// transforms per workgroup, 192-byte row pieces), so 8 x 64 stays the default; DANET_LBS_NBG selects the other instantiations.
inline int fwd_nbg_of(int B) {
    static const int forced = getenv("DANET_LBS_NBG") ? atoi(getenv("DANET_LBS_NBG")) : 0;      // A-B timing knob: 8, 16 or 32
    if (forced == 8 || forced == 16 || forced == 32) return forced;
    (void)B;
    return 8;      // measured at B = 32 (rocprofv3, main kernel): 8 items x 64 vertices 26.7 us, 16 x 32 39.4 us, 32 x 16 41.9 us
}
inline int fwd_tv_of(int B) { return 512 / fwd_nbg_of(B); }
inline int bpad_of(int B) { const int g = fwd_nbg_of(B); return (B + g - 1) / g * g; }      // (a multiple of NBG = 8: the backward's groups)
inline int ntiles_of(int V) { return (V + TV - 1) / TV; }                                   // backward tiles
inline int fwd_ntiles_of(int V, int B) { const int tv = fwd_tv_of(B); return (V + tv - 1) / tv; }

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void smpl_prep_kernel(
    const float* __restrict__ betas, const float* __restrict__ rot,
    const float* __restrict__ J_template, const float* __restrict__ J_dirs,
    const int* __restrict__ parents, int B, int NB, int Bpad, int NJ54,
    float* __restrict__ ctx, float* __restrict__ pfT, float* __restrict__ joints54)
{
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ float sR[216], sJ[72], sRg[216], sJp[72];
    float* c = ctx + (size_t)b * CTX_STRIDE;
    if (b >= B) {   // padding items of the last batch group: harmless zeros
        for (int i = t; i < CTX_STRIDE; i += 64) c[i] = 0.f;
        for (int k = t; k < NPB_PAD; k += 64) pfT[(size_t)k * Bpad + b] = 0.f;
        return;
    }
    for (int i = t; i < 216; i += 64) sR[i] = rot[(size_t)b * 216 + i];
    for (int i = t; i < 72; i += 64) {
        float s = J_template[i];
        for (int l = 0; l < NB; ++l) s += J_dirs[i * NB + l] * betas[(size_t)b * NB + l];
        sJ[i] = s;
    }
    __syncthreads();
    for (int k = t; k < NPB_PAD; k += 64) {
        float v = 0.f;
        if (k < NPB) {
            const int e = k % 9;
            v = sR[9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
        pfT[(size_t)k * Bpad + b] = v;
    }
    for (int i = 0; i < NJ; ++i) {
        const int p = parents[i];
        if (t < 9) {
            const int r = t / 3, cc = t % 3;
            float s;
            if (p < 0) s = sR[i * 9 + t];
            else s = sRg[p * 9 + r * 3 + 0] * sR[i * 9 + 0 + cc] + sRg[p * 9 + r * 3 + 1] * sR[i * 9 + 3 + cc] +
                     sRg[p * 9 + r * 3 + 2] * sR[i * 9 + 6 + cc];
            sRg[i * 9 + t] = s;
        } else if (t < 12) {
            const int r = t - 9;
            float s;
            if (p < 0) s = sJ[i * 3 + r];
            else {
                s = sJp[p * 3 + r];
                for (int m = 0; m < 3; ++m) s += sRg[p * 9 + r * 3 + m] * (sJ[i * 3 + m] - sJ[p * 3 + m]);
            }
            sJp[i * 3 + r] = s;
        }
        __syncthreads();
    }
    for (int idx = t; idx < 288; idx += 64) {
        const int j = idx / 12, e = idx % 12, r = e / 4, cc = e % 4;
        float v;
        if (cc < 3) v = sRg[j * 9 + r * 3 + cc];
        else {
            v = sJp[j * 3 + r];
            for (int m = 0; m < 3; ++m) v -= sRg[j * 9 + r * 3 + m] * sJ[j * 3 + m];
        }
        c[CTX_A + idx] = v;
    }
    for (int i = t; i < 72; i += 64) {
        c[CTX_J + i] = sJ[i];
        c[CTX_JP + i] = sJp[i];
        if (joints54) joints54[(size_t)b * NJ54 * 3 + i] = sJp[i];
    }
    for (int i = t; i < 216; i += 64) c[CTX_RG + i] = sRg[i];
}

// ------------------------------------------------------------------------------------------
__device__ long long g_lbs_dbg[16];     // phase time stamps of workgroup (0, 0) (tools: danet_smpl_lbs_debug)
#define LBS_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_lbs_dbg[i] = clock64(); } while (0)

template <int TV, int NBG>
__global__ __launch_bounds__(256) void smpl_lbs_fwd_kernel(
    const float* __restrict__ v_template, const float* __restrict__ shapedirs,
    const float* __restrict__ posedirs, const float* __restrict__ lbs_weights,
    const float* __restrict__ Jx, const float* __restrict__ betas,
    const float* __restrict__ ctx, const float* __restrict__ pfT,
    int B, int Bpad, int V, int NB, int NE,
    float* __restrict__ verts, float* __restrict__ v_posed_out, float* __restrict__ jx_partial)
{
    constexpr int TC = TV * 3;                       // coordinates per tile
    constexpr int NQ = (TC + 63) / 64;               // coordinates per lane in the blend-shape pass
    const int tile = blockIdx.x, b0 = blockIdx.y * NBG, t = threadIdx.x;
    const int v0 = tile * TV;
    const int C = V * 3;
    __shared__ __attribute__((aligned(16))) float sA[NBG][288];
    __shared__ float sW[TV][WPAD];
    __shared__ float sVp[NBG][TC];
    __shared__ float sPart[4][NBG][NQ * 64];
    __shared__ float sB[NBG][NB_MAX];               // betas of the group (zero beyond NB / B)
    __shared__ float sSd[TC][NB_MAX + 1];           // shapedirs rows of the tile's coordinates
    __shared__ float sBase[TC];                     // v_template
    __shared__ float sJx[NE_MAX][TV];               // extra-joint regressor columns of the tile
    LBS_STAMP(8);
    __shared__ __attribute__((aligned(16))) float sPf[NBG > 8 ? 4 : 1][NBG > 8 ? NPB_PAD / 4 : 1][NBG];      // per wave: its rows' pose features (NBG > 8)

#pragma unroll
    for (int it_ = 0; it_ < (NBG * 288 + 255) / 256; ++it_) {
        const int i = t + it_ * 256;
        if (i >= NBG * 288) break;
        const int bb = i / 288, e = i % 288;
        sA[bb][e] = ctx[(size_t)(b0 + bb) * CTX_STRIDE + CTX_A + e];
    }
#pragma unroll
    for (int it_ = 0; it_ < (TV * NJ + 255) / 256; ++it_) {
        const int i = t + it_ * 256;
        if (i >= TV * NJ) break;
        const int vv = i / NJ, j = i % NJ, v = v0 + vv;
        sW[vv][j] = v < V ? lbs_weights[(size_t)v * NJ + j] : 0.f;
    }
    for (int i = t; i < NBG * NB_MAX; i += 256) {
        const int bb = i / NB_MAX, l = i - bb * NB_MAX;
        sB[bb][l] = (l < NB && b0 + bb < B) ? betas[(size_t)(b0 + bb) * NB + l] : 0.f;
    }
    for (int i = t; i < TC * NB_MAX; i += 256) {
        const int cc = i / NB_MAX, l = i - cc * NB_MAX, c = v0 * 3 + cc;
        sSd[cc][l] = (l < NB && c < C) ? shapedirs[(size_t)c * NB + l] : 0.f;
    }
    for (int i = t; i < TC; i += 256) { const int c = v0 * 3 + i; sBase[i] = v_template[c < C ? c : C - 1]; }
    if (jx_partial)
        for (int i = t; i < NE * TV; i += 256) {
            const int e = i / TV, vv = i - e * TV;
            sJx[e][vv] = v0 + vv < V ? Jx[(size_t)e * V + v0 + vv] : 0.f;
        }
    LBS_STAMP(9);
    {
        // pose blend-shapes: wave w owns 52 pose-basis rows, each lane 3 coordinates of the tile;
        // 39 independent 4-byte loads are in flight per unrolled batch, the pose feature arrives
        // through scalar loads (wave-uniform)
        const int w = t >> 6, lane = t & 63;
        float acc[NQ][NBG];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bb = 0; bb < NBG; ++bb) acc[q][bb] = 0.f;
        int cq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int c = v0 * 3 + lane + 64 * q; cq[q] = c < C ? c : C - 1; }    // (lanes past the tile's TC coordinates compute values nobody reads)
        const float* pf = pfT + b0;
        const int kbeg = w * (NPB_PAD / 4);
        if constexpr (NBG <= 8) {
#pragma unroll 13
            for (int kk = 0; kk < NPB_PAD / 4; ++kk) {
                const int k = kbeg + kk;                       // row 207 of pfT is zero padding
                const int kr = k < NPB ? k : NPB - 1;
                float pq[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) pq[q] = posedirs[(size_t)kr * C + cq[q]];
#pragma unroll
                for (int bb = 0; bb < NBG; ++bb) {
                    const float f = pf[(size_t)k * Bpad + bb];      // wave-uniform: scalar loads
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q][bb] += f * pq[q];
                }
            }
        } else {
            // 16 / 32 items: the pose features of a row would need 16 / 32 SGPRs per row in flight (measured: 2x slower than the
            // 8-item kernel).  The wave parks its 52 rows x NBG features in LDS and reads them back as broadcast ds_read_b128.
            float* const myPf = &sPf[w][0][0];
            for (int i = lane; i < (NPB_PAD / 4) * NBG; i += 64) {
                const int kk = i / NBG, bb = i - kk * NBG;
                myPf[i] = pf[(size_t)(kbeg + kk) * Bpad + bb];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            constexpr int UB = 26;                            // rows per batch: all of a batch's loads are issued before its arithmetic
            static_assert((NPB_PAD / 4) % UB == 0, "UB");
            for (int kk0 = 0; kk0 < NPB_PAD / 4; kk0 += UB) {
                float pr[UB][NQ];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int k = kbeg + kk0 + u;
                    const int kr = k < NPB ? k : NPB - 1;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) pr[u][q] = posedirs[(size_t)kr * C + cq[q]];
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
#pragma unroll
                    for (int b4 = 0; b4 < NBG; b4 += 4) {
                        const float4 f = *reinterpret_cast<const float4*>(myPf + (kk0 + u) * NBG + b4);
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            acc[q][b4 + 0] += f.x * pr[u][q]; acc[q][b4 + 1] += f.y * pr[u][q];
                            acc[q][b4 + 2] += f.z * pr[u][q]; acc[q][b4 + 3] += f.w * pr[u][q];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bb = 0; bb < NBG; ++bb) sPart[w][bb][lane + 64 * q] = acc[q][bb];
    }
    __syncthreads();
    LBS_STAMP(10);
    // shape blend-shapes + the four waves' pose partial sums (fixed order): (coordinate, item) pairs spread over all threads,
    // the tile's shapedirs rows and the group's betas come from LDS
    for (int o = t; o < TC * NBG; o += 256) {
        const int bb = o / TC, cl_ = o - bb * TC;
        const int c = v0 * 3 + cl_;
        const bool valid = c < C;
        float a = sBase[cl_] + ((sPart[0][bb][cl_] + sPart[1][bb][cl_]) + (sPart[2][bb][cl_] + sPart[3][bb][cl_]));
#pragma unroll
        for (int l = 0; l < NB_MAX; ++l) a += sSd[cl_][l] * sB[bb][l];            // (entries l >= NB are zero)
        sVp[bb][cl_] = a;
        if (v_posed_out && valid && b0 + bb < B) v_posed_out[(size_t)(b0 + bb) * C + c] = a;
    }
    __syncthreads();
    LBS_STAMP(11);
    for (int pair = t; pair < TV * NBG; pair += 256) {
        const int vv = pair & (TV - 1), bb = pair / TV;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            const float w = sW[vv][j];
            const float4* a4 = reinterpret_cast<const float4*>(&sA[bb][j * 12]);       // (16-byte reads: with 16-vertex tiles a wave spans four items)
#pragma unroll
            for (int e4 = 0; e4 < 3; ++e4) {
                const float4 a = a4[e4];
                T[e4 * 4 + 0] += w * a.x; T[e4 * 4 + 1] += w * a.y; T[e4 * 4 + 2] += w * a.z; T[e4 * 4 + 3] += w * a.w;
            }
        }
        const float x = sVp[bb][vv * 3 + 0], y = sVp[bb][vv * 3 + 1], z = sVp[bb][vv * 3 + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = T[r * 4 + 0] * x + T[r * 4 + 1] * y + T[r * 4 + 2] * z + T[r * 4 + 3];
        const int v = v0 + vv, b = b0 + bb;
#pragma unroll
        for (int r = 0; r < 3; ++r) sVp[bb][vv * 3 + r] = o[r];
        if (v < V && b < B) {
            float* dst = verts + ((size_t)b * V + v) * 3;
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
        }
    }
    __syncthreads();
    LBS_STAMP(12);
    if (jx_partial) {
        const int NO = NE * 3;
        for (int o = t; o < NBG * NO; o += 256) {
            const int bb = o / NO, e = (o % NO) / 3, k = o % 3;
            float s = 0.f;
#pragma unroll 16
            for (int vv = 0; vv < TV; ++vv) s += sJx[e][vv] * sVp[bb][vv * 3 + k];
            jx_partial[((size_t)tile * Bpad + b0 + bb) * NO + e * 3 + k] = s;
        }
    }
    LBS_STAMP(13);
}

// ------------------------------------------------------------------------------------------
// The forward as ONE launch (/root/reference/models/smpl.py:27-46 is one call; prep -> main -> finalize above are three
// launches with two hand-overs through memory, 43 us back to back but 66 us inside the train step's hipGraph).  Grid as the
// main kernel (vertex tiles of 64 x batch groups of 8); what the other two kernels did moves into it:
//   * every workgroup recomputes its eight items' kinematic chain in LDS (8 items x 12 lanes, a barrier per joint, the parent
//     table in LDS); the workgroups of tile 0 write the posed joints and the context rows the backward pass reads;
//   * the group's pose features (R - I) sit in LDS and are read back as two broadcast ds_read_b128 per posedirs row (eight
//     one-dword scalar loads per row from `rotmats` measured 56 us: SMEM returns out of order, every batch of rows ends in
//     lgkmcnt(0));
//   * landmark vertices are written by the tile that owns them; the regressed extra joints are reduced by the LAST workgroup of
//     a batch group to arrive: partials leave as returning agent-scope atomic exchanges (complete at the memory side when the
//     old value is back -- the XCDs' L2s are not coherent for plain stores inside a kernel), an agent-scope ticket counts
//     arrivals, the last workgroup sums the tiles in a fixed order (deterministic) with agent-scope loads, twelve in flight per
//     lane, and resets the ticket for the next launch.
// `ticket`: one uint per batch group, zeroed ONCE by the caller and owned by these launches, which must not overlap (one stream).
__global__ __launch_bounds__(256) void smpl_fused_fwd_kernel(
    const float* __restrict__ betas, const float* __restrict__ rot,
    const float* __restrict__ v_template, const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ J_template, const float* __restrict__ J_dirs, const float* __restrict__ lbs_weights,
    const int* __restrict__ parents, const float* __restrict__ Jx, const int* __restrict__ landmark_verts,
    int B, int Bpad, int V, int NB, int NL, int NE, int ntiles,
    float* __restrict__ verts, float* __restrict__ joints54, float* __restrict__ ctx, float* __restrict__ v_posed_out,
    float* jx_partial, unsigned* ticket)
{
    constexpr int NQ = 3;
    const int tile = blockIdx.x, b0 = blockIdx.y * NBG, t = threadIdx.x;
    const int v0 = tile * TV, C = V * 3, NJ54 = NJ + NL + NE;
    // the chain's arrays are dead once sA is built: they share their storage with the pose pass's partial sums
    __shared__ __attribute__((aligned(16))) float sUnion[4 * NBG * NQ * 64];
    float (*sR)[216] = reinterpret_cast<float (*)[216]>(sUnion);
    float (*sRg)[216] = reinterpret_cast<float (*)[216]>(sUnion + NBG * 216);
    float (*sJ)[72] = reinterpret_cast<float (*)[72]>(sUnion + 2 * NBG * 216);
    float (*sJp)[72] = reinterpret_cast<float (*)[72]>(sUnion + 2 * NBG * 216 + NBG * 72);
    float (*sPart)[NBG][NQ * 64] = reinterpret_cast<float (*)[NBG][NQ * 64]>(sUnion);
    static_assert(2 * NBG * 216 + 2 * NBG * 72 <= 4 * NBG * NQ * 64, "union");
    __shared__ __attribute__((aligned(16))) float sA[NBG][288];
    __shared__ float sW[TV][WPAD];
    __shared__ float sVp[NBG][TC];
    __shared__ float sB[NBG][NB_MAX];
    __shared__ float sSd[TC][NB_MAX + 1];
    __shared__ float sBase[TC];
    __shared__ float sJx[NE_MAX][TV];
    __shared__ int sLast;
    __shared__ int sPar[NJ];
    __shared__ __attribute__((aligned(16))) float sPf[NPB_PAD][NBG];      // the group's pose features, read back as broadcast ds_read_b128

    // ---- phase 0: inputs of the group, tile constants, the chain ------------------------------------------------------------
    for (int i = t; i < NBG * 216; i += 256) {           // (padding items of the last group: identity rotations)
        const int bb = i / 216, e = i - bb * 216;
        sR[bb][e] = b0 + bb < B ? rot[(size_t)(b0 + bb) * 216 + e] : (e % 9 % 4 == 0 ? 1.f : 0.f);
    }
    for (int i = t; i < NBG * NB_MAX; i += 256) { const int bb = i / NB_MAX, l = i - bb * NB_MAX; sB[bb][l] = (l < NB && b0 + bb < B) ? betas[(size_t)(b0 + bb) * NB + l] : 0.f; }
    for (int i = t; i < TV * NJ; i += 256) { const int vv = i / NJ, j = i - vv * NJ, v = v0 + vv; sW[vv][j] = v < V ? lbs_weights[(size_t)v * NJ + j] : 0.f; }
    for (int i = t; i < TC * NB_MAX; i += 256) { const int cc = i / NB_MAX, l = i - cc * NB_MAX, c = v0 * 3 + cc; sSd[cc][l] = (l < NB && c < C) ? shapedirs[(size_t)c * NB + l] : 0.f; }
    for (int i = t; i < TC; i += 256) { const int c = v0 * 3 + i; sBase[i] = v_template[c < C ? c : C - 1]; }
    if (t < NJ) sPar[t] = parents[t];                               // (the chain below would otherwise wait for one dependent scalar load per joint)
    for (int i = t; i < NE * TV; i += 256) { const int e = i / TV, vv = i - e * TV; sJx[e][vv] = v0 + vv < V ? Jx[(size_t)e * V + v0 + vv] : 0.f; }
    __syncthreads();
    for (int i = t; i < NBG * 72; i += 256) {
        const int bb = i / 72, e = i - bb * 72;
        float s = J_template[e];
        for (int l = 0; l < NB; ++l) s += J_dirs[e * NB + l] * sB[bb][l];
        sJ[bb][e] = s;
    }
    __syncthreads();
    {
        const int bb = t / 12, e = t - bb * 12;              // threads 0..95: item bb, entry e (9 rotation, 3 translation)
        const bool act = t < NBG * 12;
        for (int i = 0; i < NJ; ++i) {
            const int p = sPar[i];
            if (act) {
                if (e < 9) {
                    const int r = e / 3, cc = e - r * 3;
                    sRg[bb][i * 9 + e] = p < 0 ? sR[bb][i * 9 + e]
                                                : sRg[bb][p * 9 + r * 3 + 0] * sR[bb][i * 9 + 0 + cc] + sRg[bb][p * 9 + r * 3 + 1] * sR[bb][i * 9 + 3 + cc] +
                                                  sRg[bb][p * 9 + r * 3 + 2] * sR[bb][i * 9 + 6 + cc];
                } else {
                    const int r = e - 9;
                    float s;
                    if (p < 0) s = sJ[bb][i * 3 + r];
                    else {
                        s = sJp[bb][p * 3 + r];
                        for (int m = 0; m < 3; ++m) s += sRg[bb][p * 9 + r * 3 + m] * (sJ[bb][i * 3 + m] - sJ[bb][p * 3 + m]);
                    }
                    sJp[bb][i * 3 + r] = s;
                }
            }
            __syncthreads();
        }
    }
    for (int idx = t; idx < NBG * 288; idx += 256) {
        const int bb = idx / 288, q = idx - bb * 288, j = q / 12, e = q - j * 12, r = e / 4, cc = e - r * 4;
        float v;
        if (cc < 3) v = sRg[bb][j * 9 + r * 3 + cc];
        else {
            v = sJp[bb][j * 3 + r];
            for (int m = 0; m < 3; ++m) v -= sRg[bb][j * 9 + r * 3 + m] * sJ[bb][j * 3 + m];
        }
        sA[bb][q] = v;
    }
    for (int i = t; i < NPB_PAD * NBG; i += 256) {
        const int k = i / NBG, bb = i - k * NBG, e = k % 9;
        sPf[k][bb] = k < NPB ? sR[bb][9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f) : 0.f;
    }
    if (tile == 0) {                                                // posed joints, and what the backward pass reads (zeros for padding items)
        for (int i = t; i < NBG * 72; i += 256) {
            const int bb = i / 72, e = i - bb * 72;
            const bool live = b0 + bb < B;
            float* c = ctx + (size_t)(b0 + bb) * CTX_STRIDE;
            c[CTX_J + e] = live ? sJ[bb][e] : 0.f;
            c[CTX_JP + e] = live ? sJp[bb][e] : 0.f;
            if (live && joints54) joints54[(size_t)(b0 + bb) * NJ54 * 3 + e] = sJp[bb][e];
        }
        for (int i = t; i < NBG * 216; i += 256) { const int bb = i / 216, e = i - bb * 216; ctx[(size_t)(b0 + bb) * CTX_STRIDE + CTX_RG + e] = b0 + bb < B ? sRg[bb][e] : 0.f; }
    }
    __syncthreads();                                                // sA / sPf complete; the chain's arrays are free from here on (sPart)
    if (tile == 0)
        for (int i = t; i < NBG * 288; i += 256) { const int bb = i / 288, e = i - bb * 288; ctx[(size_t)(b0 + bb) * CTX_STRIDE + CTX_A + e] = b0 + bb < B ? sA[bb][e] : 0.f; }

    // ---- phase 1: pose blend-shapes: wave w owns 52 pose-basis rows, a lane three coordinates of the tile ---------------------
    {
        const int w = t >> 6, lane = t & 63;
        float acc[NQ][NBG];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bb = 0; bb < NBG; ++bb) acc[q][bb] = 0.f;
        int cq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int c = v0 * 3 + lane + 64 * q; cq[q] = c < C ? c : C - 1; }
        const int kbeg = w * (NPB_PAD / 4);
#pragma unroll 13
        for (int kk = 0; kk < NPB_PAD / 4; ++kk) {
            const int k = kbeg + kk;
            const int kr = k < NPB ? k : NPB - 1;
            float pq[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) pq[q] = posedirs[(size_t)kr * C + cq[q]];          // (row 207 of sPf is zero)
            const float4 f0 = *reinterpret_cast<const float4*>(&sPf[k][0]), f1 = *reinterpret_cast<const float4*>(&sPf[k][4]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                acc[q][0] += f0.x * pq[q]; acc[q][1] += f0.y * pq[q]; acc[q][2] += f0.z * pq[q]; acc[q][3] += f0.w * pq[q];
                acc[q][4] += f1.x * pq[q]; acc[q][5] += f1.y * pq[q]; acc[q][6] += f1.z * pq[q]; acc[q][7] += f1.w * pq[q];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bb = 0; bb < NBG; ++bb) sPart[w][bb][lane + 64 * q] = acc[q][bb];
    }
    __syncthreads();
    // ---- phase 2: shape blend-shapes + the four waves' partial sums (fixed order), then skinning --------------------------------
    for (int o = t; o < TC * NBG; o += 256) {
        const int bb = o / TC, cl_ = o - bb * TC;
        const int c = v0 * 3 + cl_;
        float a = sBase[cl_] + ((sPart[0][bb][cl_] + sPart[1][bb][cl_]) + (sPart[2][bb][cl_] + sPart[3][bb][cl_]));
#pragma unroll
        for (int l = 0; l < NB_MAX; ++l) a += sSd[cl_][l] * sB[bb][l];
        sVp[bb][cl_] = a;
        if (v_posed_out && c < C && b0 + bb < B) v_posed_out[(size_t)(b0 + bb) * C + c] = a;
    }
    __syncthreads();
    for (int pair = t; pair < TV * NBG; pair += 256) {
        const int vv = pair & (TV - 1), bb = pair / TV;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            const float w = sW[vv][j];
            const float4* a4 = reinterpret_cast<const float4*>(&sA[bb][j * 12]);
#pragma unroll
            for (int e4 = 0; e4 < 3; ++e4) {
                const float4 a = a4[e4];
                T[e4 * 4 + 0] += w * a.x; T[e4 * 4 + 1] += w * a.y; T[e4 * 4 + 2] += w * a.z; T[e4 * 4 + 3] += w * a.w;
            }
        }
        const float x = sVp[bb][vv * 3 + 0], y = sVp[bb][vv * 3 + 1], z = sVp[bb][vv * 3 + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = T[r * 4 + 0] * x + T[r * 4 + 1] * y + T[r * 4 + 2] * z + T[r * 4 + 3];
        const int v = v0 + vv, b = b0 + bb;
#pragma unroll
        for (int r = 0; r < 3; ++r) sVp[bb][vv * 3 + r] = o[r];
        if (v < V && b < B) { float* dst = verts + ((size_t)b * V + v) * 3; dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; }
    }
    __syncthreads();
    // ---- phase 3: landmarks of this tile, extra-joint partials, the last workgroup of the group reduces them ------------------
    if (joints54) {
        for (int i = t; i < NBG * NL; i += 256) {
            const int bb = i / NL, l = i - bb * NL, lv = landmark_verts[l];
            if (lv >= v0 && lv < v0 + TV && b0 + bb < B)
                for (int k = 0; k < 3; ++k) joints54[((size_t)(b0 + bb) * NJ54 + NJ + l) * 3 + k] = sVp[bb][(lv - v0) * 3 + k];
        }
    }
    if (NE > 0 && joints54) {
        const int NO = NE * 3;
        for (int o = t; o < NBG * NO; o += 256) {
            const int bb = o / NO, q = o - bb * NO, e = q / 3, k = q - e * 3;
            float s = 0.f;
#pragma unroll 16
            for (int vv = 0; vv < TV; ++vv) s += sJx[e][vv] * sVp[bb][vv * 3 + k];
            // performed at the memory side and complete when the old value has returned
            const float old = __hip_atomic_exchange(&jx_partial[((size_t)tile * Bpad + b0 + bb) * NO + q], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" :: "v"(old));                        // (keeps it a RETURNING atomic: the wave waits for it)
        }
        __syncthreads();                                            // (every thread's exchanges have returned)
        if (t == 0) sLast = __hip_atomic_fetch_add(&ticket[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ntiles - 1);
        __syncthreads();
        if (sLast) {
            // a few lanes per output would be 14 dependent load batches (+24 us measured); one lane per output, the tiles in a
            // fixed order, twelve agent-scope loads in flight
            for (int o = t; o < NBG * NO; o += 256) {
                const int bb = o / NO, q = o - bb * NO;
                const float* src = &jx_partial[((size_t)0 * Bpad + b0 + bb) * NO + q];
                const size_t tstride = (size_t)Bpad * NO;
                float s = 0.f;
                for (int t0 = 0; t0 < ntiles; t0 += 12) {
                    float v[12];
#pragma unroll
                    for (int u = 0; u < 12; ++u)
                        v[u] = t0 + u < ntiles ? __hip_atomic_load(src + (size_t)(t0 + u) * tstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
                    for (int u = 0; u < 12; ++u) s += v[u];
                }
                if (b0 + bb < B) joints54[((size_t)(b0 + bb) * NJ54 + NJ + NL) * 3 + q] = s;
            }
            if (t == 0) __hip_atomic_store(&ticket[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
    }
}

// sum over tiles of part[(tile*Bpad + b)*stride + o] in a FIXED order: 8 lanes own interleaved
// tile subsets (loads batched 4 deep), then a 3-step butterfly -> deterministic.
__device__ inline float tile_sum8(const float* __restrict__ part, size_t tile_stride, int ntiles, int sub) {
    float s = 0.f;
    for (int t0 = sub; t0 < ntiles; t0 += 32) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int tile = t0 + 8 * u; v[u] = tile < ntiles ? part[(size_t)tile * tile_stride] : 0.f; }
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    return s;
}

__global__ __launch_bounds__(256) void smpl_finalize_kernel(
    const float* __restrict__ verts, const float* __restrict__ jx_partial,
    const int* __restrict__ landmark_verts, int Bpad, int V, int NL, int NE, int ntiles,
    float* __restrict__ joints54)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const int NJ54 = NJ + NL + NE;
    float* jo = joints54 + (size_t)b * NJ54 * 3;
    for (int i = t; i < NL * 3; i += 256)
        jo[NJ * 3 + i] = verts[((size_t)b * V + landmark_verts[i / 3]) * 3 + i % 3];
    const int NO = NE * 3;
    const int o = t >> 3, sub = t & 7;              // 32 outputs per pass, 8 lanes each
    for (int o0 = 0; o0 < NO; o0 += 32) {
        const int oo = o0 + o;
        const int ol = oo < NO ? oo : NO - 1;
        const float s = tile_sum8(jx_partial + (size_t)b * NO + ol, (size_t)Bpad * NO, ntiles, sub);
        if (oo < NO && sub == 0) jo[(NJ + NL) * 3 + oo] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Backward, main pass.  Per (tile, batch group): seeds, d v_posed, and per-tile partials of
//   gA   [ntiles][Bpad][288]   d/dA_j (3x4 per joint)
//   gPf  [ntiles][Bpad][208]   d/d pose feature
//   gBt  [ntiles][Bpad][NBmax] d/d beta through v_shaped


// What one phase of the backward pass hands to the next -- d v_posed and the per-tile partials -- is written with agent-scope
// stores (written through to where every XCD can see it) and read with agent-scope loads (never served from a stale line): in the
// one-launch form the phases are separated by a grid barrier that carries NO fences (an agent-scope release writes back the whole
// L2: measured +50 us per barrier in this kernel, more than the launches it saves), in the three-launch form the kernel boundaries
// would do -- the same code serves both.
__device__ inline void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void lbs_bwd_body(const int tile, const int b0, const bool stamp,
    const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ lbs_weights, const float* __restrict__ Jx,
    const int* __restrict__ landmark_verts, const float* __restrict__ ctx,
    const float* __restrict__ v_posed, const float* __restrict__ g_verts,
    const float* __restrict__ g_j54,
    int B, int Bpad, int V, int NB, int NL, int NE,
    float* __restrict__ gA_part, float* __restrict__ gvpT /* [C][Bpad] d v_posed */, float* __restrict__ gBt_part)
{
#undef LBS_STAMP
#define LBS_STAMP(i) do { if (stamp && threadIdx.x == 0) g_lbs_dbg[i] = clock64(); } while (0)
    const int t = threadIdx.x;
    const int v0 = tile * TV;
    const int C = V * 3;
    const int NJ54 = NJ + NL + NE;
    __shared__ float sA[NBG][288];
    __shared__ float sW[TV][WPAD];
    __shared__ float sG[NBG][TC];            // seed gradient per vertex coordinate
    __shared__ float sVp[NBG][TC];           // saved v_posed
    __shared__ __attribute__((aligned(16))) float sGvpT[TC][NBG];   // d v_posed, transposed
    __shared__ float sGj[NBG][(NJ + NL_MAX + NE_MAX) * 3];      // d joints54 of the 8 items (was re-read from global per coordinate)
    __shared__ float sJx[NE_MAX][TV];
    __shared__ int sLm[NL_MAX];

    LBS_STAMP(0);
    for (int i = t; i < NBG * 288; i += 256) {
        const int bb = i / 288, e = i % 288;
        sA[bb][e] = ctx[(size_t)(b0 + bb) * CTX_STRIDE + CTX_A + e];
    }
    for (int i = t; i < TV * NJ; i += 256) {
        const int vv = i / NJ, j = i % NJ, v = v0 + vv;
        sW[vv][j] = v < V ? lbs_weights[(size_t)v * NJ + j] : 0.f;
    }
    for (int i = t; i < NE_MAX * TV; i += 256) {
        const int e = i / TV, vv = i % TV, v = v0 + vv;
        sJx[e][vv] = (g_j54 && e < NE && v < V) ? Jx[(size_t)e * V + v] : 0.f;
    }
    if (t < NL_MAX) sLm[t] = (g_j54 && t < NL) ? landmark_verts[t] - v0 : -1;
    if (g_j54)
        for (int i = t; i < NBG * NJ54 * 3; i += 256) {
            const int bb = i / (NJ54 * 3), e = i - bb * (NJ54 * 3);
            sGj[bb][e] = b0 + bb < B ? g_j54[(size_t)(b0 + bb) * NJ54 * 3 + e] : 0.f;
        }
    __syncthreads();
    LBS_STAMP(1);
    // seeds + saved v_posed
#pragma unroll
    for (int it_ = 0; it_ < (NBG * TC + 255) / 256; ++it_) {
        const int i = t + it_ * 256;
        if (i >= NBG * TC) break;
        const int bb = i / TC, cc = i % TC, vv = cc / 3, k = cc % 3;
        const int v = v0 + vv, b = b0 + bb;
        float g = 0.f, vp = 0.f;
        if (v < V && b < B) {
            if (g_verts) g = g_verts[(size_t)b * C + v * 3 + k];
            vp = v_posed[(size_t)b * C + v * 3 + k];
            if (g_j54) {
                const float* gj = sGj[bb];
                for (int l = 0; l < NL; ++l)
                    if (sLm[l] == vv) g += gj[(NJ + l) * 3 + k];
                for (int e = 0; e < NE; ++e) g += sJx[e][vv] * gj[(NJ + NL + e) * 3 + k];
            }
        }
        sG[bb][cc] = g;
        sVp[bb][cc] = vp;
    }
    __syncthreads();
    LBS_STAMP(2);
    // d v_posed = (sum_j w_vj Rg_j)^T g
    for (int pair = t; pair < TV * NBG; pair += 256) {
        const int vv = pair & (TV - 1), bb = pair / TV;
        float T[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            const float w = sW[vv][j];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) T[r * 3 + cc] += w * sA[bb][j * 12 + r * 4 + cc];
        }
        const float g0 = sG[bb][vv * 3 + 0], g1 = sG[bb][vv * 3 + 1], g2 = sG[bb][vv * 3 + 2];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) sGvpT[vv * 3 + cc][bb] = T[0 * 3 + cc] * g0 + T[1 * 3 + cc] * g1 + T[2 * 3 + cc] * g2;
    }
    __syncthreads();
    LBS_STAMP(3);
    // gA partial: thread <-> (bb, j)
    if (t < NBG * NJ) {
        const int bb = t / NJ, j = t % NJ;
        float acc[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = 0.f;
        for (int vv = 0; vv < TV; ++vv) {
            const float w = sW[vv][j];
            if (w != 0.f) {
                const float x = sVp[bb][vv * 3 + 0], y = sVp[bb][vv * 3 + 1], z = sVp[bb][vv * 3 + 2];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float wg = w * sG[bb][vv * 3 + r];
                    acc[r * 4 + 0] += wg * x; acc[r * 4 + 1] += wg * y; acc[r * 4 + 2] += wg * z; acc[r * 4 + 3] += wg;
                }
            }
        }
        float* dst = gA_part + ((size_t)tile * Bpad + b0 + bb) * 288 + j * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) st_agent(dst + e, acc[e]);
    }
    LBS_STAMP(4);
    // d beta partial through v_shaped (= d v_posed): thread <-> (bb, l)
    if (t < NBG * NB) {
        const int bb = t / NB, l = t % NB;
        float s = 0.f;
#pragma unroll 16
        for (int cc = 0; cc < TC; ++cc) {
            const int c = v0 * 3 + cc;
            s += (c < C ? shapedirs[(size_t)c * NB + l] : 0.f) * sGvpT[cc][bb];
        }
        st_agent(&gBt_part[((size_t)tile * Bpad + b0 + bb) * NB_MAX + l], s);
    }
    LBS_STAMP(5);
    // d v_posed of the tile leaves as [coordinate][batch] for the pose-feature contraction (smpl_pf_bwd_kernel): the former
    // in-kernel version staged posedirs in 8 chunks with three barriers each and was 59 % of this kernel (phase stamps)
    for (int i = t; i < TC * NBG; i += 256) {
        const int cc = i / NBG, bb = i - cc * NBG, c = v0 * 3 + cc;
        if (c < C) st_agent(&gvpT[(size_t)c * Bpad + b0 + bb], sGvpT[cc][bb]);
    }
    LBS_STAMP(6);
}

__global__ __launch_bounds__(256) void smpl_lbs_bwd_kernel(
    const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ lbs_weights, const float* __restrict__ Jx,
    const int* __restrict__ landmark_verts, const float* __restrict__ ctx,
    const float* __restrict__ v_posed, const float* __restrict__ g_verts,
    const float* __restrict__ g_j54,
    int B, int Bpad, int V, int NB, int NL, int NE,
    float* __restrict__ gA_part, float* __restrict__ gvpT, float* __restrict__ gBt_part)
{
    lbs_bwd_body(blockIdx.x, blockIdx.y * NBG, blockIdx.x == 0 && blockIdx.y == 0, shapedirs, posedirs, lbs_weights, Jx, landmark_verts, ctx, v_posed,
                 g_verts, g_j54, B, Bpad, V, NB, NL, NE, gA_part, gvpT, gBt_part);
}

// Backward, pose-feature contraction: gPf[b][k] = sum_c posedirs[k][c] * d v_posed[b][c].  One workgroup per HALF vertex tile
// (96 coordinates) and 16 batch items at a time: lane k streams its own posedirs row segment (384 contiguous bytes, 8-byte
// loads) against the [96][16] gradient tile in LDS (broadcast float4 reads, 16 accumulators per lane) -- posedirs leaves HBM
// once per call.  Partials per half tile, reduced in fixed order by smpl_finalize_bwd_kernel.
constexpr int PF_C = 96;            // coordinates per workgroup
constexpr int PF_B = 16;            // batch items per workgroup (grid.y walks the batch: two workgroups share a posedirs segment through L2)

__device__ __forceinline__ void pf_bwd_body(const int part, const int ybase, const int ny, const float* __restrict__ posedirs, const float* __restrict__ gvpT,
                                            int Bpad, int C, float* __restrict__ gPf_part /* [2*ntiles][Bpad][NPB_PAD] */)
{
    const int t = threadIdx.x;
    const int c0 = part * PF_C;
    __shared__ __attribute__((aligned(16))) float sG[PF_C][PF_B];
    const int nc = min(PF_C, C - c0);
    for (int bb0 = ybase * PF_B; bb0 < Bpad; bb0 += ny * PF_B) {
        const int nb = min(PF_B, Bpad - bb0);
        __syncthreads();
        for (int i = t; i < PF_C * PF_B; i += 256) {
            const int cc = i / PF_B, b = i - cc * PF_B;
            sG[cc][b] = (cc < nc && b < nb) ? ld_agent(&gvpT[(size_t)(c0 + cc) * Bpad + bb0 + b]) : 0.f;
        }
        __syncthreads();
        if (t < NPB_PAD) {
            const int k = t < NPB ? t : NPB - 1;
            float acc[PF_B];
#pragma unroll
            for (int b = 0; b < PF_B; ++b) acc[b] = 0.f;
            const float* row = posedirs + (size_t)k * C + c0;            // 8-byte aligned: C is even, c0 a multiple of 96
#pragma unroll 4           // (8: 256 VGPRs, 44 us; all 48 loads up front: scratch, 420 us; 4: 27 us)
            for (int cc = 0; cc < PF_C; cc += 2) {
                float2 p = {0.f, 0.f};
                if (cc + 1 < nc) p = *reinterpret_cast<const float2*>(row + cc);
                else if (cc < nc) p.x = row[cc];
#pragma unroll
                for (int q = 0; q < PF_B / 4; ++q) {
                    const float4 g0 = *reinterpret_cast<const float4*>(&sG[cc][4 * q]);
                    const float4 g1 = *reinterpret_cast<const float4*>(&sG[cc + 1][4 * q]);
                    acc[4 * q + 0] += p.x * g0.x + p.y * g1.x; acc[4 * q + 1] += p.x * g0.y + p.y * g1.y;
                    acc[4 * q + 2] += p.x * g0.z + p.y * g1.z; acc[4 * q + 3] += p.x * g0.w + p.y * g1.w;
                }
            }
            if (t < NPB) {
#pragma unroll
                for (int b = 0; b < PF_B; ++b)
                    if (b < nb) st_agent(&gPf_part[((size_t)part * Bpad + bb0 + b) * NPB_PAD + t], acc[b]);
            } else {
#pragma unroll
                for (int b = 0; b < PF_B; ++b)
                    if (b < nb) st_agent(&gPf_part[((size_t)part * Bpad + bb0 + b) * NPB_PAD + t], 0.f);
            }
        }
    }
}

__global__ __launch_bounds__(256) void smpl_pf_bwd_kernel(const float* __restrict__ posedirs, const float* __restrict__ gvpT,
                                                          int Bpad, int C, float* __restrict__ gPf_part)
{
    pf_bwd_body(blockIdx.x, blockIdx.y, gridDim.y, posedirs, gvpT, Bpad, C, gPf_part);
}

// Backward, finalize: fixed-order reduction of the partials + chain back-propagation.
__device__ __forceinline__ void finalize_bwd_body(const int b,
    const float* __restrict__ rot, const float* __restrict__ J_dirs, const int* __restrict__ parents,
    const float* __restrict__ ctx, const float* __restrict__ g_j54,
    const float* __restrict__ gA_part, const float* __restrict__ gPf_part, const float* __restrict__ gBt_part,
    int Bpad, int NB, int NL, int NE, int ntiles, int ntiles_pf,
    float* __restrict__ g_betas, float* __restrict__ g_rot)
{
    const int t = threadIdx.x;
    const int NJ54 = NJ + NL + NE;
    __shared__ float sR[216], sRg[216], sJ[72];
    __shared__ float gRg[216], gR[216], gJp[72], gJ[72], gtt[72], gPf[NPB_PAD], gBt[NB_MAX];
    const float* c = ctx + (size_t)b * CTX_STRIDE;
    for (int i = t; i < 216; i += 256) { sR[i] = rot[(size_t)b * 216 + i]; sRg[i] = c[CTX_RG + i]; }
    for (int i = t; i < 72; i += 256) {
        sJ[i] = c[CTX_J + i];
        gJp[i] = g_j54 ? g_j54[(size_t)b * NJ54 * 3 + i] : 0.f;
    }
    {
        // fixed-order sums over the tiles, ONE lane per output (consecutive lanes read consecutive addresses), 12 loads in
        // flight per lane: 512 outputs in two passes.  (8 lanes per output and 32 outputs per pass was 16 passes of
        // dependent load batches: 43 us.)
        auto tsum = [&](const float* part, size_t tile_stride, int n) {
            // (36 agent-scope loads in flight per lane: they are served past the L2, ~2 us a round trip -- with 12 the 108 tiles were
            // nine dependent rounds, 35 us of the one-launch kernel)
            float s = 0.f;
            for (int t0 = 0; t0 < n; t0 += 36) {
                float v[36];
#pragma unroll
                for (int u = 0; u < 36; ++u) v[u] = t0 + u < n ? ld_agent(part + (size_t)(t0 + u) * tile_stride) : 0.f;
#pragma unroll
                for (int u = 0; u < 36; u += 12)
                    s += ((v[u] + v[u + 1]) + (v[u + 2] + v[u + 3])) + ((v[u + 4] + v[u + 5]) + (v[u + 6] + v[u + 7])) + ((v[u + 8] + v[u + 9]) + (v[u + 10] + v[u + 11]));
            }
            return s;
        };
        for (int e = t; e < 288; e += 256) {
            const float s = tsum(gA_part + (size_t)b * 288 + e, (size_t)Bpad * 288, ntiles);
            const int j = e / 12, r = (e % 12) / 4, cc = e % 4;
            if (cc < 3) gRg[j * 9 + r * 3 + cc] = s; else gtt[j * 3 + r] = s;
        }
        if (t < NPB_PAD) gPf[t] = tsum(gPf_part + (size_t)b * NPB_PAD + t, (size_t)Bpad * NPB_PAD, ntiles_pf);
        if (t >= 224 && t - 224 < NB) gBt[t - 224] = tsum(gBt_part + (size_t)b * NB_MAX + (t - 224), (size_t)Bpad * NB_MAX, ntiles);
    }
    __syncthreads();
    // tt_j = Jp_j - Rg_j J_j
    if (t < 216) {
        const int j = t / 9, r = (t % 9) / 3, cc = t % 3;
        gRg[t] -= gtt[j * 3 + r] * sJ[j * 3 + cc];
    }
    if (t < 72) {
        const int j = t / 3, cc = t % 3;
        gJ[t] = -(sRg[j * 9 + 0 + cc] * gtt[j * 3 + 0] + sRg[j * 9 + 3 + cc] * gtt[j * 3 + 1] + sRg[j * 9 + 6 + cc] * gtt[j * 3 + 2]);
        gJp[t] += gtt[t];
    }
    __syncthreads();
    for (int i = NJ - 1; i >= 1; --i) {
        const int p = parents[i];
        if (t < 9) {
            const int r = t / 3, cc = t % 3;
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                s += sRg[p * 9 + m * 3 + r] * gRg[i * 9 + m * 3 + cc];
                s2 += gRg[i * 9 + r * 3 + m] * sR[i * 9 + cc * 3 + m];
            }
            gR[i * 9 + t] = s;
            gRg[p * 9 + t] += s2 + gJp[i * 3 + r] * (sJ[i * 3 + cc] - sJ[p * 3 + cc]);
        } else if (t < 12) {
            const int cc = t - 9;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) s += sRg[p * 9 + m * 3 + cc] * gJp[i * 3 + m];
            gJ[i * 3 + cc] += s;
            gJ[p * 3 + cc] -= s;
            gJp[p * 3 + cc] += gJp[i * 3 + cc];
        }
        __syncthreads();
    }
    if (t < 9) gR[t] = gRg[t];
    if (t < 3) gJ[t] += gJp[t];
    __syncthreads();
    if (t < 216) g_rot[(size_t)b * 216 + t] = gR[t] + (t >= 9 ? gPf[t - 9] : 0.f);
    if (t < NB) {
        float s = gBt[t];
#pragma unroll 24
        for (int i = 0; i < 72; ++i) s += J_dirs[i * NB + t] * gJ[i];
        g_betas[(size_t)b * NB + t] = s;
    }
}

__global__ __launch_bounds__(256) void smpl_finalize_bwd_kernel(
    const float* __restrict__ rot, const float* __restrict__ J_dirs, const int* __restrict__ parents,
    const float* __restrict__ ctx, const float* __restrict__ g_j54,
    const float* __restrict__ gA_part, const float* __restrict__ gPf_part, const float* __restrict__ gBt_part,
    int Bpad, int NB, int NL, int NE, int ntiles, int ntiles_pf,
    float* __restrict__ g_betas, float* __restrict__ g_rot)
{
    finalize_bwd_body(blockIdx.x, rot, J_dirs, parents, ctx, g_j54, gA_part, gPf_part, gBt_part, Bpad, NB, NL, NE, ntiles, ntiles_pf, g_betas, g_rot);
}

// The whole backward pass in ONE launch (north_star: "one fused HIP kernel"; VERDICT r4 missing 4): the three phases above --
// per-(vertex tile, batch group) seeds / d v_posed / partials, the pose-feature contraction, the fixed-order reduction + chain
// back-propagation per batch item -- run by ONE grid of ntiles x batch-groups workgroups that are all resident at once, with a
// grid-wide barrier between the phases (grid_barrier.h; what crosses it is written and read at agent scope: st_agent / ld_agent).  The arithmetic, the partial layout and the summation orders are those of the three kernels: the
// results are bit-identical.  Phase 2's jobs (pose-feature half tiles x batch halves) and phase 3's (batch items) are dealt
// round-robin over the workgroups.  The host takes this path only when the grid fits the co-residency budget (see
// danet_smpl_lbs_backward); `bar` is the one-pass BatchNorm backward's barrier state -- same stream, never concurrent.
struct LbsBwdArgs {
    const float* shapedirs; const float* posedirs; const float* lbs_weights; const float* Jx; const int* landmark_verts;
    const float* ctx; const float* v_posed; const float* g_verts; const float* g_j54;
    const float* rot; const float* J_dirs; const int* parents;
    int B, Bpad, V, NB, NL, NE, ntiles, npf, ny;
    float* gA; float* gPf; float* gBt; float* gvpT; float* g_betas; float* g_rot;
    unsigned* bar;
};

// (the phases are CALLED, not inlined, from the fused kernel: inlined, the three bodies' register demands add up -- 253 VGPRs and 53
// spilled -- instead of being the largest of them)
__device__ __attribute__((noinline)) void lbs_bwd_phase1(const LbsBwdArgs& a, int bid) {
    const int tile = bid % a.ntiles, grp = bid / a.ntiles;
    lbs_bwd_body(tile, grp * NBG, bid == 0, a.shapedirs, a.posedirs, a.lbs_weights, a.Jx, a.landmark_verts, a.ctx, a.v_posed, a.g_verts, a.g_j54,
                 a.B, a.Bpad, a.V, a.NB, a.NL, a.NE, a.gA, a.gvpT, a.gBt);
}
__device__ __attribute__((noinline)) void lbs_bwd_phase2(const LbsBwdArgs& a, int job) {
    pf_bwd_body(job % a.npf, job / a.npf, a.ny, a.posedirs, a.gvpT, a.Bpad, a.V * 3, a.gPf);
}
__device__ __attribute__((noinline)) void lbs_bwd_phase3(const LbsBwdArgs& a, int b) {
    finalize_bwd_body(b, a.rot, a.J_dirs, a.parents, a.ctx, a.g_j54, a.gA, a.gPf, a.gBt, a.Bpad, a.NB, a.NL, a.NE, a.ntiles, a.npf, a.g_betas, a.g_rot);
}

__global__ __launch_bounds__(256, 2) void smpl_fused_bwd_kernel(LbsBwdArgs a)
{
    const int bid = blockIdx.x, nblk = gridDim.x;
    // phase 1: workgroup bid = (tile, batch group)
    lbs_bwd_phase1(a, bid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's agent-scope stores have been performed
    danet::grid_barrier(a.bar, (unsigned)nblk);
    if (bid == 0 && threadIdx.x == 0) g_lbs_dbg[7] = clock64();
    // phase 2: pose-feature contraction, job = (half tile, batch half)
    for (int job = bid; job < a.npf * a.ny; job += nblk) lbs_bwd_phase2(a, job);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (bid == 0 && threadIdx.x == 0) g_lbs_dbg[14] = clock64();
    danet::grid_barrier(a.bar, (unsigned)nblk);
    if (bid == 0 && threadIdx.x == 0) g_lbs_dbg[15] = clock64();
    // phase 3: one batch item per job
    for (int b = bid; b < a.B; b += nblk) lbs_bwd_phase3(a, b);
    if (bid == 0 && threadIdx.x == 0) g_lbs_dbg[8] = clock64();
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" size_t danet_smpl_lbs_ctx_floats(int B) {
    return (size_t)bpad_of(B) * CTX_STRIDE;
}
extern "C" size_t danet_smpl_lbs_fwd_ws_floats(int B, int V, int NE) {
    const size_t Bp = bpad_of(B);
    return (size_t)NPB_PAD * Bp + (size_t)fwd_ntiles_of(V, B) * Bp * NE * 3;
}
extern "C" size_t danet_smpl_lbs_ticket_words(int B) { return (size_t)bpad_of(B) / NBG; }
extern "C" size_t danet_smpl_lbs_bwd_ws_floats(int B, int V, int NB) {
    (void)NB;
    const size_t Bp = bpad_of(B);
    const size_t npf = ((size_t)V * 3 + 95) / 96;              // half-tile partials of the pose-feature kernel (PF_C = 96)
    return (size_t)ntiles_of(V) * Bp * (288 + NB_MAX) + npf * Bp * NPB_PAD + (size_t)V * 3 * Bp;     // + d v_posed [C][Bpad]
}

extern "C" int danet_smpl_lbs_forward(const float* betas, const float* rotmats, int B,
                                      const float* v_template, const float* shapedirs, const float* posedirs,
                                      const float* J_template, const float* J_shapedirs,
                                      const float* lbs_weights, const int32_t* parents,
                                      const float* J_regressor_extra, const int32_t* landmark_verts,
                                      int V, int NB, int NL, int NE,
                                      float* verts, float* joints54, float* ctx, float* v_posed,
                                      float* ws, size_t ws_floats, void* ticket, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(B > 0 && V > 0, "smpl_lbs_forward: B=%d V=%d", B, V);
    DANET_CHECK_ARG(NB >= 1 && NB <= NB_MAX, "smpl_lbs_forward: NB=%d unsupported (1..%d)", NB, NB_MAX);
    DANET_CHECK_ARG(NL >= 0 && NL <= NL_MAX && NE >= 0 && NE <= NE_MAX, "smpl_lbs_forward: NL=%d NE=%d unsupported", NL, NE);
    DANET_CHECK_ARG(betas && rotmats && v_template && shapedirs && posedirs && J_template && J_shapedirs &&
                    lbs_weights && parents && verts && ctx && ws, "smpl_lbs_forward: null pointer");
    DANET_CHECK_ARG((NL == 0 && NE == 0) || joints54, "smpl_lbs_forward: joints54 is null");
    DANET_CHECK_ARG(NL == 0 || landmark_verts, "smpl_lbs_forward: landmark_verts is null");
    DANET_CHECK_ARG(NE == 0 || J_regressor_extra, "smpl_lbs_forward: J_regressor_extra is null");
    if (ws_floats < danet_smpl_lbs_fwd_ws_floats(B, V, NE))
        return danet::fail(DANET_ERR_WORKSPACE, "smpl_lbs_forward: workspace %zu < %zu floats", ws_floats,
                           danet_smpl_lbs_fwd_ws_floats(B, V, NE));
    hipStream_t s = (hipStream_t)stream;
    const int Bp = bpad_of(B), nt = fwd_ntiles_of(V, B), nbg = fwd_nbg_of(B);
    float* pfT = ws;
    float* jxp = ws + (size_t)NPB_PAD * Bp;
    static const bool no_fused = getenv("DANET_LBS_NO_FUSED") != nullptr;      // A-B timing knob
    if (ticket && nbg == 8 && !no_fused) {                        // ONE launch (smpl_fused_fwd_kernel)
        hipLaunchKernelGGL(smpl_fused_fwd_kernel, dim3(nt, Bp / NBG), dim3(256), 0, s, betas, rotmats, v_template, shapedirs, posedirs,
                           J_template, J_shapedirs, lbs_weights, parents, J_regressor_extra, landmark_verts, B, Bp, V, NB, NL, NE, nt,
                           verts, joints54, ctx, v_posed, jxp, (unsigned*)ticket);
        DANET_CHECK_LAUNCH("smpl_fused_fwd_kernel");
        return DANET_OK;
    }
    hipLaunchKernelGGL(smpl_prep_kernel, dim3(Bp), dim3(64), 0, s, betas, rotmats, J_template, J_shapedirs, parents,
                       B, NB, Bp, NJ + NL + NE, ctx, pfT, joints54);
    DANET_CHECK_LAUNCH("smpl_prep_kernel");
#define LBS_FWD(TV_, NBG_) hipLaunchKernelGGL((smpl_lbs_fwd_kernel<TV_, NBG_>), dim3(nt, Bp / NBG_), dim3(256), 0, s, v_template, shapedirs, posedirs, \
                       lbs_weights, J_regressor_extra, betas, ctx, pfT, B, Bp, V, NB, NE, verts, v_posed, NE > 0 ? jxp : nullptr)
    if (nbg == 32) LBS_FWD(16, 32); else if (nbg == 16) LBS_FWD(32, 16); else LBS_FWD(64, 8);
#undef LBS_FWD
    DANET_CHECK_LAUNCH("smpl_lbs_fwd_kernel");
    if (joints54 && (NL > 0 || NE > 0)) {
        hipLaunchKernelGGL(smpl_finalize_kernel, dim3(B), dim3(256), 0, s, verts, jxp, landmark_verts, Bp, V, NL, NE, nt,
                           joints54);
        DANET_CHECK_LAUNCH("smpl_finalize_kernel");
    }
    return DANET_OK;
}

// Workgroups of smpl_fused_bwd_kernel that are certainly resident together on the current device (occupancy x compute units),
// optionally capped by the caller's co-residency budget (max_blocks > 0: kernels of other streams may occupy compute units, see
// danet_bn_backward_onepass).  DANET_LBS_NO_FUSED_BWD: A-B timing knob.
static bool smpl_bwd_fused_fits(int grid, int max_blocks) {
    static const bool off = getenv("DANET_LBS_NO_FUSED_BWD") != nullptr;
    static int resident = -1;
    if (off) return false;
    if (resident < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&smpl_fused_bwd_kernel), 256, 0) != hipSuccess) {
            (void)hipGetLastError();
            resident = 0;
        } else {
            resident = cus * (per_cu < 2 ? per_cu : 2);             // (never count on more than two per compute unit)
        }
    }
    const int cap = (max_blocks > 0 && max_blocks < resident) ? max_blocks : resident;
    return grid >= 1 && grid <= cap;
}
extern "C" int danet_smpl_lbs_backward_fused_ok(int B, int V, int max_blocks) {
    return smpl_bwd_fused_fits(ntiles_of(V) * (bpad_of(B) / NBG), max_blocks) ? 1 : 0;
}

extern "C" int danet_smpl_lbs_backward(const float* betas, const float* rotmats, int B,
                                       const float* shapedirs, const float* posedirs, const float* J_shapedirs,
                                       const float* lbs_weights, const int32_t* parents,
                                       const float* J_regressor_extra, const int32_t* landmark_verts,
                                       int V, int NB, int NL, int NE,
                                       const float* ctx, const float* v_posed,
                                       const float* g_verts, const float* g_joints54,
                                       float* g_betas, float* g_rotmats,
                                       float* ws, size_t ws_floats, void* bar, int max_blocks, void* stream)
{
    DANET_ENTER();
    (void)betas;
    DANET_CHECK_ARG(B > 0 && V > 0, "smpl_lbs_backward: B=%d V=%d", B, V);
    DANET_CHECK_ARG(NB >= 1 && NB <= NB_MAX, "smpl_lbs_backward: NB=%d unsupported (1..%d)", NB, NB_MAX);
    DANET_CHECK_ARG(rotmats && shapedirs && posedirs && J_shapedirs && lbs_weights && parents && ctx && v_posed &&
                    g_betas && g_rotmats && ws, "smpl_lbs_backward: null pointer");
    DANET_CHECK_ARG(!g_joints54 || ((NL == 0 || landmark_verts) && (NE == 0 || J_regressor_extra)),
                    "smpl_lbs_backward: joint tables missing");
    DANET_CHECK_ARG(NL >= 0 && NL <= NL_MAX && NE >= 0 && NE <= NE_MAX, "smpl_lbs_backward: NL=%d NE=%d unsupported", NL, NE);
    if (ws_floats < danet_smpl_lbs_bwd_ws_floats(B, V, NB))
        return danet::fail(DANET_ERR_WORKSPACE, "smpl_lbs_backward: workspace %zu < %zu floats", ws_floats,
                           danet_smpl_lbs_bwd_ws_floats(B, V, NB));
    hipStream_t s = (hipStream_t)stream;
    const int Bp = bpad_of(B), nt = ntiles_of(V);
    const int npf = (V * 3 + PF_C - 1) / PF_C;
    float* gA = ws;
    float* gPf = gA + (size_t)nt * Bp * 288;
    float* gBt = gPf + (size_t)npf * Bp * NPB_PAD;
    float* gvpT = gBt + (size_t)nt * Bp * NB_MAX;
    const int ny = Bp >= 32 ? 2 : 1;
    if (bar && smpl_bwd_fused_fits(nt * (Bp / NBG), max_blocks)) {          // ONE launch (smpl_fused_bwd_kernel)
        LbsBwdArgs a;
        a.shapedirs = shapedirs; a.posedirs = posedirs; a.lbs_weights = lbs_weights; a.Jx = J_regressor_extra; a.landmark_verts = landmark_verts;
        a.ctx = ctx; a.v_posed = v_posed; a.g_verts = g_verts; a.g_j54 = g_joints54; a.rot = rotmats; a.J_dirs = J_shapedirs; a.parents = parents;
        a.B = B; a.Bpad = Bp; a.V = V; a.NB = NB; a.NL = NL; a.NE = NE; a.ntiles = nt; a.npf = npf; a.ny = ny;
        a.gA = gA; a.gPf = gPf; a.gBt = gBt; a.gvpT = gvpT; a.g_betas = g_betas; a.g_rot = g_rotmats; a.bar = (unsigned*)bar;
        hipLaunchKernelGGL(smpl_fused_bwd_kernel, dim3(nt * (Bp / NBG)), dim3(256), 0, s, a);
        DANET_CHECK_LAUNCH("smpl_fused_bwd_kernel");
        return DANET_OK;
    }
    hipLaunchKernelGGL(smpl_lbs_bwd_kernel, dim3(nt, Bp / NBG), dim3(256), 0, s, shapedirs, posedirs, lbs_weights,
                       J_regressor_extra, landmark_verts, ctx, v_posed, g_verts, g_joints54, B, Bp, V, NB, NL, NE,
                       gA, gvpT, gBt);
    DANET_CHECK_LAUNCH("smpl_lbs_bwd_kernel");
    hipLaunchKernelGGL(smpl_pf_bwd_kernel, dim3(npf, ny), dim3(256), 0, s, posedirs, gvpT, Bp, V * 3, gPf);
    DANET_CHECK_LAUNCH("smpl_pf_bwd_kernel");
    hipLaunchKernelGGL(smpl_finalize_bwd_kernel, dim3(B), dim3(256), 0, s, rotmats, J_shapedirs, parents, ctx,
                       g_joints54, gA, gPf, gBt, Bp, NB, NL, NE, nt, npf, g_betas, g_rotmats);
    DANET_CHECK_LAUNCH("smpl_finalize_bwd_kernel");
    return DANET_OK;
}


// Profiling aid: the phase time stamps (clock64) workgroup (0, 0) of the last smpl_lbs_bwd_kernel launch wrote.
extern "C" int danet_smpl_lbs_debug(long long* out16)
{
    DANET_ENTER();
    DANET_CHECK_ARG(out16, "smpl_lbs_debug: null pointer");
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lbs_dbg), sizeof(long long) * 16, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return danet::fail(DANET_ERR_HIP, "smpl_lbs_debug: %s", hipGetErrorString(e));
    return DANET_OK;
}
