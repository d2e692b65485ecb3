// PROTOTYPE (round 3, compile-checked only: the round's GPU budget was spent): the SMPL forward as ONE launch -- DESIGN.md 3.3
// "the one-launch plan".  Not part of libdanet_hip.so.  The harness below runs the library's three-launch forward
// (danet_smpl_lbs_forward) and this kernel on the same seeded model / inputs, prints the largest differences of vertices and
// joints and times both as back-to-back launches.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/experiments/smpl_fused_fwd.hip \
//          -Ldanet-densepose2smpl_amd/csrc -ldanet_hip -Wl,-rpath,'$ORIGIN/../../danet-densepose2smpl_amd/csrc' -o tools/experiments/smpl_fused_fwd
//   run  : tools/experiments/smpl_fused_fwd [B]
//
// What is different from smpl_lbs.hip's prep / main / finalize:
//   * every workgroup (64 vertices x 8 items) recomputes its items' kinematic chain in LDS: 8 items x 12 lanes, a barrier per
//     joint; the workgroups of tile 0 also write the posed joints;
//   * the pose feature (R - I of joints 1..23) is read straight from `rotmats` with wave-uniform one-dword scalar loads, the
//     identity as `acc -= posedirs row` on a joint's three diagonal rows -- no transposed copy in memory;
//   * landmark vertices are written by the tile that owns them; the regressed extra joints are reduced by the LAST workgroup
//     of a batch group to arrive: partials leave with agent-scope atomic exchanges (performed at the memory side: the XCDs' L2s
//     are not coherent for plain stores inside a kernel), an agent-scope ticket counts arrivals, the last one sums the tiles in
//     a fixed order with agent-scope loads and resets the ticket.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "danet_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

namespace {
constexpr int NJ = 24, NPB = 207, TV = 64, TC = TV * 3, NBG = 8, WPAD = 25, NB_MAX = 16, NE_MAX = 28, NQ = 3;

template <int PF>
__global__ __launch_bounds__(256) void smpl_fused_fwd_kernel(
    const float* __restrict__ betas, const float* __restrict__ rot,
    const float* __restrict__ v_template, const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ J_template, const float* __restrict__ J_dirs, const float* __restrict__ lbs_weights,
    const int* __restrict__ parents, const float* __restrict__ Jx, const int* __restrict__ landmark_verts,
    int B, int V, int NB, int NL, int NE, int ntiles,
    float* __restrict__ verts, float* __restrict__ joints54, float* jx_partial, unsigned* ticket)
{
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
    __shared__ __attribute__((aligned(16))) float sPf[PF ? 208 : 1][NBG];      // PF = 1: the group's pose features, read back as broadcast ds_read_b128

    // ---- phase 0: inputs of the group, tile constants, the chain ------------------------------------------------------------
    for (int i = t; i < NBG * 216; i += 256) { const int bb = i / 216, e = i - bb * 216; sR[bb][e] = b0 + bb < B ? rot[(size_t)(b0 + bb) * 216 + e] : (e % 9 % 4 == 0 ? 1.f : 0.f); }
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
    if (tile == 0)
        for (int i = t; i < NBG * 72; i += 256) { const int bb = i / 72, e = i - bb * 72; if (b0 + bb < B) joints54[(size_t)(b0 + bb) * NJ54 * 3 + e] = sJp[bb][e]; }
    if (PF)
        for (int i = t; i < 208 * NBG; i += 256) {
            const int k = i / NBG, bb = i - k * NBG, e = k % 9;
            sPf[PF ? k : 0][bb] = k < NPB ? sR[bb][9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f) : 0.f;
        }
    __syncthreads();                                                // the chain's arrays are free from here on (sPart)

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
        const int kbeg = w * 52;
        const float* rb[NBG];
#pragma unroll
        for (int bb = 0; bb < NBG; ++bb) rb[bb] = rot + (size_t)(b0 + bb < B ? b0 + bb : B - 1) * 216 + 9;      // (padding items: any valid row; their results are dropped)
        if constexpr (PF == 0) {
#pragma unroll 4
            for (int kk = 0; kk < 52; ++kk) {
                const int k = kbeg + kk;
                const int kr = k < NPB ? k : NPB - 1;
                float pq[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) pq[q] = k < NPB ? posedirs[(size_t)kr * C + cq[q]] : 0.f;
                const int e = kr % 9;
                const bool diag = e == 0 || e == 4 || e == 8;          // (wave-uniform)
#pragma unroll
                for (int bb = 0; bb < NBG; ++bb) {
                    const float f = rb[bb][kr];                         // wave-uniform address: a scalar load
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q][bb] += f * pq[q];
                }
                if (diag) {
#pragma unroll
                    for (int bb = 0; bb < NBG; ++bb)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[q][bb] -= pq[q];
                }
            }
        } else {
#pragma unroll 13
            for (int kk = 0; kk < 52; ++kk) {
                const int k = kbeg + kk;
                const int kr = k < NPB ? k : NPB - 1;
                float pq[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) pq[q] = posedirs[(size_t)kr * C + cq[q]];          // (row 207 of sPf is zero)
                const float4 f0 = *reinterpret_cast<const float4*>(&sPf[PF ? k : 0][0]), f1 = *reinterpret_cast<const float4*>(&sPf[PF ? k : 0][4]);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[q][0] += f0.x * pq[q]; acc[q][1] += f0.y * pq[q]; acc[q][2] += f0.z * pq[q]; acc[q][3] += f0.w * pq[q];
                    acc[q][4] += f1.x * pq[q]; acc[q][5] += f1.y * pq[q]; acc[q][6] += f1.z * pq[q]; acc[q][7] += f1.w * pq[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bb = 0; bb < NBG; ++bb) sPart[w][bb][lane + 64 * q] = acc[q][bb];
    }
    __syncthreads();
    // ---- phase 2: shape blend-shapes + the four waves' partial sums, then skinning ---------------------------------------------
    for (int o = t; o < TC * NBG; o += 256) {
        const int bb = o / TC, cl_ = o - bb * TC;
        float a = sBase[cl_] + ((sPart[0][bb][cl_] + sPart[1][bb][cl_]) + (sPart[2][bb][cl_] + sPart[3][bb][cl_]));
#pragma unroll
        for (int l = 0; l < NB_MAX; ++l) a += sSd[cl_][l] * sB[bb][l];
        sVp[bb][cl_] = a;
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
    for (int i = t; i < NBG * NL; i += 256) {
        const int bb = i / NL, l = i - bb * NL, lv = landmark_verts[l];
        if (lv >= v0 && lv < v0 + TV && b0 + bb < B)
            for (int k = 0; k < 3; ++k) joints54[((size_t)(b0 + bb) * NJ54 + NJ + l) * 3 + k] = sVp[bb][(lv - v0) * 3 + k];
    }
    if (NE > 0) {
        const int NO = NE * 3;
        const int Bpad = gridDim.y * NBG;
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
            // one lane per output, the tiles in a fixed order, twelve agent-scope loads in flight
            if (t < NBG * NO) {
                const int bb = t / NO, q = t - bb * NO;
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

float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65535.f - 0.5f; }

template <typename T> T* upload(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
}  // namespace

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32, V = 6890, NB = 10, NL = 21, NE = 9, C = V * 3, NJ54 = NJ + NL + NE;
    unsigned seed = 12345;
    std::vector<float> vt(C), sd((size_t)C * NB), pd((size_t)NPB * C), jt(72), jd(72 * NB), lw((size_t)V * NJ), jx((size_t)NE * V), betas(B * NB), rot((size_t)B * 216);
    std::vector<int> parents = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21}, lm(NL);
    for (auto& v : vt) v = frand(seed); for (auto& v : sd) v = 0.02f * frand(seed); for (auto& v : pd) v = 0.01f * frand(seed);
    for (auto& v : jt) v = frand(seed); for (auto& v : jd) v = 0.05f * frand(seed);
    for (int v = 0; v < V; ++v) { float s = 0; for (int j = 0; j < NJ; ++j) { lw[(size_t)v * NJ + j] = (j % 6 == v % 6) ? frand(seed) + 0.6f : 0.f; s += lw[(size_t)v * NJ + j]; } for (int j = 0; j < NJ; ++j) lw[(size_t)v * NJ + j] /= s; }
    for (auto& v : jx) v = (frand(seed) + 0.5f) / V;
    for (int l = 0; l < NL; ++l) lm[l] = (int)((l * 331u + 17u) % V);
    for (auto& v : betas) v = 2.f * frand(seed);
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < NJ; ++j) {                             // rotations about a random axis (Rodrigues)
            float ax[3] = {frand(seed), frand(seed), frand(seed)}; const float n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-6f;
            for (float& a : ax) a /= n;
            const float th = 1.5f * frand(seed), c = std::cos(th), s = std::sin(th);
            float* R = &rot[(size_t)b * 216 + j * 9];
            const float K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
                float kk = 0; for (int m = 0; m < 3; ++m) kk += K[r * 3 + m] * K[m * 3 + q];
                R[r * 3 + q] = (r == q ? 1.f : 0.f) + s * K[r * 3 + q] + (1 - c) * kk;
            }
        }
    float *d_vt = upload(vt), *d_sd = upload(sd), *d_pd = upload(pd), *d_jt = upload(jt), *d_jd = upload(jd), *d_lw = upload(lw), *d_jx = upload(jx), *d_b = upload(betas), *d_r = upload(rot);
    int *d_par = upload(parents), *d_lm = upload(lm);
    const size_t nctx = danet_smpl_lbs_ctx_floats(B), nws = danet_smpl_lbs_fwd_ws_floats(B, V, NE);
    const int Bpad = (B + NBG - 1) / NBG * NBG, ntiles = (V + TV - 1) / TV;
    float *verts0, *verts1, *j0, *j1, *ctx, *ws, *jxp; unsigned* ticket;
    CK(hipMalloc(&verts0, (size_t)B * C * 4)); CK(hipMalloc(&verts1, (size_t)B * C * 4)); CK(hipMalloc(&j0, (size_t)B * NJ54 * 12)); CK(hipMalloc(&j1, (size_t)B * NJ54 * 12));
    CK(hipMalloc(&ctx, nctx * 4)); CK(hipMalloc(&ws, nws * 4)); CK(hipMalloc(&jxp, (size_t)ntiles * Bpad * NE * 3 * 4)); CK(hipMalloc(&ticket, 64 * 4));
    CK(hipMemset(ticket, 0, 64 * 4)); CK(hipMemset(j1, 0, (size_t)B * NJ54 * 12));
    auto ref = [&]() { if (danet_smpl_lbs_forward(d_b, d_r, B, d_vt, d_sd, d_pd, d_jt, d_jd, d_lw, d_par, d_jx, d_lm, V, NB, NL, NE, verts0, j0, ctx, nullptr, ws, nws, nullptr) != 0) { printf("reference failed: %s\n", danet_last_error()); exit(1); } };
    auto fused0 = [&]() { hipLaunchKernelGGL(smpl_fused_fwd_kernel<0>, dim3(ntiles, Bpad / NBG), dim3(256), 0, 0, d_b, d_r, d_vt, d_sd, d_pd, d_jt, d_jd, d_lw, d_par, d_jx, d_lm, B, V, NB, NL, NE, ntiles, verts1, j1, jxp, ticket); };
    auto fused1 = [&]() { hipLaunchKernelGGL(smpl_fused_fwd_kernel<1>, dim3(ntiles, Bpad / NBG), dim3(256), 0, 0, d_b, d_r, d_vt, d_sd, d_pd, d_jt, d_jd, d_lw, d_par, d_jx, d_lm, B, V, NB, NL, NE, ntiles, verts1, j1, jxp, ticket); };
    ref();
    CK(hipDeviceSynchronize());
    std::vector<float> h0((size_t)B * C), h1((size_t)B * C), g0((size_t)B * NJ54 * 3), g1((size_t)B * NJ54 * 3);
    CK(hipMemcpy(h0.data(), verts0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g0.data(), j0, g0.size() * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 100;
    float ms_ref;
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) ref(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1));
    for (int variant = 0; variant < 2; ++variant) {
        CK(hipMemset(verts1, 0, (size_t)B * C * 4)); CK(hipMemset(j1, 0, (size_t)B * NJ54 * 12));
        if (variant == 0) { fused0(); fused0(); } else { fused1(); fused1(); }      // (twice: the ticket reset is part of what is checked)
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h1.data(), verts1, h1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(g1.data(), j1, g1.size() * 4, hipMemcpyDeviceToHost));
        float dv = 0, dj = 0;
        for (size_t i = 0; i < h0.size(); ++i) dv = std::fmax(dv, std::fabs(h0[i] - h1[i]));
        for (size_t i = 0; i < g0.size(); ++i) dj = std::fmax(dj, std::fabs(g0[i] - g1[i]));
        float ms_fused;
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) { if (variant == 0) fused0(); else fused1(); } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_fused, e0, e1));
        printf("{\"smpl_fused_fwd\": {\"B\": %d, \"pose_feature\": \"%s\", \"max_abs_diff_verts\": %.3g, \"max_abs_diff_joints\": %.3g, \"three_launches_us\": %.1f, \"one_launch_us\": %.1f}}\n",
               B, variant == 0 ? "scalar loads from rotmats" : "LDS broadcast", dv, dj, ms_ref * 1e3 / reps, ms_fused * 1e3 / reps);
    }
    return 0;
}
