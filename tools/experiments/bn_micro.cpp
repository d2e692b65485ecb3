// Per-launch time of the BatchNorm entry points of libdanet_hip.so on the four HRNet branch shapes (B = 32), called
// straight through the C ABI (no framework in the loop): forward multi (with / without residual), backward two-kernel
// multi, backward one-pass; and the plain elementwise-stream floor for the same bytes (tools/experiments/stream_width.hip).
// build: hipcc --offload-arch=gfx950 -O2 tools/experiments/bn_micro.cpp -Iinclude -Ldanet-densepose2smpl_amd/csrc -ldanet_hip
//        -Wl,-rpath,'$ORIGIN/../../danet-densepose2smpl_amd/csrc' -o tools/experiments/bn_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "danet_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define DK(x) do { int r_ = (x); if (r_ != 0) { printf("danet error %d at %d: %s\n", r_, __LINE__, danet_last_error()); exit(1); } } while (0)

struct BnFwdJob { const void* x; const void* res; void* y; const float* gamma; const float* beta; float* running_mean; float* running_var;
                  float* saved; float* sums; void* mask; int64_t M; int C, sums_state, relu; };
struct BnBwdJob { const void* dy; const void* x; const void* y; const float* gamma; const float* saved; void* dx; void* dres; float* dparam;
                  float* red; const float* beta; const void* mask; int64_t M; int C, red_state, relu, mask_mode; };

template <class F>
double time_us(F f, int reps = 50)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main()
{
    const int B = 32, C[4] = {48, 96, 192, 384}, S[4] = {64, 32, 16, 8};
    void *x[4], *y[4], *res[4], *dy[4], *dx[4], *dres[4], *mask[4];
    float *gamma[4], *beta[4], *rm[4], *rv[4], *saved[4], *sums[4], *red[4], *dparam[4];
    double bytes = 0;
    for (int i = 0; i < 4; ++i) {
        const size_t n = (size_t)B * S[i] * S[i] * C[i];
        bytes += n * 2;
        for (void** p : {&x[i], &y[i], &res[i], &dy[i], &dx[i], &dres[i]}) { CK(hipMalloc(p, n * 2)); CK(hipMemset(*p, 0x3c, n * 2)); }
        CK(hipMalloc(&mask[i], n / 4));
        for (float** p : {&gamma[i], &beta[i], &rm[i], &rv[i]}) { CK(hipMalloc((void**)p, C[i] * 4)); CK(hipMemset(*p, 0, C[i] * 4)); }
        CK(hipMalloc((void**)&saved[i], 2 * C[i] * 4)); CK(hipMemset(saved[i], 0, 2 * C[i] * 4));
        CK(hipMalloc((void**)&dparam[i], 2 * C[i] * 4));
        const size_t ws = danet_bn_ws_floats(C[i]) * 4;
        CK(hipMalloc((void**)&sums[i], ws)); CK(hipMemset(sums[i], 0, ws));
        CK(hipMalloc((void**)&red[i], ws)); CK(hipMemset(red[i], 0, ws));
    }
    unsigned* bar; CK(hipMalloc((void**)&bar, 4096)); CK(hipMemset(bar, 0, 4096));
    printf("four branches: %.1f MB per tensor set\n", bytes / 1e6);
    for (int n : {4, 1}) {
        for (int with_res : {0, 1}) {
            std::vector<BnFwdJob> fj(n);
            std::vector<BnBwdJob> bj(n);
            for (int i = 0; i < n; ++i) {
                const int64_t M = (int64_t)B * S[i] * S[i];
                fj[i] = BnFwdJob{x[i], with_res ? res[i] : nullptr, y[i], gamma[i], beta[i], rm[i], rv[i], saved[i], sums[i], with_res ? mask[i] : nullptr, M, C[i], 2, 1};
                bj[i] = BnBwdJob{dy[i], x[i], y[i], gamma[i], saved[i], dx[i], with_res ? dres[i] : nullptr, dparam[i], red[i], beta[i], with_res ? mask[i] : nullptr,
                                 M, C[i], 1, 1, with_res ? 1 : 2};
            }
            double b = 0;
            for (int i = 0; i < n; ++i) b += (double)B * S[i] * S[i] * C[i] * 2;
            const double tf = time_us([&] { DK(danet_bn_forward_multi(fj.data(), n, 0.1f, 1e-5f, nullptr)); });
            printf("n=%d res=%d forward multi      %7.2f us  (%.1f MB moved: %.2f TB/s)\n", n, with_res, tf, b * (2 + with_res) / 1e6, b * (2 + with_res) / tf / 1e6);
            // (the reductions accumulate into `red` launch after launch: values grow, timing is unaffected)
            const double tb = time_us([&] { DK(danet_bn_backward_multi(bj.data(), n, nullptr)); });
            printf("n=%d res=%d backward two-kernel %7.2f us  (%.1f MB moved: %.2f TB/s)\n", n, with_res, tb, b * (5 + with_res) / 1e6, b * (5 + with_res) / tb / 1e6);
            {   // apply alone: the sums pre-accumulated (red_state 2)
                std::vector<BnBwdJob> aj = bj;
                for (auto& j : aj) j.red_state = 2;
                const double ta = time_us([&] { DK(danet_bn_backward_multi(aj.data(), n, nullptr)); });
                printf("n=%d res=%d   of which apply     %7.2f us  (reduce %.2f us)\n", n, with_res, ta, tb - ta);
            }
            if (danet_bn_backward_onepass_ok(bj.data(), n, 0)) {
                const double to = time_us([&] { DK(danet_bn_backward_onepass(bj.data(), n, bar, 0, nullptr)); });
                printf("n=%d res=%d backward one-pass   %7.2f us  (%.1f MB moved: %.2f TB/s)\n", n, with_res, to, b * (3 + with_res) / 1e6, b * (3 + with_res) / to / 1e6);
            } else printf("n=%d res=%d one-pass: not taken\n", n, with_res);
        }
    }
    unsigned hb[4]; CK(hipMemcpy(hb, bar, 16, hipMemcpyDeviceToHost));
    printf("barrier error flag %u\n", hb[2]);
    return 0;
}
