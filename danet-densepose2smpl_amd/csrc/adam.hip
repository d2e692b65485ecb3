// Multi-tensor Adam step (the optimizer of /root/reference/train/trainer.py:42-44: torch.optim.Adam, lr 1e-4,
// betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) over ALL parameters in one launch.
//
// A step is pure streaming: 16 B read + 12 B written per element (~2.9 GB for DaNet's 102 M parameters).  PyTorch's
// fused multi-tensor Adam needs ~31 launches and reaches ~1.8 TB/s here; this kernel walks a device table of
// <= 32768-element chunks {param, grad, offset into the flat moment buffers, count} with one workgroup per
// chunk.  The learning rate and the step count are read from device memory, so the step can sit inside a
// captured hipGraph and the host can still decay the rate between replays.
//
// Per-parameter step counts, as torch.optim.Adam keeps them: a parameter that received no gradient in a step (gradient
// pointer NULL, or used[param] == 0 where all gradients are views of one store; `used` is the number of RANKS in which the
// parameter received a gradient, all-reduced with the gradients, so every replica decides alike) is skipped entirely -- moments untouched --
// and idle[param] counts those steps; its bias corrections use step - idle[param], i.e. the number of steps it WAS updated
// in.  (The reference pre-trains the IUV estimator alone for 5000 steps, /root/reference/train/base_trainer.py:74: the
// regressor's first update must see step 1, not 5001.)
//
// Poison guard: `poison` points at the error word of THIS device's one-pass BatchNorm backward grid barrier (norm_act.hip,
// bar[2]); `poison_sum` at a float that holds the SUM of that word over all ranks (distributed.GradStore stamps the local word
// into the tail of its last gradient bucket, whose all-reduce sums it -- like the `used` mask, a global fact).  A barrier
// that timed out in this step left garbage gradients on its rank, and after the all-reduce in every rank's sums; with either
// value set EVERY parameter is skipped on EVERY rank (and counted in idle, so the bias corrections stay those of the updates
// that were applied): a poisoned step can never reach the weights or the moments of any replica, whatever the host does or
// does not check (Trainer.check_onepass reports it, on all ranks alike).  One process: poison alone (poison_sum NULL).
#include "common.h"

namespace {

struct AdamChunk { float* p; const float* g; long off; int n; int param; };     // param: 2 * parameter index + (1 for the parameter's first chunk)

__global__ __launch_bounds__(256) void adam_kernel(const AdamChunk* __restrict__ table, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ lr_p, const float* __restrict__ step_p,
                                                   const float* __restrict__ used, float* __restrict__ idle,
                                                   float beta1, float beta2, float eps, float gscale, const int* __restrict__ poison,
                                                   const float* __restrict__ poison_sum)
{
    const AdamChunk c = table[blockIdx.x];
    const int pi = c.param >> 1;
    if (!c.g || (used && !(used[pi] > 0.f)) || (poison && poison[0] != 0) || (poison_sum && poison_sum[0] > 0.f)) {                             // parameter without a gradient this step: skipped, and counted
        if (idle && (c.param & 1) && threadIdx.x == 0) idle[pi] += 1.f;
        return;
    }
    const float t = step_p[0] - (idle ? idle[pi] : 0.f);           // (nobody writes idle[pi] in a launch that reads it)
    const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
    const float step_size = lr_p[0] / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    float* __restrict__ mp = m + c.off;
    float* __restrict__ vp = v + c.off;
    const int n4 = c.n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 g = reinterpret_cast<const float4*>(c.g)[i];
        g.x *= gscale; g.y *= gscale; g.z *= gscale; g.w *= gscale;
        float4 p = reinterpret_cast<float4*>(c.p)[i];
        float4 mm = reinterpret_cast<float4*>(mp)[i];
        float4 vv = reinterpret_cast<float4*>(vp)[i];
#define ADAM1(x) { mm.x = beta1 * mm.x + (1.f - beta1) * g.x; vv.x = beta2 * vv.x + (1.f - beta2) * g.x * g.x; \
                   p.x -= step_size * mm.x / (sqrtf(vv.x) * inv_sqrt_bc2 + eps); }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(c.p)[i] = p;
        reinterpret_cast<float4*>(mp)[i] = mm;
        reinterpret_cast<float4*>(vp)[i] = vv;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
        const float g = c.g[i] * gscale;
        const float mm = beta1 * mp[i] + (1.f - beta1) * g, vv = beta2 * vp[i] + (1.f - beta2) * g * g;
        mp[i] = mm; vp[i] = vv;
        c.p[i] -= step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    }
}

}  // namespace

extern "C" size_t danet_adam_chunk_bytes(void) { return sizeof(AdamChunk); }

// table: nchunks entries { float* p; const float* g (NULL = skip); int64 off; int32 n; int32 param } on the device (param = 2 *
// parameter index + 1 for the parameter's first chunk); p, g and the moment buffers m, v (+ off) must be 16-byte aligned for
// every chunk; lr and step (the 1-based GLOBAL step count, as a float) live on the device.  used (NULL = every parameter with a
// gradient pointer): float per parameter, > 0 = some rank produced a gradient this step; idle (NULL = none): float per
// parameter, the steps it was skipped in, maintained here.  grad_scale multiplies every gradient (1 / world size turns
// all-reduced sums into the average without a pass of its own).  poison (NULL = none): device int; non-zero = skip the
// whole step; poison_sum (NULL = none): device float, the all-reduced sum of the ranks' words; > 0 = skip (see the header).
extern "C" int danet_adam_step(const void* table, int nchunks, float* m, float* v, const float* lr, const float* step,
                               const float* used, float* idle, float beta1, float beta2, float eps, float grad_scale,
                               const int* poison, const float* poison_sum, void* stream)
{
    DANET_ENTER();
    DANET_CHECK_ARG(table && nchunks > 0 && m && v && lr && step, "adam_step: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)table, m, v, lr, step, used, idle, beta1, beta2, eps, grad_scale, poison, poison_sum);
    DANET_CHECK_LAUNCH("adam_kernel");
    return DANET_OK;
}
