// fp32 verification convolution (BASELINE config C4's arithmetic type; tests pin the network structure against the
// reference's fp32 golden vectors through it).  Plain direct kernels -- one thread per output element, fp32 FMAs in
// the reference's summation structure (taps outer, channels inner) -- correctness first, no MFMA: the bf16 kernels
// (conv3x3.hip, conv_fast.hip) are the performance path and are checked per layer against F.conv2d.
// Tensors: x [B,H,W,Cin], y [B,OH,OW,Cout] NHWC fp32; w in torch's [Cout][Cin/groups][R][S] layout (unpacked).
#include "common.h"

namespace {

struct F32P { int B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups; };

__global__ __launch_bounds__(256) void conv_f32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, F32P p, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % p.Cout);
    long pix = i / p.Cout;
    const int ow = (int)(pix % p.OW); pix /= p.OW;
    const int oh = (int)(pix % p.OH);
    const int b = (int)(pix / p.OH);
    const int Cin_g = p.Cin / p.groups, Cout_g = p.Cout / p.groups, g = co / Cout_g;
    float acc = bias ? bias[co] : 0.f;
    for (int r = 0; r < p.R; ++r) {
        const int ih = oh * p.stride - p.pad + r * p.dil;
        if (ih < 0 || ih >= p.H) continue;
        for (int s = 0; s < p.S; ++s) {
            const int iw = ow * p.stride - p.pad + s * p.dil;
            if (iw < 0 || iw >= p.W) continue;
            const float* xp = x + (((size_t)b * p.H + ih) * p.W + iw) * p.Cin + g * Cin_g;
            const float* wp = w + ((size_t)co * Cin_g * p.R + r) * p.S + s;
            for (int ci = 0; ci < Cin_g; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * p.R * p.S], acc);
        }
    }
    y[i] = acc;
}

// dx[b,ih,iw,ci] = sum_{co,r,s} dy[b,oh,ow,co] w[co][ci][r][s]  with  oh*stride - pad + r*dil == ih  (likewise columns)
__global__ __launch_bounds__(256) void conv_f32_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, F32P p, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % p.Cin);
    long pix = i / p.Cin;
    const int iw = (int)(pix % p.W); pix /= p.W;
    const int ih = (int)(pix % p.H);
    const int b = (int)(pix / p.H);
    const int Cin_g = p.Cin / p.groups, Cout_g = p.Cout / p.groups, g = c / Cin_g, ci = c - g * Cin_g;
    float acc = 0.f;
    for (int r = 0; r < p.R; ++r) {
        const int th = ih + p.pad - r * p.dil;
        if (th < 0 || th % p.stride) continue;
        const int oh = th / p.stride;
        if (oh >= p.OH) continue;
        for (int s = 0; s < p.S; ++s) {
            const int tw = iw + p.pad - s * p.dil;
            if (tw < 0 || tw % p.stride) continue;
            const int ow = tw / p.stride;
            if (ow >= p.OW) continue;
            const float* gp = dy + (((size_t)b * p.OH + oh) * p.OW + ow) * p.Cout + g * Cout_g;
            const float* wp = w + (((size_t)(g * Cout_g) * Cin_g + ci) * p.R + r) * p.S + s;
            for (int co = 0; co < Cout_g; ++co) acc = fmaf(gp[co], wp[(size_t)co * Cin_g * p.R * p.S], acc);
        }
    }
    dx[i] = acc;
}

// dw[co][ci][r][s] = sum_{b,oh,ow} dy[b,oh,ow,co] x[b,ih,iw,g*Cin_g+ci]: one 64-lane wave per weight element
__global__ __launch_bounds__(256) void conv_f32_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, F32P p, long total)
{
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (e >= total) return;
    const int Cin_g = p.Cin / p.groups, Cout_g = p.Cout / p.groups;
    long q = e;
    const int s = (int)(q % p.S); q /= p.S;
    const int r = (int)(q % p.R); q /= p.R;
    const int ci = (int)(q % Cin_g);
    const int co = (int)(q / Cin_g);
    const int g = co / Cout_g;
    const long npix = (long)p.B * p.OH * p.OW;
    float acc = 0.f;
    for (long m = lane; m < npix; m += 64) {
        const int ow = (int)(m % p.OW);
        const long t = m / p.OW;
        const int oh = (int)(t % p.OH), b = (int)(t / p.OH);
        const int ih = oh * p.stride - p.pad + r * p.dil, iw = ow * p.stride - p.pad + s * p.dil;
        if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) continue;
        acc = fmaf(dy[m * p.Cout + co], x[(((size_t)b * p.H + ih) * p.W + iw) * p.Cin + g * Cin_g + ci], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) dw[e] = acc;
}

bool fill(F32P& p, int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups) {
    if (!(B > 0 && H > 0 && W > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0 && dil > 0 &&
          groups > 0 && Cin % groups == 0 && Cout % groups == 0)) return false;
    p = F32P{B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups};
    return true;
}

}  // namespace

// mode 0: y = conv(x, w) (+ bias); mode 1: dx = data gradient (a := dy [B,OH,OW,Cout], out := dx [B,H,W,Cin]);
// mode 2: dw = weight gradient (a := x, b := dy, out := dw in torch's [Cout][Cin/groups][R][S] layout).
extern "C" int danet_conv_f32(int mode, const float* a, const float* b, const float* bias, float* out,
                              int B, int H, int W, int Cin, int OH, int OW, int Cout, int R, int S, int stride, int pad, int dil, int groups,
                              void* stream)
{
    DANET_ENTER();
    F32P p;
    DANET_CHECK_ARG(a && b && out && mode >= 0 && mode <= 2 && fill(p, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups), "conv_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {
        const long total = (long)B * OH * OW * Cout;
        hipLaunchKernelGGL(conv_f32_fwd_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, st, a, b, bias, out, p, total);
    } else if (mode == 1) {
        const long total = (long)B * H * W * Cin;
        hipLaunchKernelGGL(conv_f32_dgrad_kernel, dim3(danet::cdiv(total, 256)), dim3(256), 0, st, a, b, out, p, total);
    } else {
        const long total = (long)Cout * (Cin / groups) * R * S;
        hipLaunchKernelGGL(conv_f32_wgrad_kernel, dim3(danet::cdiv(total * 64, 256)), dim3(256), 0, st, a, b, out, p, total);
    }
    DANET_CHECK_LAUNCH("conv_f32_kernel");
    return DANET_OK;
}
