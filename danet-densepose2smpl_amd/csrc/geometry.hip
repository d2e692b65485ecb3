// Rotation helpers of /root/reference/utils/geometry.py as single-launch HIP kernels
// (the reference issues ~10 tiny torch kernels for each of these).
#include "common.h"

namespace {

__device__ inline void quat_to_R(float w, float x, float y, float z, float* R) {
    // geometry.py:25-45 -- normalise, then the standard (w,x,y,z) -> R expansion
    const float n = sqrtf(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;    R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;    R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
}

__global__ void batch_rodrigues_kernel(const float* __restrict__ theta, int N, float* __restrict__ R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float tx = theta[i * 3 + 0], ty = theta[i * 3 + 1], tz = theta[i * 3 + 2];
    // geometry.py:16-22: angle = ||theta + 1e-8||, axis = theta / angle, quaternion of half angle
    const float ax = tx + 1e-8f, ay = ty + 1e-8f, az = tz + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float nx = tx / angle, ny = ty / angle, nz = tz / angle;
    const float h = angle * 0.5f;
    const float c = cosf(h), s = sinf(h);
    float out[9];
    quat_to_R(c, s * nx, s * ny, s * nz, out);
#pragma unroll
    for (int e = 0; e < 9; ++e) R[(size_t)i * 9 + e] = out[e];
}

__global__ void rodrigues_smplx_kernel(const float* __restrict__ theta, int N, float* __restrict__ R) {
    // smplx.lbs.batch_rodrigues: angle = ||theta + 1e-8||, K = skew(theta/angle),
    // R = I + sin K + (1 - cos) K K
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float tx = theta[i * 3 + 0], ty = theta[i * 3 + 1], tz = theta[i * 3 + 2];
    const float ax = tx + 1e-8f, ay = ty + 1e-8f, az = tz + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float rx = tx / angle, ry = ty / angle, rz = tz / angle;
    const float s = sinf(angle), c1 = 1.0f - cosf(angle);
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const float kk = K[r * 3 + 0] * K[0 + cc] + K[r * 3 + 1] * K[3 + cc] + K[r * 3 + 2] * K[6 + cc];
            R[(size_t)i * 9 + r * 3 + cc] = (r == cc ? 1.f : 0.f) + s * K[r * 3 + cc] + c1 * kk;
        }
}

__global__ void rot6d_fwd_kernel(const float* __restrict__ x, int N, float* __restrict__ R) {
    // geometry.py:55-61: x viewed [3,2]; a1 = x[:,0], a2 = x[:,1]; F.normalize eps = 1e-12
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* p = x + (size_t)i * 6;
    const float a1[3] = {p[0], p[2], p[4]}, a2[3] = {p[1], p[3], p[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    float* o = R + (size_t)i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[r * 3 + 0] = b1[r]; o[r * 3 + 1] = b2[r]; o[r * 3 + 2] = b3[r]; }
}

__global__ void rot6d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gR, int N, float* __restrict__ gx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* p = x + (size_t)i * 6;
    const float* g = gR + (size_t)i * 9;
    const float a1[3] = {p[0], p[2], p[4]}, a2[3] = {p[1], p[3], p[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float gb1[3] = {g[0], g[3], g[6]}, gb2[3] = {g[1], g[4], g[7]};
    const float gb3[3] = {g[2], g[5], g[8]};
    // b3 = b1 x b2
    gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1]; gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2]; gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
    gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1]; gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2]; gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
    // b2 = u / |u|
    const float t2 = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
    const float gu[3] = {(gb2[0] - b2[0] * t2) / n2, (gb2[1] - b2[1] * t2) / n2, (gb2[2] - b2[2] * t2) / n2};
    // u = a2 - (b1.a2) b1
    const float gub1 = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
    const float ga2[3] = {gu[0] - gub1 * b1[0], gu[1] - gub1 * b1[1], gu[2] - gub1 * b1[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) gb1[k] += -d * gu[k] - gub1 * a2[k];
    // b1 = a1 / |a1|
    const float t1 = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
    const float ga1[3] = {(gb1[0] - b1[0] * t1) / n1, (gb1[1] - b1[1] * t1) / n1, (gb1[2] - b1[2] * t1) / n1};
    float* o = gx + (size_t)i * 6;
    o[0] = ga1[0]; o[2] = ga1[1]; o[4] = ga1[2];
    o[1] = ga2[0]; o[3] = ga2[1]; o[5] = ga2[2];
}

}  // namespace

#define LAUNCH_1D(kernel, N, stream, ...)                                                         \
    hipLaunchKernelGGL(kernel, dim3(danet::cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int danet_batch_rodrigues(const float* theta, int N, float* R, void* stream) {
    DANET_ENTER();
    DANET_CHECK_ARG(theta && R && N > 0, "batch_rodrigues: bad arguments");
    LAUNCH_1D(batch_rodrigues_kernel, N, stream, theta, N, R);
    DANET_CHECK_LAUNCH("batch_rodrigues_kernel");
    return DANET_OK;
}
extern "C" int danet_rodrigues_smplx(const float* theta, int N, float* R, void* stream) {
    DANET_ENTER();
    DANET_CHECK_ARG(theta && R && N > 0, "rodrigues_smplx: bad arguments");
    LAUNCH_1D(rodrigues_smplx_kernel, N, stream, theta, N, R);
    DANET_CHECK_LAUNCH("rodrigues_smplx_kernel");
    return DANET_OK;
}
extern "C" int danet_rot6d_to_rotmat_forward(const float* x, int N, float* R, void* stream) {
    DANET_ENTER();
    DANET_CHECK_ARG(x && R && N > 0, "rot6d_to_rotmat_forward: bad arguments");
    LAUNCH_1D(rot6d_fwd_kernel, N, stream, x, N, R);
    DANET_CHECK_LAUNCH("rot6d_fwd_kernel");
    return DANET_OK;
}
extern "C" int danet_rot6d_to_rotmat_backward(const float* x, const float* gR, int N, float* gx, void* stream) {
    DANET_ENTER();
    DANET_CHECK_ARG(x && gR && gx && N > 0, "rot6d_to_rotmat_backward: bad arguments");
    LAUNCH_1D(rot6d_bwd_kernel, N, stream, x, gR, N, gx);
    DANET_CHECK_LAUNCH("rot6d_bwd_kernel");
    return DANET_OK;
}
