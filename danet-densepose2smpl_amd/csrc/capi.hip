// Library-level entry points: version + thread-local error string.
#include "common.h"
#include "conv_common.h"

namespace danet {
char* last_error_buf() {
    static thread_local char buf[kErrLen] = {0};
    return buf;
}

namespace {
__global__ __launch_bounds__(256) void zero_kernel(unsigned* __restrict__ p, size_t n4, size_t tail0, size_t n) {
    uint4* const q = reinterpret_cast<uint4*>(p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) q[i] = uint4{0u, 0u, 0u, 0u};
    for (size_t i = tail0 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

hipError_t zero_async(void* p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || (reinterpret_cast<size_t>(p) & 3)) return hipErrorInvalidValue;
    const size_t n = bytes / 4;
    const bool al16 = (reinterpret_cast<size_t>(p) & 15) == 0;
    const size_t n4 = al16 ? n / 4 : 0, tail0 = n4 * 4;
    size_t blocks = ((al16 ? n4 : n) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned*)p, n4, tail0, n);
    return hipGetLastError();
}
}  // namespace danet

extern "C" int danet_version(void) { return 100; }

// The library's only run-time switch board (A-B timing and tests; production code never calls it): include/danet_hip.h DANET_KNOB_*.
extern "C" long danet_knob(int id, long value) {
    using namespace danet_conv;
    switch (id) {
        case DANET_KNOB_C3_ENABLE: case DANET_KNOB_C3_MT: case DANET_KNOB_C3_KW: case DANET_KNOB_C3_BLOCKS: case DANET_KNOB_C3_WANT:
            return conv3x3_knob(id, value);
        case DANET_KNOB_C3S_ENABLE: case DANET_KNOB_C3S_BLOCKS: case DANET_KNOB_C3S_KW: case DANET_KNOB_C3S_WANT:
        case DANET_KNOB_C3S_BALANCE: case DANET_KNOB_C3S_TILE_COST:
            return conv3x3s_knob(id, value);
        case DANET_KNOB_PW: return conv_pw_knob(value);
        case DANET_KNOB_PW_WGRAD: return conv_pw_wgrad_knob(value);
        case DANET_KNOB_STEM: return conv_stem_knob(value);
        case DANET_KNOB_STEM_DGRAD: return conv_stem_dgrad_knob(value);
        case DANET_KNOB_C3A: return conv3x3a_knob(value);
        case DANET_KNOB_G3: return conv_g3_knob(value);
        case DANET_KNOB_BN_BLOCK_BYTES: return bn_block_bytes_knob(value);
        default: return -1;
    }
}
extern "C" const char* danet_last_error(void) { return danet::last_error_buf(); }
