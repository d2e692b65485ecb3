// Shared helpers of libdanet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "danet_hip.h"

namespace danet {

char* last_error_buf();
constexpr int kErrLen = 512;

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), kErrLen, fmt, ap);
    va_end(ap);
    return code;
}

#define DANET_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return ::danet::fail(DANET_ERR_ARG, __VA_ARGS__); } while (0)

// first statement of every enqueueing entry point: drop any sticky error left by an earlier,
// unrelated HIP call so that DANET_CHECK_LAUNCH reports only our own launch failures
#define DANET_ENTER() (void)hipGetLastError()

#define DANET_CHECK_LAUNCH(name) \
    do { hipError_t e_ = hipGetLastError(); \
         if (e_ != hipSuccess) return ::danet::fail(DANET_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Zero `bytes` (a multiple of 4) at p (4-byte aligned) with a KERNEL on `stream` (capi.hip).  The library never enqueues
// hipMemsetAsync: captured into a hipGraph it becomes a memset NODE, and on this stack (ROCm 7.0 runtime under torch 2.10)
// a memset node followed by a kernel that accumulates into the same buffer was observed to race -- round 5 found one head
// bias gradient in two turning into inf after a few replays of the captured train step (tools/replay_probe.py; the bias
// gradients of danet_channel_sum were the step's only memset nodes), never in eager execution.
hipError_t zero_async(void* p, size_t bytes, hipStream_t stream);

}  // namespace danet
