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

}  // namespace danet
