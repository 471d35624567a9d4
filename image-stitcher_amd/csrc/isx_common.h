// Shared host-side helpers for libisx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/isx.h"

// thread-local error text returned by isx_last_error()
int isx_fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define ISX_HIP_TRY(expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return isx_fail(ISX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                              \
    } while (0)

#define ISX_REQUIRE(cond, ...)                          \
    do {                                                \
        if (!(cond)) return isx_fail(ISX_E_ARG, __VA_ARGS__); \
    } while (0)

static inline int isx_launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return isx_fail(ISX_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return ISX_OK;
}
