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

// HIP forms the global size gridDim.x * blockDim.x in 32 bits: a launch of 2^32 threads or more silently runs
// a truncated grid.  Launches over an item list are therefore split; this is the number of items per launch
// (a multiple of 8, which the XCD-aware block -> item maps rely on), 0 if not even 8 items fit.
static inline int isx_items_per_launch(int blocks_per_item, int threads) {
    const int64_t max_blocks = ((int64_t)1 << 31) / threads;
    const int64_t per = (max_blocks / blocks_per_item) & ~(int64_t)7;
    return (int)(per > (1 << 30) ? (1 << 30) : per);
}

// bbox of gray != 0 pixels: init + per-block partials into partial_dev (ISX_BBOX_PARTIAL_BLOCKS*4 int32 of device
// scratch) + finish into result_dev (5 int32).  Shared by isx_bbox_nonzero and its static-scratch variant.
int isx_bbox_nonzero_launch(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int32_t *result_dev,
                            int32_t *partial_dev, hipStream_t st);
