// libisx_hip.so -- the one entry point that allocates nothing (static device scratch).
#include "isx_common.h"
// ------------------------------------------------------------------------------------------------
// isx_bbox_nonzero_static: isx_bbox_nonzero for callers that cannot allocate (tight serving loops, harnesses that
// forbid hipMalloc; graph capture only after one eager warm-up call per device, see include/isx.h): the per-block
// partial table lives in a library-owned static device array instead
// of the tail of a caller-allocated accumulator, so the caller passes only its 5-int result.  No
// counterpart in the reference.  The entry point calls no hipMalloc / hipFree / allocating routine.
//   * the array is 32 KiB per GPU the library runs on, held until the process ends;
//   * every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
//     threads) race -- one stream per GPU; for SERIALISED calls the array keeping the last call's partials is
//     harmless, because a call only reads the entries [0, grid) its own kernel has just written;
//   * `scratch` below is a host-side stand-in for the symbol, NOT a device address: host code uses the
//     per-device pointer obtained once with hipGetSymbolAddress, and the kernels take it as an argument.
// ------------------------------------------------------------------------------------------------
#include <mutex>
static __device__ __align__(256) int32_t scratch[ISX_BBOX_PARTIAL_BLOCKS * 4];
#define ISX_MAX_DEVICES 64
static int32_t *g_scratch_ptr[ISX_MAX_DEVICES]; // per device: each GPU holds its own copy at its own address
static std::mutex g_scratch_mutex;

// device pointer of `scratch` on the CURRENT device (resolved on the first call per device)
static int static_scratch_ptr(int32_t **out) {
    int dev = -1;
    ISX_HIP_TRY(hipGetDevice(&dev));
    ISX_REQUIRE(dev >= 0 && dev < ISX_MAX_DEVICES, "device ordinal %d outside 0..%d", dev, ISX_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        ISX_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        ISX_REQUIRE(p, "hipGetSymbolAddress returned a null address for the static scratch");
        g_scratch_ptr[dev] = (int32_t *)p;
    }
    *out = g_scratch_ptr[dev];
    return ISX_OK;
}

extern "C" int isx_bbox_nonzero_static(const uint8_t *img_dev, int h, int w, int c, int64_t pitch,
                                       int32_t *bbox_out_dev, void *stream) {
    ISX_REQUIRE(bbox_out_dev, "null bbox_out_dev");
    ISX_REQUIRE(h <= ISX_BBOX_STATIC_MAX_DIM && w <= ISX_BBOX_STATIC_MAX_DIM,
                "isx_bbox_nonzero_static takes images of up to %d x %d pixels (got %d x %d): use isx_bbox_nonzero",
                ISX_BBOX_STATIC_MAX_DIM, ISX_BBOX_STATIC_MAX_DIM, w, h);
    if (stream) { // the static array of the CURRENT device is used: the stream must belong to it
        int cur = -1;
        hipDevice_t sdev = -1;
        ISX_HIP_TRY(hipGetDevice(&cur));
        ISX_HIP_TRY(hipStreamGetDevice((hipStream_t)stream, &sdev));
        ISX_REQUIRE((int)sdev == cur, "stream belongs to device %d but device %d is current", (int)sdev, cur);
    }
    int32_t *partial = nullptr;
    const int rc = static_scratch_ptr(&partial);
    if (rc != ISX_OK) return rc;
    return isx_bbox_nonzero_launch(img_dev, h, w, c, pitch, bbox_out_dev, partial, (hipStream_t)stream);
}
