// isx_ctx: the per-device workspace handle of include/isx.h (private layout, shared by the frame sources).
#pragma once
#include "isx_common.h"

struct isx_ctx {
    int device;
    uint8_t *wbuf;
    size_t wbuf_cap;
    int32_t *rows; // gradient: qh x int4 {GL, GR, L0, R0} (gray != 0 extent, channel-0 != 0 extent; canvas x)
    size_t rows_cap;
    int32_t *cols; // 2 x qw {top, bottom}
    size_t cols_cap;
    uint64_t *mask; // qh x nchunk x 4 ballots
    size_t mask_cap;
    int32_t *out_dev; // R_INTS result ints, then the table folded on the host
    size_t out_cap;
    int32_t *out_host; // pinned mirror of out_dev
    size_t out_host_cap;
    int pending;    // kernels queued whose results have not been read back: 1 = gradient, 2 = alpha
    int fold_n;     // entries of the table behind the result block (gradient: workgroups of K3; alpha: rows of Q)
    int fold_qy0;   // alpha: canvas y of the table's first row
    int32_t last[ISX_FRAME_RESULT_INTS];
    // staging of the composite host calls (isx_frame_host.hip): one device block and its pinned mirror
    uint8_t *hs_dev;
    size_t hs_dev_cap;
    uint8_t *hs_host;
    size_t hs_host_cap;
};

template <typename T>
static inline int ctx_ensure(T **p, size_t *cap, size_t need) {
    if (*cap >= need) return ISX_OK;
    if (*p) ISX_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 2;
    ISX_HIP_TRY(hipMalloc((void **)p, want * sizeof(T)));
    *cap = want;
    return ISX_OK;
}

static inline int ctx_ensure_out(isx_ctx *c, size_t ints) {
    int rc = ctx_ensure(&c->out_dev, &c->out_cap, ints);
    if (rc != ISX_OK) return rc;
    if (c->out_host_cap < c->out_cap) {
        if (c->out_host) ISX_HIP_TRY(hipHostFree(c->out_host));
        c->out_host = nullptr;
        c->out_host_cap = 0;
        ISX_HIP_TRY(hipHostMalloc((void **)&c->out_host, c->out_cap * 4, hipHostMallocDefault));
        c->out_host_cap = c->out_cap;
    }
    return ISX_OK;
}


// device block + pinned mirror of at least `bytes` for the composite host calls
static inline int ctx_ensure_staging(isx_ctx *c, size_t bytes) {
    int rc = ctx_ensure(&c->hs_dev, &c->hs_dev_cap, bytes);
    if (rc != ISX_OK) return rc;
    if (c->hs_host_cap < c->hs_dev_cap) {
        if (c->hs_host) ISX_HIP_TRY(hipHostFree(c->hs_host));
        c->hs_host = nullptr;
        c->hs_host_cap = 0;
        ISX_HIP_TRY(hipHostMalloc((void **)&c->hs_host, c->hs_dev_cap, hipHostMallocDefault));
        c->hs_host_cap = c->hs_dev_cap;
    }
    return ISX_OK;
}
