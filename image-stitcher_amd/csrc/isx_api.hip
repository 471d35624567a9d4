// libisx_hip.so -- error plumbing and library-level entry points.
#include "isx_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int isx_fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

extern "C" int isx_version(void) { return ISX_ABI_VERSION; }

extern "C" const char *isx_last_error(void) { return g_err; }

extern "C" int isx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// HOST.  The union-find of ArbitraryStitcher.get_collections (reference stitcher.py:283-296 over
// utils/structs.py:16-42 DynamicConnectivity): weighted quick-union with path halving; union(p, q) hangs
// the smaller tree under the larger one and, on equal sizes, q's root under p's root.  The caller passes
// the ordered pairs that reached min_match_count in the reference's visiting order; parent/size come back
// in the state the reference's object would be in.
static int uf_find(int32_t *parent, int p) {
    while (parent[p] != p) {
        parent[p] = parent[parent[p]];
        p = parent[p];
    }
    return p;
}

extern "C" int isx_union_find(int n, const int32_t *first_host, const int32_t *second_host, int64_t m,
                              int32_t *parent_io_host, int32_t *size_io_host) {
    ISX_REQUIRE(n >= 0 && m >= 0, "negative size");
    ISX_REQUIRE(parent_io_host && size_io_host && (m == 0 || (first_host && second_host)), "null pointer");
    for (int64_t e = 0; e < m; e++) {
        const int p = first_host[e], q = second_host[e];
        ISX_REQUIRE(p >= 0 && p < n && q >= 0 && q < n, "pair %lld = (%d, %d) outside 0..%d", (long long)e, p, q, n - 1);
        int a = uf_find(parent_io_host, p), b = uf_find(parent_io_host, q);
        if (a == b) continue;
        if (size_io_host[a] < size_io_host[b]) {
            const int t = a;
            a = b;
            b = t;
        }
        parent_io_host[b] = a;
        size_io_host[a] += size_io_host[b];
    }
    return ISX_OK;
}
