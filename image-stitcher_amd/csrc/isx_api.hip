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

extern "C" int isx_version(void) { return 1; }

extern "C" const char *isx_last_error(void) { return g_err; }

extern "C" int isx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
