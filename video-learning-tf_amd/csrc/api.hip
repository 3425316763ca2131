// Error convention and library-level queries of libvltf_hip.so (include/vltf.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[1024] = "";

void vl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vl_last_error(void) { return g_err; }

extern "C" int vl_version(void) { return 1; }

extern "C" int vl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        vl_set_error("hipGetDeviceCount failed");
        return -1;
    }
    return n;
}
