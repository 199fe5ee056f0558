// abi.hip -- error reporting / versioning shared by every entry point of libstraps_hip.so
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void straps_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int straps_abi_version(void) { return STRAPS_ABI_VERSION; }
extern "C" const char* straps_last_error(void) { return g_err; }
extern "C" int straps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
