// Host-side error plumbing of the C ABI (include/ea_mi355x.h): nothing throws or aborts across it.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ea_mi355x.h"

static thread_local char g_err[512] = "no error";

void ea_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ea_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ea_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return EA_OK;
}

extern "C" const char* ea_last_error_string(void) { return g_err; }
extern "C" int ea_version(void) { return 100; }
