// Host-side error plumbing of the C ABI (include/ea_mi355x.h): nothing throws or aborts across it.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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

// Tuning / benchmarking switches.  Each kernel file owns its switch; results never depend on them.
int ea_gemm_tile_set(int v);      // ea_gemm.hip:      0 (auto) | 128 | 256
int ea_gemm_mfma_set(int v);      // ea_gemm.hip: 16 | 32
int ea_conv_mfma_set(int v);      // ea_conv.hip: 16 | 32
int ea_attn_variant_set(int v);   // ea_attention.hip: 1 | 2
int ea_conv_tile_set(int v);      // ea_conv.hip:      0 (auto) | 128 | 256 | 512

extern "C" int ea_set_option(const char* name, int value) {
    if (!name) {
        ea_set_error("ea_set_option: null name");
        return EA_ERR_ARG;
    }
    int rc;
    if (!strcmp(name, "gemm_tile")) rc = ea_gemm_tile_set(value);
    else if (!strcmp(name, "gemm_mfma")) rc = ea_gemm_mfma_set(value);
    else if (!strcmp(name, "attn_variant")) rc = ea_attn_variant_set(value);
    else if (!strcmp(name, "conv_tile")) rc = ea_conv_tile_set(value);
    else if (!strcmp(name, "conv_mfma")) rc = ea_conv_mfma_set(value);
    else {
        ea_set_error("ea_set_option: unknown option '%s'", name);
        return EA_ERR_ARG;
    }
    if (rc != 0) {
        ea_set_error("ea_set_option: value %d is not valid for '%s'", value, name);
        return EA_ERR_ARG;
    }
    return EA_OK;
}
