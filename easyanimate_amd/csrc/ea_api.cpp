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

// Dispatch counters: which kernel variant served a call (tests assert that the kernels a parity test is meant to
// cover actually ran; host-side bookkeeping only, one relaxed increment per launch).
namespace {
struct Counter { char name[48]; long long n; };
Counter g_counters[64];
int g_ncounters = 0;
const char* g_last = "";
}  // namespace

void ea_count(const char* name) {
    g_last = name;   // string literals of the dispatch sites
    for (int i = 0; i < g_ncounters; ++i)
        if (!strcmp(g_counters[i].name, name)) { ++g_counters[i].n; return; }
    if (g_ncounters < 64) {
        strncpy(g_counters[g_ncounters].name, name, sizeof(g_counters[0].name) - 1);
        g_counters[g_ncounters++].n = 1;
    }
}

extern "C" long long ea_get_counter(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < g_ncounters; ++i)
        if (!strcmp(g_counters[i].name, name)) return g_counters[i].n;
    return 0;
}

extern "C" int ea_counter_name(int index, char* buf, int buf_len) {
    if (index < 0 || index >= g_ncounters || !buf || buf_len <= 0) return EA_ERR_ARG;
    strncpy(buf, g_counters[index].name, (size_t)buf_len - 1);
    buf[buf_len - 1] = 0;
    return EA_OK;
}

extern "C" const char* ea_last_dispatch(void) { return g_last; }

extern "C" void ea_reset_counters(void) {
    for (int i = 0; i < g_ncounters; ++i) g_counters[i].n = 0;
}
extern "C" int ea_version(void) { return 100; }

// Tuning / benchmarking switches.  Each kernel file owns its switch; results never depend on them.
int ea_gemm_tile_set(int v);      // ea_gemm.hip:      0 (auto) | 128 | 256
int ea_gemm_mfma_set(int v);      // ea_gemm.hip: 16 | 32
int ea_conv_mfma_set(int v);      // ea_conv.hip: 16 | 32
int ea_attn_variant_set(int v);   // ea_attention.hip: 1 | 2
int ea_conv_tile_set(int v);
int ea_conv_m512_set(int v);      // ea_conv.hip: 0 | 1      // ea_conv.hip:      0 (auto) | 128 | 256 | 512

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
    else if (!strcmp(name, "conv_m512")) rc = ea_conv_m512_set(value);
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
