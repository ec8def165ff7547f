// Host-side error plumbing of the C ABI (include/ea_mi355x.h): nothing throws or aborts across it.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ea_mi355x.h"
#ifndef EA_BUILD_VARIANTS
#define EA_BUILD_VARIANTS 0   // see ea_common.h
#endif

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
extern "C" int ea_version(void) { return 114; }   // 114: ea_tile_blend, ea_tile_corner_blend (VAE tiling), options ln_wgs / ln_nt; 113: head groups (ea_qkv_gemm_norm_rope_grouped_bf16, ea_attention_fwd_{range,segments}_heads_bf16), options gemm_w4a / conv_w4a; 112: ea_attention_window_mapped_fwd_bf16, ea_permute_cols_bf16; 111: the default library requires the softmax scale folded into Q (ea_attention_fwd*); 110: K / V^T geometry + parts in the QKV entry points

// Tuning / benchmarking switches.  Each kernel file owns its switch; results never depend on them.
#define EA_OPTION(n) int ea_##n##_set(int v); int ea_##n##_get();
EA_OPTION(gemm_tile)      // ea_gemm.hip:      0 (auto) | 128 | 256
EA_OPTION(gemm_mfma)      // ea_gemm.hip:      16 | 32
EA_OPTION(gemm_w4a)       // ea_gemm.hip:      bits 1 (GEMM) | 2 (fused QKV): four-wave 256 x 256 kernels with the hand-placed main loop; default 3
EA_OPTION(conv_mfma)      // ea_conv.hip:      16 | 32
EA_OPTION(conv_tile)      // ea_conv.hip:      0 (auto) | 128 | 256 | 512 | 1024
EA_OPTION(conv_m512)      // ea_conv.hip:      0 | 1
EA_OPTION(conv_w4a)       // ea_conv.hip:      bits 1 (512 x 128 tiles) | 2 (256 x 256 tiles): four-wave row-slab kernels, hand-placed main loop
EA_OPTION(attn_variant)   // ea_attention.hip: 3 (EA_BUILD_VARIANTS=1 libraries: also 1 | 2)
EA_OPTION(attn_stages)    // ea_attention.hip: 2 | 3 LDS stages per operand of the v3 kernel (3: K / V^T tiles requested one tile earlier)
EA_OPTION(attn_nw)        // ea_attention.hip: 4 | 8 waves per workgroup of the v3 kernel (8: one 512-query workgroup per CU, one K / V^T stream)
EA_OPTION(ln_wgs)         // ea_norm.hip:      workgroups per batch element of the sweeping LayerNorm-modulate kernel (default 1024; 0 = the 32-rows-per-workgroup kernel)
EA_OPTION(ln_nt)          // ea_norm.hip:      bits 1 (streaming loads) | 2 (streaming stores) of the sweeping kernel; default 3
#undef EA_OPTION
// read-only: was this library built with EA_BUILD_VARIANTS=1 (the cross-check kernel generations are present)?
int ea_build_variants_get() { return EA_BUILD_VARIANTS; }
int ea_build_variants_set(int v) { return v == EA_BUILD_VARIANTS ? 0 : -1; }
namespace {
struct Option { const char* name; int (*set)(int); int (*get)(); };
#define EA_OPTION(n) {#n, ea_##n##_set, ea_##n##_get}
const Option g_options[] = {EA_OPTION(gemm_tile), EA_OPTION(gemm_mfma), EA_OPTION(gemm_w4a), EA_OPTION(conv_mfma),
                            EA_OPTION(conv_tile), EA_OPTION(conv_m512), EA_OPTION(conv_w4a), EA_OPTION(attn_variant), EA_OPTION(attn_nw), EA_OPTION(attn_stages), EA_OPTION(ln_wgs), EA_OPTION(ln_nt), EA_OPTION(build_variants)};
#undef EA_OPTION
const Option* find_option(const char* name) {
    for (const Option& o : g_options)
        if (name && !strcmp(name, o.name)) return &o;
    return nullptr;
}
}  // namespace

extern "C" int ea_get_option(const char* name, int* value) {
    const Option* o = find_option(name);
    if (!o || !value) {
        ea_set_error("ea_get_option: unknown option '%s' or null value", name ? name : "(null)");
        return EA_ERR_ARG;
    }
    *value = o->get();
    return EA_OK;
}

extern "C" int ea_set_option(const char* name, int value) {
    const Option* o = find_option(name);
    if (!o) {
        ea_set_error("ea_set_option: unknown option '%s'", name ? name : "(null)");
        return EA_ERR_ARG;
    }
    const int rc = o->set(value);
    if (rc != 0) {
        ea_set_error("ea_set_option: value %d is not valid for '%s'", value, name);
        return EA_ERR_ARG;
    }
    return EA_OK;
}
