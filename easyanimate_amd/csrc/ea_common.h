// Shared device/host helpers for libea_mi355x (gfx950 only; wave = 64 lanes).
#pragma once
// EA_BUILD_VARIANTS=1 (python -m easyanimate_amd.build with EA_BUILD_VARIANTS=1 in the environment) also compiles the kernel
// generations that no product call reaches any more -- attention v1 over plain keys, the 32x32x16 row-slab convolution, the
// four-wave GEMM experiment -- and lets ea_set_option select them; they exist as bitwise / numerical cross-checks of the
// kernels that replaced them.  The default library carries one kernel per job.
#ifndef EA_BUILD_VARIANTS
#define EA_BUILD_VARIANTS 0
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ea_mi355x.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define EA_WAVE 64

// ---- error plumbing (host) --------------------------------------------------------------------
void ea_set_error(const char* fmt, ...);
int ea_check_launch(const char* what);
void ea_count(const char* name);   // dispatch counter (ea_get_counter)

#define EA_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            ea_set_error(__VA_ARGS__); \
            return EA_ERR_ARG;       \
        }                            \
    } while (0)

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, ((unsigned int)b) << 16);
}
// round-to-nearest-even, NaN preserved (same as torch's float->bfloat16)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (bf16_t)f);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// tanh-approximate GELU as torch defines it, 0.5*x*(1+tanh(u)) with u = sqrt(2/pi)*(x+0.044715x^3), evaluated as
// x * sigmoid(2u) = x / (1 + 2^(-2u*log2(e))): 5 plain VALU + v_exp_f32 + v_rcp_f32 (the GEMM epilogue runs it on
// 128 values per thread with no MFMA work left to hide it under).
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;   // -2*sqrt(2/pi)*log2(e)
    const float c1 = c0 * 0.044715f;
    const float a = x * __builtin_fmaf(x * x, c1, c0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
