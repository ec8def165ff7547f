"""ctypes binding of libea_mi355x.so (C ABI declared in include/ea_mi355x.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so) and loads it RTLD_GLOBAL.  It must be in the
# process BEFORE libea_mi355x.so is dlopen'ed so that the kernels register with, and launch through, the same
# runtime that owns torch's streams and allocations (otherwise: "no ROCm-capable device is detected").
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# EA_LIB_PATH: load another build of the same ABI (compiler-flag A/B runs: tools/build_variants.sh)
LIB_PATH = os.environ.get("EA_LIB_PATH") or os.path.join(_HERE, "lib", "libea_mi355x.so")

_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float

# name -> argtypes, exactly as include/ea_mi355x.h
PROTOTYPES = {
    "ea_layernorm_modulate_bf16": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _L, _L, _F, _P],
    "ea_rmsnorm_bf16": [_P, _P, _P, _I, _I, _F, _P],
    "ea_linear_small_m": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ea_timestep_sinusoid": [_P, _P, _I, _I, _I, _P],
    "ea_gemm_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _I, _P],
    "ea_gemm_bf16_kblocked": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _I, _I, _P],
    "ea_gemm_bf16_w8": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _I, _P],
    "ea_qknorm_rope_bf16": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "ea_qkv_gemm_norm_rope_bf16": [_P] * 16 + [_I, _I, _I, _I, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P],
    "ea_qkv_gemm_norm_rope_grouped_bf16": [_P] * 16 + [_I, _I, _I, _I, _L, _L, _I, _I, _I, _I, _I, _I, _L, _F, _F, _P],
    "ea_qkv_gemm_norm_rope_bf16_w8": [_P] * 16 + [_I, _I, _I, _I, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P],
    "ea_attention_fwd_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _F, _P],
    "ea_attention_fwd_segments_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _I, _F, _P, _I, _P],
    "ea_attention_fwd_segments_heads_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _I, _F, _P, _I, _I, _I, _L, _P],
    "ea_attention_fwd_range_heads_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I, _I, _L, _P],
    "ea_attention_d512_fwd_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _F, _P],
    "ea_attention_window_fwd_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P],
    "ea_attention_window_mapped_fwd_bf16": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _I, _F, _P],
    "ea_permute_cols_bf16": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ea_attention_fwd_range_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P],
    "ea_patchify": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ea_unpatchify": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ea_cfg_euler_step": [_P, _P, _L, _F, _F, _I, _I, _P],
    "ea_cfg_rescale_euler_step": [_P, _P, _L, _F, _F, _F, _P, _I, _P, _I, _P],
    "ea_teacache_rel_l1_bf16": [_P, _P, _L, _P, _I, _P, _P],
    "ea_bf16_binary": [_P, _P, _P, _L, _I, _P],
    "ea_gated_residual_bf16": [_P, _P, _P, _P, _I, _L, _I, _L, _P],
    "ea_rope_half_scatter_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _L, _P],
    "ea_silu_mul_bf16": [_P, _P, _P, _L, _I, _L, _L, _P],
    "ea_attention_causal_gqa_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "ea_tile_blend": [_P, _P, _I, _L, _I, _I, _L, _I, _L, _P],
    "ea_tile_corner_blend": [_P, _P, _I, _L, _I, _I, _I, _I, _P],
    "ea_conv3d_cl_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ea_conv3d_cl_stats_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P],
    "ea_conv3d_cl_subpixel_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P],
    "ea_groupnorm_finalize_bf16": [_P, _P, _I, _L, _I, _I, _I, _F, _P],
    "ea_conv3d_tap_gather_f32": [_P, _P, _P, _I, _I, _I, _L, _I, _I, _P],
    "ea_im2col3d_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ea_groupnorm_stats_bf16": [_P, _P, _P, _I, _L, _I, _I, _I, _F, _P],
    "ea_groupnorm_apply_bf16": [_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P],
    "ea_softmax_rows_bf16": [_P, _P, _L, _I, _F, _P],
    "ea_softmax_rows_f32in": [_P, _P, _L, _I, _F, _P],
    "ea_ncdhw_to_ndhwc": [_P, _P, _I, _I, _L, _I, _P],
    "ea_ndhwc_to_ncdhw": [_P, _P, _I, _I, _L, _I, _I, _P],
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m easyanimate_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.ea_last_error_string.restype = ctypes.c_char_p
    lib.ea_last_error_string.argtypes = []
    lib.ea_version.restype = c_int
    lib.ea_version.argtypes = []
    lib.ea_attention_state_bytes.restype = c_int64
    lib.ea_attention_state_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.ea_conv3d_cl_tmerge_ok.restype = c_int
    lib.ea_conv3d_cl_tmerge_ok.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.ea_conv3d_cl_blocked_ok.restype = c_int
    lib.ea_conv3d_cl_blocked_ok.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.ea_set_option.restype = c_int
    lib.ea_set_option.argtypes = [ctypes.c_char_p, c_int]
    lib.ea_get_option.restype = c_int
    lib.ea_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(c_int)]
    lib.ea_get_counter.restype = ctypes.c_longlong
    lib.ea_get_counter.argtypes = [ctypes.c_char_p]
    lib.ea_counter_name.restype = c_int
    lib.ea_counter_name.argtypes = [c_int, ctypes.c_char_p, c_int]
    lib.ea_last_dispatch.restype = ctypes.c_char_p
    lib.ea_last_dispatch.argtypes = []
    lib.ea_reset_counters.restype = None
    lib.ea_reset_counters.argtypes = []
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.ea_last_error_string().decode()}")


def set_option(name: str, value: int) -> None:
    """ea_set_option: tuning / benchmarking switches (e.g. "gemm_tile" 0|128|256)."""
    call("ea_set_option", name.encode(), int(value))


def get_option(name: str) -> int:
    """ea_get_option: the current value of a tuning switch."""
    v = ctypes.c_int(0)
    call("ea_get_option", name.encode(), ctypes.byref(v))
    return v.value


def reset_counters() -> None:
    load().ea_reset_counters()


def counters() -> dict:
    """ea_get_counter / ea_counter_name: {kernel variant: launches since the last reset} (non-zero entries)."""
    lib = load()
    out, buf, i = {}, ctypes.create_string_buffer(64), 0
    while lib.ea_counter_name(i, buf, 64) == 0:
        n = lib.ea_get_counter(buf.value)
        if n:
            out[buf.value.decode()] = int(n)
        i += 1
    return out


def last_dispatch() -> str:
    """Name of the kernel variant the most recent counted launch used (ea_last_dispatch)."""
    return load().ea_last_dispatch().decode()
