"""Derived-parameter cache: the kernels want fp32 biases/norm affines and bf16 GEMM weights (optionally
K-padded).  Derived copies are cached per parameter and invalidated when the parameter is modified in place
(`_version`), re-pointed (`data_ptr`) or moved, so load_state_dict / LoRA merges are observed."""
from __future__ import annotations

import os
import weakref
from typing import Callable, Dict, Tuple

import torch

_cache: Dict[Tuple[int, str], Tuple[tuple, torch.Tensor]] = {}
_finalized: set = set()      # (id(parameter), tag) keys that already carry a weakref.finalize


def _key(p: torch.Tensor):
    return (p.data_ptr(), p._version, p.device, p.dtype, tuple(p.shape))


def derived(p: torch.Tensor, tag: str, fn: Callable[[torch.Tensor], torch.Tensor]) -> torch.Tensor:
    k = (id(p), tag)
    ent = _cache.get(k)
    sig = _key(p)
    if ent is not None and ent[0] == sig:
        return ent[1]
    with torch.no_grad():
        val = fn(p.detach())
    _cache[k] = (sig, val)
    if k not in _finalized:      # ONE finalizer per (parameter, tag) for the parameter's lifetime: every pipeline call clears the
        try:                     # cache (invalidate_weight_cache) and recomputes -- it must not also add a finalizer each time
            weakref.finalize(p, _drop, k)
            _finalized.add(k)
        except TypeError:
            pass
    return val


def _drop(k) -> None:
    _cache.pop(k, None)
    _finalized.discard(k)


def invalidate_weight_cache() -> None:
    """Drop every derived copy.  Needed only after writes the version counter cannot see, i.e. `param.data.add_(...)`
    style updates (the reference's utils/lora_utils.py:merge_lora / unmerge_lora) on parameters that are NOT used in
    place -- contiguous 2-D bf16 Linear weights are read directly and need nothing.  One derived copy of a bf16 Linear weight
    exists: the K-blocked form of ff.net.2 (kblocked_weight); FeedForward refreshes it on every un-frozen call
    (weights_frozen), so a forward -> `.data` merge -> forward sequence on the bare transformer sees the merged weight too."""
    _cache.clear()


def drop_tag(tag: str) -> None:
    """Drop the derived copies of one kind (e.g. "kblock")."""
    for k in [k for k in _cache if k[1] == tag]:
        del _cache[k]


_frozen = 0


class weights_frozen:
    """`with weights_frozen():` -- the caller promises that no parameter is written inside (a sampling loop).  Outside of it
    FeedForward.forward re-derives the K-blocked copy of its ff.net.2 weight on each call (kblocked_weight: 2 x 75 MB of traffic = about
    30 us per use, 96 uses = about 3 ms per forward at the 12B size, 0.1 % of a config-3 step), because a `weight.data += delta` LoRA merge (utils/lora_utils.py:369-433) between two forwards is invisible
    to the version counter; the pipelines' loops run frozen (they drop every derived copy once per call instead)."""

    def __enter__(self):
        global _frozen
        _frozen += 1
        return self

    def __exit__(self, *exc):
        global _frozen
        _frozen -= 1
        return False


def frozen() -> bool:
    return _frozen > 0


def f32(p):
    """fp32 contiguous view/copy of a parameter (bias, LayerNorm affine)."""
    if p is None:
        return None
    if p.dtype == torch.float32 and p.is_contiguous():
        return p.detach()
    return derived(p, "f32", lambda t: t.float().contiguous())


# fp8 weight storage (the reference's `model_cpu_offload_and_qfloat8`): True = the block GEMMs read the fp8 parameter itself
# (ea_gemm_bf16_w8 / ea_qkv_gemm_norm_rope_bf16_w8 widen it inside the kernel: one byte per weight in HBM, no bf16 copy);
# False = up-cast once into the derived-parameter cache (two more bytes per weight, the bf16 kernels).  Same results.
FP8_NATIVE_GEMM = os.environ.get("EA_FP8_NATIVE_GEMM", "1") != "0"


def gemm_weight(p):
    """The [N, K] weight operand of ops.gemm / ops.qkv_gemm_norm_rope for a Linear: the parameter itself when it is
    contiguous bf16 or (fp8 storage mode, FP8_NATIVE_GEMM) contiguous float8_e4m3fn with K % 64 == 0, else a cached bf16 copy."""
    if FP8_NATIVE_GEMM and p.dtype == torch.float8_e4m3fn and p.dim() == 2 and p.is_contiguous() and p.shape[1] % 64 == 0:
        return p.detach()
    return bf16_weight(p)


def bf16_weight(p, k_pad: int | None = None):
    """bf16 [N, K(_pad)] GEMM weight; the parameter itself when it already is contiguous bf16."""
    def make(t):
        t2 = t.reshape(t.shape[0], -1).to(torch.bfloat16)
        if k_pad is not None and k_pad != t2.shape[1]:
            t2 = torch.nn.functional.pad(t2, (0, k_pad - t2.shape[1]))
        return t2.contiguous()
    if p.dtype == torch.bfloat16 and p.dim() == 2 and p.is_contiguous() and (k_pad is None or k_pad == p.shape[1]):
        return p.detach()
    return derived(p, f"bf16w{k_pad}", make)


def kblocked_weight(p):
    """The K-blocked copy [K / 64, N, 64] of a bf16 Linear weight [N, K] (ops.gemm_kblocked; cached like every derived weight:
    load_state_dict / in-place writes are seen through the version counter).  `.data` writes are not, so outside a
    weights_frozen() region the copy is re-derived IN PLACE on every use (2 x 75 MB of traffic = about 30 us at the 12B size; 96 uses =
    about 3 ms per forward -- the pipelines' loops, bench.py included, run frozen and pay it once per call): 2 more bytes per weight for
    the layers that use it."""
    make = lambda t: t.to(torch.bfloat16).reshape(t.shape[0], t.shape[1] // 64, 64).permute(1, 0, 2)
    ent = _cache.get((id(p), "kblock"))
    val = derived(p, "kblock", lambda t: make(t).contiguous())
    if ent is not None and ent[1] is val and not frozen():
        with torch.no_grad():
            val.copy_(make(p.detach()))
    return val
