"""Seeded synthetic weights (there are no checkpoints offline): per-tensor generators keyed by the
state-dict name, so the product model, the oracle restatement and the shim-hosted reference all get
bit-identical values independent of construction order (SURVEY.md 8d "value distributions / seeds")."""
from __future__ import annotations

import concurrent.futures as cf
import contextlib
import math
import os
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, style: str = "default") -> torch.Tensor:
    """PyTorch-default-like init: weights U(-1/sqrt(fan_in), +), biases likewise; norm gains 1+N(0,.02).
    style="stress": adaLN linears x4 and norm affines perturbed x5 so that gates/scales are O(1) and kernel
    errors are visible at the output (SURVEY 8d noise-floor finding).  A "_bf16" suffix rounds the values to bf16."""
    if style.endswith("_bf16"):
        # bf16-representable values (what a released bf16 checkpoint holds): the fp32 reference and the bf16 product
        # then really do start from IDENTICAL weights
        return synth_tensor(name, shape, seed, style[:-5]).to(torch.bfloat16).to(torch.float32)
    g = _gen(seed, name)
    is_bias = name.endswith(".bias")
    is_norm = any(t in name for t in (".norm.", "norm_q", "norm_k", "norm_final", ".norm1.", ".norm2.", "group_norm",
                                      "conv_norm_out", "text_proj.0")) and len(shape) == 1 and "linear" not in name
    if is_norm:
        amp = 0.1 if style == "stress" else 0.02
        base = 0.0 if is_bias else 1.0
        return base + amp * torch.randn(shape, generator=g)
    if len(shape) == 1:
        # bias of a linear/conv: bound by the fan_in of its weight is unknown here -> small uniform
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.02
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    bound = 1.0 / math.sqrt(fan_in)
    w = (torch.rand(shape, generator=g) * 2 - 1) * bound
    if style == "stress" and (".norm1.linear" in name or ".norm2.linear" in name or "norm_out.linear" in name):
        w = w * 4.0
    return w


_POOL = None


def _pool() -> cf.ThreadPoolExecutor:
    """Every tensor has its own generator (keyed by name), so tensors can be generated concurrently with identical results;
    torch.rand / randn release the GIL.  (A 7B state dict: 35 s on one core, a few seconds on eight.)"""
    global _POOL
    if _POOL is None:
        _POOL = cf.ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) - 1)))
    return _POOL


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, style: str = "default") -> Dict[str, torch.Tensor]:
    keys = list(shapes)
    vals = list(_pool().map(lambda k: synth_tensor(k, tuple(shapes[k]), seed, style), keys))
    return dict(zip(keys, vals))


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0, style: str = "default", chunk_device=None) -> None:
    """In-place synthetic init of every parameter/buffer of `module` (on whatever device it lives); generation runs a few tensors
    ahead of the copies in a thread pool, at most 16 tensors in flight."""
    items = [(n, p) for n, p in list(module.named_parameters()) + list(module.named_buffers()) if p.dtype.is_floating_point]
    window = 16
    futs = {}
    for i in range(min(window, len(items))):
        futs[i] = _pool().submit(synth_tensor, items[i][0], tuple(items[i][1].shape), seed, style)
    for i, (name, p) in enumerate(items):
        t = futs.pop(i).result()
        if i + window < len(items):
            n2, p2 = items[i + window]
            futs[i + window] = _pool().submit(synth_tensor, n2, tuple(p2.shape), seed, style)
        p.copy_(t.to(device=p.device, dtype=p.dtype))


@contextlib.contextmanager
def skip_init():
    """Construct nn.Linear / Conv / LayerNorm modules WITHOUT their default random initialisation (the synthetic fill overwrites
    every value anyway): `with skip_init(): m = Model.from_config(cfg)` -- seconds instead of half a minute at 7B."""
    saved = {}
    for cls in (torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.LayerNorm, torch.nn.GroupNorm, torch.nn.Embedding):
        saved[cls] = cls.reset_parameters
        cls.reset_parameters = lambda self: None
    try:
        yield
    finally:
        for cls, fn in saved.items():
            cls.reset_parameters = fn
