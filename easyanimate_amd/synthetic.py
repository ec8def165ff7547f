"""Seeded synthetic weights (there are no checkpoints offline): per-tensor generators keyed by the
state-dict name, so the product model, the oracle restatement and the shim-hosted reference all get
bit-identical values independent of construction order (SURVEY.md 8d "value distributions / seeds")."""
from __future__ import annotations

import math
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


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, style: str = "default") -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, tuple(v), seed, style) for k, v in shapes.items()}


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0, style: str = "default", chunk_device=None) -> None:
    """In-place synthetic init of every parameter/buffer of `module` (on whatever device it lives)."""
    for name, p in list(module.named_parameters()) + list(module.named_buffers()):
        if not p.dtype.is_floating_point:
            continue
        t = synth_tensor(name, tuple(p.shape), seed, style)
        p.copy_(t.to(device=p.device, dtype=p.dtype))
