"""Norm layers of the MMDiT block -- same classes, constructor arguments and parameter names as
/root/reference/easyanimate/models/norm.py:16-42,135-166; arithmetic in ea_layernorm_modulate_bf16 /
ea_rmsnorm_bf16 / ea_linear_small_m."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import nn

from . import ops
from ._params import bf16_weight, f32


def _as_bf16_3d(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x if x.is_contiguous() else x.contiguous()


class FP32LayerNorm(nn.LayerNorm):
    """reference: norm.py:16-26.  LayerNorm statistics and affine in fp32, bf16 in/out."""

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        x = _as_bf16_3d(inputs.reshape(-1, inputs.shape[-2], inputs.shape[-1]) if inputs.dim() != 3 else inputs)
        w = f32(self.weight) if self.elementwise_affine else None
        b = f32(self.bias) if self.elementwise_affine else None
        y = ops.layernorm_modulate(x, w, b, None, None, self.eps)
        return y.reshape(inputs.shape).to(inputs.dtype)


class EasyAnimateRMSNorm(nn.Module):
    """reference: norm.py:28-42."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        x = _as_bf16_3d(hidden_states)
        return ops.rmsnorm(x, f32(self.weight), self.variance_epsilon)

    def extra_repr(self):
        return f"{tuple(self.weight.shape)}, eps={self.variance_epsilon}"


class EasyAnimateLayerNormZero(nn.Module):
    """reference: norm.py:135-166.  One GEMV builds the fp32 modulation table [B, 6*dim]
    (shift, scale, gate, enc_shift, enc_scale, enc_gate); the LN kernels read scale/shift straight out
    of it and the gates are handed (as views) to the GEMM epilogues."""

    def __init__(self, conditioning_dim: int, embedding_dim: int, elementwise_affine: bool = True, eps: float = 1e-5,
                 bias: bool = True, norm_type: str = "fp32_layer_norm") -> None:
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim, bias=bias)
        if norm_type == "layer_norm":
            self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=elementwise_affine, eps=eps)
        elif norm_type == "fp32_layer_norm":
            self.norm = FP32LayerNorm(embedding_dim, elementwise_affine=elementwise_affine, eps=eps)
        else:
            raise ValueError(
                f"Unsupported `norm_type` ({norm_type}) provided. Supported ones are: 'layer_norm', 'fp32_layer_norm'.")
        self.embedding_dim = embedding_dim

    def modulation_table(self, temb: torch.Tensor) -> torch.Tensor:
        return ops.linear_small_m(temb.float().contiguous(), bf16_weight(self.linear.weight), f32(self.linear.bias),
                                  act_in=1)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        d = self.embedding_dim
        tab = self.modulation_table(temb)  # [B, 6d] fp32
        affine = self.norm.elementwise_affine
        w = f32(self.norm.weight) if affine else None
        b = f32(self.norm.bias) if affine else None
        eps = self.norm.eps
        h = ops.layernorm_modulate(_as_bf16_3d(hidden_states), w, b, tab[:, d:2 * d], tab[:, 0:d], eps)
        e = ops.layernorm_modulate(_as_bf16_3d(encoder_hidden_states), w, b, tab[:, 4 * d:5 * d], tab[:, 3 * d:4 * d], eps)
        return h, e, tab[:, None, 2 * d:3 * d], tab[:, None, 5 * d:6 * d]
