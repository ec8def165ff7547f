"""MMDiT block -- same classes / constructor arguments / parameter names as
/root/reference/easyanimate/models/attention.py:1028-1163 and the diffusers `Attention` / `FeedForward`
holders it instantiates.  Arithmetic: HIP kernels only."""
from __future__ import annotations

import inspect
from typing import Optional, Tuple

import torch
from torch import nn

import os

from . import ops
from ._params import f32, gemm_weight, kblocked_weight
from .norm import EasyAnimateLayerNormZero, FP32LayerNorm
from .processor import EasyAnimateAttnProcessor2_0, EasyAnimateSWAttnProcessor2_0

KBLOCKED_FFN = os.environ.get("EA_KBLOCKED_FFN", "1") != "0"   # False: the feed-forward pair on row-major operands (A/B, tests)


class Attention(nn.Module):
    """Parameter holder with the diffusers `Attention` surface used on this path: to_q/to_k/to_v (bias),
    norm_q/norm_k = LayerNorm(dim_head, eps) shared by all heads, to_out = [Linear, Dropout], a pluggable
    processor, and a forward that filters kwargs by the processor's signature."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0, bias: bool = False, qk_norm: Optional[str] = None, eps: float = 1e-5,
                 processor=None, out_bias: bool = True, **unused):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.query_dim = query_dim
        self.heads = heads
        self.dim_head = dim_head
        self.is_cross_attention = cross_attention_dim is not None
        self.scale = dim_head ** -0.5
        if qk_norm is None:
            self.norm_q = None
            self.norm_k = None
        elif qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps)
        else:
            raise ValueError(f"unknown qk_norm: {qk_norm}")
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.set_processor(processor if processor is not None else EasyAnimateAttnProcessor2_0())

    def set_processor(self, processor) -> None:
        self.processor = processor
        self._proc_params = set(inspect.signature(processor.__call__).parameters.keys())

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        kw = {k: v for k, v in cross_attention_kwargs.items() if k in self._proc_params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GELU(nn.Module):
    """diffusers GELU(dim_in, dim_out, approximate): Linear + gelu; parameter name `proj`."""

    def __init__(self, dim_in: int, dim_out: int, approximate: str = "none", bias: bool = True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate


class FeedForward(nn.Module):
    """diffusers FeedForward(dim, activation_fn="gelu-approximate", final_dropout=True): keys net.0.proj.*, net.2.*.
    forward(x, residual=None, gate=None): with residual/gate the second GEMM's epilogue computes
    residual + gate * ff(x) (attention.py:1161-1162)."""

    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, dropout: float = 0.0,
                 activation_fn: str = "geglu", final_dropout: bool = False, inner_dim=None, bias: bool = True):
        super().__init__()
        if inner_dim is None:
            inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        if activation_fn != "gelu-approximate":
            raise NotImplementedError("only activation_fn='gelu-approximate' (the EasyAnimate V5/V5.1 setting) has a HIP epilogue")
        self.net = nn.ModuleList([GELU(dim, inner_dim, approximate="tanh", bias=bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states: torch.Tensor, residual: Optional[torch.Tensor] = None,
                gate: Optional[torch.Tensor] = None, *args, **kwargs) -> torch.Tensor:
        x = hidden_states if hidden_states.dtype == torch.bfloat16 else hidden_states.to(torch.bfloat16)
        fc1, fc2 = self.net[0].proj, self.net[2]
        w1, w2 = gemm_weight(fc1.weight), gemm_weight(fc2.weight)
        if (KBLOCKED_FFN and x.dim() == 3 and x.is_contiguous() and w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16
                and ops.kblocked_ok(x.shape[0], x.shape[1], w1.shape[0], x.shape[2])
                and ops.kblocked_ok(x.shape[0], x.shape[1], w2.shape[0], w1.shape[0])):
            # the inner activation lives K-BLOCKED ([B, inner / 64, M, 64]): the first GEMM's 64-column wave tiles write whole blocks,
            # and the second GEMM's LDS-DMA reads a K tile of 256 rows as one contiguous 32 KiB block instead of 256 pieces 24 KiB
            # apart (inner = 12 288) -- the layout that GEMM's main loop was waiting on; its weight likewise (cached)
            B = x.shape[0]
            h = ops.gemm_kblocked(x, w1, f32(fc1.bias), ops.EPI_BIAS_GELU_TANH, ops.LAYOUT_C)
            w2b = kblocked_weight(fc2.weight)
            if residual is not None:
                return ops.gemm_kblocked(h, w2b, f32(fc2.bias), ops.EPI_BIAS_GATE_RES, ops.LAYOUT_A | ops.LAYOUT_W,
                                         res=residual if residual.is_contiguous() else residual.contiguous(), gate=gate.reshape(B, -1))
            return ops.gemm_kblocked(h, w2b, f32(fc2.bias), ops.EPI_BIAS, ops.LAYOUT_A | ops.LAYOUT_W)
        h = ops.gemm(x, gemm_weight(fc1.weight), f32(fc1.bias), ops.EPI_BIAS_GELU_TANH)
        if residual is not None:
            B = x.shape[0]
            return ops.gemm(h, gemm_weight(fc2.weight), f32(fc2.bias), ops.EPI_BIAS_GATE_RES,
                            res=residual, gate=gate.reshape(B, -1))
        return ops.gemm(h, gemm_weight(fc2.weight), f32(fc2.bias), ops.EPI_BIAS)


class EasyAnimateDiTBlock(nn.Module):
    """reference: easyanimate/models/attention.py:1028-1163."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, time_embed_dim: int,
                 dropout: float = 0.0, activation_fn: str = "gelu-approximate", norm_elementwise_affine: bool = True,
                 norm_eps: float = 1e-6, final_dropout: bool = True, ff_inner_dim: Optional[int] = None,
                 ff_bias: bool = True, qk_norm: bool = True, after_norm: bool = False,
                 norm_type: str = "fp32_layer_norm", is_mmdit_block: bool = True, is_swa: bool = False):
        super().__init__()
        self.norm1 = EasyAnimateLayerNormZero(time_embed_dim, dim, norm_elementwise_affine, norm_eps,
                                              norm_type=norm_type, bias=True)
        self.is_swa = is_swa
        self.attn1 = Attention(query_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads,
                               qk_norm="layer_norm" if qk_norm else None, eps=1e-6, bias=True,
                               processor=EasyAnimateSWAttnProcessor2_0() if is_swa else EasyAnimateAttnProcessor2_0())
        if is_mmdit_block:
            self.attn2 = Attention(query_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads,
                                   qk_norm="layer_norm" if qk_norm else None, eps=1e-6, bias=True,
                                   processor=EasyAnimateSWAttnProcessor2_0() if is_swa else EasyAnimateAttnProcessor2_0())
        else:
            self.attn2 = None
        self.norm2 = EasyAnimateLayerNormZero(time_embed_dim, dim, norm_elementwise_affine, norm_eps,
                                              norm_type=norm_type, bias=True)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout,
                              inner_dim=ff_inner_dim, bias=ff_bias)
        if is_mmdit_block:
            self.txt_ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout,
                                      inner_dim=ff_inner_dim, bias=ff_bias)
        else:
            self.txt_ff = None
        if after_norm:
            self.norm3 = FP32LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        else:
            self.norm3 = None

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor,
                image_rotary_emb: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, num_frames=None, height=None,
                width=None, sp=None) -> Tuple[torch.Tensor, torch.Tensor]:
        # Norm + Attn + gated residual (attention.py:1118-1141), the residual add fused in the out-proj GEMMs
        norm_h, norm_e, gate_msa, enc_gate_msa = self.norm1(hidden_states, encoder_hidden_states, temb)
        hidden_states, encoder_hidden_states = self.attn1(
            hidden_states=norm_h, encoder_hidden_states=norm_e, image_rotary_emb=image_rotary_emb, attn2=self.attn2,
            residual=hidden_states, encoder_residual=encoder_hidden_states, gate=gate_msa, encoder_gate=enc_gate_msa,
            sp=sp, num_frames=num_frames, height=height, width=width)
        # Norm + FFN + gated residual (attention.py:1144-1162), fused in the second FFN GEMM
        norm_h, norm_e, gate_ff, enc_gate_ff = self.norm2(hidden_states, encoder_hidden_states, temb)
        txt_ff = self.txt_ff if self.txt_ff is not None else self.ff
        if self.norm3 is not None:
            # after_norm (attention.py:1150-1155): FP32LayerNorm between the FFN and its gated residual, so the residual
            # cannot ride in the GEMM epilogue: FFN (plain bias epilogue) -> LayerNorm kernel -> gated-residual kernel
            B = hidden_states.shape[0]
            hidden_states = ops.gated_residual(self.norm3(self.ff(norm_h)), hidden_states, gate_ff.reshape(B, -1))
            encoder_hidden_states = ops.gated_residual(self.norm3(txt_ff(norm_e)), encoder_hidden_states, enc_gate_ff.reshape(B, -1))
            return hidden_states, encoder_hidden_states
        hidden_states = self.ff(norm_h, residual=hidden_states, gate=gate_ff)
        encoder_hidden_states = txt_ff(norm_e, residual=encoder_hidden_states, gate=enc_gate_ff)
        return hidden_states, encoder_hidden_states
