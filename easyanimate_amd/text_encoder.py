"""The text-encoder step on the device library (second half).

Reference call site: easyanimate/pipeline/pipeline_easyanimate.py:438-447 --

    prompt_embeds = text_encoder(input_ids=..., attention_mask=..., output_hidden_states=True).hidden_states[-2]

with `Qwen2VLForConditionalGeneration` (Qwen2-VL-7B) in the slot (predict_t2v.py:205-214): a decoder-only LLM run ONCE per
pipeline call over 256 right-padded prompt tokens (~7e12 FLOP against 50 x 4.5e15 for the loop).  The encoder itself is a
`transformers` dependency of the reference, not code in /root/reference; its decoder layer is restated here from transformers'
Qwen2-VL / Qwen2 modelling code (pinned by tests/test_text_encoder_gpu.py against the installed transformers implementation):

    x -> RMSNorm -> q / k / v Linear (bias) -> rotate-half RoPE (M-RoPE with text-only positions = the 1-D form) -> causal,
    key-padding-masked grouped-query attention -> o Linear -> + x -> RMSNorm -> down(SiLU(gate) * up) -> +

`Qwen2VLTextEncoderHIP(hf_model)` wraps the loaded transformers model WITHOUT copying its weights and answers the one call the
pipelines make: `enc(input_ids=, attention_mask=, output_hidden_states=True).hidden_states[-2]`, plus `.device` / `.dtype`.
Linear layers run on ea_gemm_bf16 (residual adds in its epilogue), norms on ea_rmsnorm_bf16, the rest on ea_text.hip.
Text-only: image / video inputs are refused (the EasyAnimate pipelines never pass any).  There is no CPU fallback."""
from __future__ import annotations

import types
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._params import f32, gemm_weight


def _find_text_model(hf_model):
    """The decoder stack of a transformers Qwen2-VL / Qwen2 model, whatever the library version calls the path to it
    (4.4x: model; >= 4.52 / 5.x: model.language_model)."""
    cands = [hf_model]
    for path in ("model.language_model", "model", "language_model"):
        m = hf_model
        for a in path.split("."):
            m = getattr(m, a, None)
            if m is None:
                break
        if m is not None:
            cands.append(m)
    for m in cands:
        if hasattr(m, "layers") and hasattr(m, "embed_tokens") and hasattr(m, "norm"):
            return m
    raise TypeError(f"{type(hf_model).__name__}: no decoder stack (layers / embed_tokens / norm) found")


class Qwen2VLTextEncoderHIP(nn.Module):
    def __init__(self, hf_model, padded_positions: str = "arange"):
        """padded_positions: the rotary position of the right-PADDED slots (real tokens are 0 .. n-1 either way; the DiT consumes
        the padded rows too, transformer3d.py:1502).  "arange" = transformers 5.x (the installed version: text-only prompts get
        the text model's own arange(S)); "cumsum_fill1" = transformers 4.46 - 4.5x (get_rope_index: cumsum(mask) - 1, padding -> 1)."""
        super().__init__()
        if padded_positions not in ("arange", "cumsum_fill1"):
            raise ValueError(f"padded_positions must be 'arange' or 'cumsum_fill1', not {padded_positions!r}")
        self.padded_positions = padded_positions
        self.hf = hf_model                                 # (registered: .to() / offload hooks reach the shared parameters)
        text = _find_text_model(hf_model)
        cfg = text.config
        self.hidden = cfg.hidden_size
        self.q_heads = cfg.num_attention_heads
        self.kv_heads = getattr(cfg, "num_key_value_heads", None) or self.q_heads
        self.head_dim = getattr(cfg, "head_dim", None) or self.hidden // self.q_heads
        self.eps = cfg.rms_norm_eps
        rp = getattr(cfg, "rope_parameters", None) or getattr(cfg, "rope_scaling", None) or {}
        self.theta = float(rp.get("rope_theta", None) or getattr(cfg, "rope_theta", None) or 10000.0)
        if self.head_dim not in (64, 128):
            raise NotImplementedError(f"Qwen2VLTextEncoderHIP: head_dim {self.head_dim} (ea_attention_causal_gqa_bf16 is built for 64 and 128)")
        if getattr(cfg, "use_sliding_window", False):
            raise NotImplementedError("Qwen2VLTextEncoderHIP: sliding-window layers are not implemented (Qwen2-VL-7B does not use them)")
        if getattr(cfg, "hidden_act", "silu") != "silu":
            raise NotImplementedError(f"Qwen2VLTextEncoderHIP: hidden_act {cfg.hidden_act!r}")
        self._text = [text]                                # (a list: not registered a second time)
        self._ones = {}

    # ---- what the pipelines read -------------------------------------------------------------
    @property
    def device(self):
        return next(self.hf.parameters()).device

    @property
    def dtype(self):
        return next(self.hf.parameters()).dtype

    @property
    def config(self):
        return self.hf.config

    def _gate_ones(self, n: int, device) -> torch.Tensor:
        key = (n, str(device))
        if key not in self._ones:
            self._ones[key] = torch.ones(n, dtype=torch.float32, device=device)
        return self._ones[key]

    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                output_hidden_states: bool = False, pixel_values=None, pixel_values_videos=None, inputs_embeds=None, **kwargs):
        if pixel_values is not None or pixel_values_videos is not None:
            raise NotImplementedError("Qwen2VLTextEncoderHIP is text-only (the EasyAnimate pipelines pass prompts only)")
        text = self._text[0]
        if inputs_embeds is None:
            if input_ids is None:
                raise ValueError("input_ids or inputs_embeds")
            inputs_embeds = text.embed_tokens(input_ids)
        x = inputs_embeds
        if not x.is_cuda:
            raise RuntimeError("Qwen2VLTextEncoderHIP: inputs must be on the GPU; there is no CPU fallback")
        if x.dtype != torch.bfloat16:
            raise RuntimeError(f"Qwen2VLTextEncoderHIP: the encoder must be loaded in bfloat16 (got {x.dtype}); the library computes in bf16")
        B, S, d = x.shape
        dev = x.device
        D, Hq, Hkv = self.head_dim, self.q_heads, self.kv_heads
        # ---- positions and key padding (the three M-RoPE axes carry the same ids for text, so the sectioned cos / sin equal the
        # 1-D ones; padded slots: see __init__)
        if attention_mask is not None:
            am = attention_mask.to(device=dev, dtype=torch.long)
            if not bool((am[:, 1:] <= am[:, :-1]).all()):
                raise NotImplementedError("Qwen2VLTextEncoderHIP: only right-padded prompts (padding_side='right', pipeline_easyanimate.py:431)")
            if self.padded_positions == "arange":
                pos = torch.arange(S, device=dev)[None].expand(B, S)
            else:
                pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
            valid = am.sum(-1).to(torch.int32).contiguous()
        else:
            pos = torch.arange(S, device=dev)[None].expand(B, S)
            valid = None
        inv_freq = 1.0 / (self.theta ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
        ang = pos.to(torch.float32)[..., None] * inv_freq                     # [B, S, D/2]
        ang = torch.cat([ang, ang], dim=-1).reshape(B * S, D)
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        s_pad = ops.round_up(S, 32)
        vt = torch.zeros(B, Hkv, D, s_pad, dtype=torch.bfloat16, device=dev)
        h = x.reshape(B * S, d).contiguous()
        hidden_states = [x]
        for layer in text.layers:
            att, mlp = layer.self_attn, layer.mlp
            n1 = ops.rmsnorm(h, f32(layer.input_layernorm.weight), self.eps)
            q = ops.gemm(n1, gemm_weight(att.q_proj.weight), f32(att.q_proj.bias) if att.q_proj.bias is not None else None)
            k = ops.gemm(n1, gemm_weight(att.k_proj.weight), f32(att.k_proj.bias) if att.k_proj.bias is not None else None)
            v = ops.gemm(n1, gemm_weight(att.v_proj.weight), f32(att.v_proj.bias) if att.v_proj.bias is not None else None)
            qh = ops.rope_half_scatter(q, Hq, D, B, S, cos, sin)
            kh = ops.rope_half_scatter(k, Hkv, D, B, S, cos, sin)
            vt[..., :S] = v.view(B, S, Hkv, D).permute(0, 2, 3, 1)            # V^T (columns >= S stay zero)
            a = ops.attention_causal_gqa(qh, kh, vt, S, D ** -0.5, valid)
            ob = f32(att.o_proj.bias) if att.o_proj.bias is not None else None
            h = ops.gemm(a.view(B * S, Hq * D), gemm_weight(att.o_proj.weight), ob, ops.EPI_BIAS_GATE_RES, res=h, gate=self._gate_ones(d, dev))
            n2 = ops.rmsnorm(h, f32(layer.post_attention_layernorm.weight), self.eps)
            g = ops.gemm(n2, gemm_weight(mlp.gate_proj.weight), None)
            u = ops.gemm(n2, gemm_weight(mlp.up_proj.weight), None)
            act = ops.silu_mul(g, u)
            h = ops.gemm(act, gemm_weight(mlp.down_proj.weight), None, ops.EPI_BIAS_GATE_RES, res=h, gate=self._gate_ones(d, dev))
            hidden_states.append(h.view(B, S, d))
        last = ops.rmsnorm(h, f32(text.norm.weight), self.eps).view(B, S, d)
        # transformers: hidden_states = (embeddings, layer 1 .. layer L-1 outputs, NORMED layer L output)
        hidden_states[-1] = last
        return types.SimpleNamespace(last_hidden_state=last, hidden_states=tuple(hidden_states) if output_hidden_states else None)


def use_hip_text_encoder(pipeline, index: int = 0):
    """Swap the transformers model in a pipeline's text_encoder slot for its device-library forward (weights shared)."""
    name = "text_encoder" if index == 0 else "text_encoder_2"
    enc = getattr(pipeline, name)
    if enc is not None and not isinstance(enc, Qwen2VLTextEncoderHIP):
        setattr(pipeline, name, Qwen2VLTextEncoderHIP(enc))
    return getattr(pipeline, name)
