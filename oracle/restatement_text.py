"""TEST INFRASTRUCTURE ONLY.  CPU restatement (plain PyTorch, fp32) of the text-only forward of transformers' Qwen2-VL decoder -- the
model the reference loads into its text_encoder slot (predict_t2v.py:205-214) and calls at pipeline_easyanimate.py:438-447.  The
algorithm lives in the third-party dependency `transformers` (not vendored in /root/reference); it is pinned here by
tests/test_oracle_cpu.py::test_text_restatement_equals_transformers against the INSTALLED transformers implementation, and serves
as the checker of easyanimate_amd/text_encoder.py's step order (same steps, device kernels) in tests/test_text_encoder_gpu.py."""
import torch
import torch.nn.functional as F


def _rms(x, w, eps):
    v = x.float()
    return w * (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)


def text_hidden_states(sd, cfg, input_ids, attention_mask, padded_positions="arange"):
    """sd: state dict of the decoder stack (keys embed_tokens.weight, layers.{i}.*, norm.weight); cfg: dict with hidden_size,
    num_attention_heads, num_key_value_heads, rms_norm_eps, rope_theta, num_hidden_layers.  Returns the tuple transformers returns as
    `.hidden_states`: (embeddings, layer 1 .. L-1 outputs, normed layer L output)."""
    d, Hq, Hkv, eps, theta, L = (cfg[k] for k in ("hidden_size", "num_attention_heads", "num_key_value_heads", "rms_norm_eps", "rope_theta",
                                                  "num_hidden_layers"))
    D = d // Hq
    x = sd["embed_tokens.weight"][input_ids]
    B, S, _ = x.shape
    am = attention_mask.long()
    # positions: transformers 5.x (installed here) lets the text model infer arange(S) for prompts without image / video tokens,
    # padding included; transformers 4.46 - 4.5x computed them in Qwen2VLForConditionalGeneration.get_rope_index (text-only
    # branch: cumsum(mask) - 1, padded slots -> 1).  Real tokens get 0 .. n-1 either way; only the padded rows differ.
    pos = torch.arange(S)[None].expand(B, S) if padded_positions == "arange" else (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
    inv_freq = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    ang = pos.float()[..., None] * inv_freq
    ang = torch.cat([ang, ang], -1)                                        # the three M-RoPE sections carry identical ids
    cos, sin = ang.cos()[:, None], ang.sin()[:, None]                      # [B, 1, S, D]
    keys = torch.arange(S)
    bad = (keys[None, None, None, :] >= am.sum(-1)[:, None, None, None]) | (keys[None, None, None, :] > keys[None, None, :, None])
    rot = lambda t: torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1)
    hs = [x]
    h = x
    for i in range(L):
        p = f"layers.{i}."
        n1 = _rms(h, sd[p + "input_layernorm.weight"], eps)
        q = F.linear(n1, sd[p + "self_attn.q_proj.weight"], sd.get(p + "self_attn.q_proj.bias")).view(B, S, Hq, D).transpose(1, 2)
        k = F.linear(n1, sd[p + "self_attn.k_proj.weight"], sd.get(p + "self_attn.k_proj.bias")).view(B, S, Hkv, D).transpose(1, 2)
        v = F.linear(n1, sd[p + "self_attn.v_proj.weight"], sd.get(p + "self_attn.v_proj.bias")).view(B, S, Hkv, D).transpose(1, 2)
        q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
        k, v = k.repeat_interleave(Hq // Hkv, 1), v.repeat_interleave(Hq // Hkv, 1)
        sc = (q @ k.transpose(2, 3)) * D ** -0.5
        a = (sc.masked_fill(bad, torch.finfo(sc.dtype).min).softmax(-1) @ v).transpose(1, 2).reshape(B, S, Hq * D)
        h = h + F.linear(a, sd[p + "self_attn.o_proj.weight"], sd.get(p + "self_attn.o_proj.bias"))
        n2 = _rms(h, sd[p + "post_attention_layernorm.weight"], eps)
        h = h + F.linear(F.silu(F.linear(n2, sd[p + "mlp.gate_proj.weight"])) * F.linear(n2, sd[p + "mlp.up_proj.weight"]), sd[p + "mlp.down_proj.weight"])
        hs.append(h)
    hs[-1] = _rms(h, sd["norm.weight"], eps)
    return tuple(hs)
