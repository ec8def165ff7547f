"""TEST INFRASTRUCTURE ONLY -- oracle restatement of the MAGVIT causal 3-D VAE (AutoencoderKLMagvit, V5/V5.1
settings: spatial_group_norm=True, "spatial" mid-block attention).  Plain PyTorch on NCDHW tensors, pure functions
over a state dict with the reference's key names.

It is written in the *monolithic* causal form (padding_flag 0 of vaemodules/common.py:89-96: replicate-pad the first
frame, one convolution over the whole clip).  The reference's inference path is the *chunked* form (padding_flag
3/4 with per-conv frame caches, common.py:97-141, omnigen_enc_dec.py:283-291,621-629); the two coincide under the
V5 settings (SURVEY.md 8c property 1).  The golden vectors in tests/golden/vae_*.pt are produced by the unchanged
reference in its chunked mode, so tests/test_oracle_cpu.py pins this restatement -- and that equivalence -- to it.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def causal_conv3d(sd: SD, pre: str, x, stride=(1, 1, 1), padding=1):
    """CausalConv3d, common.py:84-96 (flag 0); strided down-samplers pre-pad right/bottom (downsamplers.py:44-46,92-94)."""
    w, b = sd[pre + "weight"], sd[pre + "bias"]
    kt = w.shape[2]
    x = F.pad(x, (0, 0, 0, 0, kt - 1, 0), mode="replicate")
    return F.conv3d(x, w, b, stride=stride, padding=(0, padding, padding))


def group_norm_per_frame(sd: SD, pre: str, x, groups: int, eps: float = 1e-6):
    """common.py:301-305: GroupNorm applied to (b t) c h w."""
    B, C, T, H, W = x.shape
    y = F.group_norm(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), groups, sd[pre + "weight"], sd[pre + "bias"], eps)
    return y.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)


def res_block(sd: SD, pre: str, x, groups: int):
    """ResidualBlock3D.forward, common.py:298-323"""
    sc = x
    if pre + "shortcut.weight" in sd:
        sc = F.conv3d(x, sd[pre + "shortcut.weight"], sd[pre + "shortcut.bias"])
    h = F.silu(group_norm_per_frame(sd, pre + "norm1.", x, groups))
    h = causal_conv3d(sd, pre + "conv1.", h)
    h = F.silu(group_norm_per_frame(sd, pre + "norm2.", h, groups))
    h = causal_conv3d(sd, pre + "conv2.", h)
    return h + sc


def spatial_attention(sd: SD, pre: str, x, groups: int):
    """SpatialAttention + AttnProcessor2_0 (vaemodules/attention.py:391-423, attention_processors.py:76-139): 1 head."""
    B, C, T, H, W = x.shape
    h = x.permute(0, 2, 3, 4, 1).reshape(B * T, H * W, C)
    res = h
    hn = F.group_norm(h.transpose(1, 2), groups, sd[pre + "group_norm.weight"], sd[pre + "group_norm.bias"], 1e-6).transpose(1, 2)
    q = F.linear(hn, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"])[:, None]
    k = F.linear(hn, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"])[:, None]
    v = F.linear(hn, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"])[:, None]
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False, scale=C ** -0.5)[:, 0]
    o = F.linear(o, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"]) + res
    return o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def mid_block(sd: SD, pre: str, x, groups: int):
    """MidBlock3D.forward, mid_blocks.py:183-196"""
    x = res_block(sd, pre + "convs.0.", x, groups)
    i = 0
    while pre + f"convs.{i + 1}.norm1.weight" in sd:
        if pre + f"attentions.{i}.to_q.weight" in sd:
            x = spatial_attention(sd, pre + f"attentions.{i}.", x, groups)
        x = res_block(sd, pre + f"convs.{i + 1}.", x, groups)
        i += 1
    return x


def _n_convs(sd, pre):
    n = 0
    while pre + f"convs.{n}.norm1.weight" in sd:
        n += 1
    return n


def encoder(sd: SD, x, groups: int, temporal_down=(False, True, True, False)):
    """Encoder.single_forward on the whole clip, omnigen_enc_dec.py:230-277"""
    x = causal_conv3d(sd, "encoder.conv_in.", x)
    i = 0
    while f"encoder.down_blocks.{i}.convs.0.norm1.weight" in sd:
        pre = f"encoder.down_blocks.{i}."
        for j in range(_n_convs(sd, pre)):
            x = res_block(sd, pre + f"convs.{j}.", x, groups)
        if pre + "downsampler.conv.weight" in sd:
            x = F.pad(x, (0, 1, 0, 1))
            x = causal_conv3d(sd, pre + "downsampler.conv.", x, stride=(2 if temporal_down[i] else 1, 2, 2), padding=0)
        i += 1
    x = mid_block(sd, "encoder.mid_block.", x, groups)
    x = F.silu(group_norm_per_frame(sd, "encoder.conv_norm_out.", x, groups))
    return causal_conv3d(sd, "encoder.conv_out.", x)


def decoder(sd: SD, z, groups: int, temporal_up=(False, True, True, False)):
    """Decoder.single_forward on the whole clip, omnigen_enc_dec.py:555-615; up-samplers upsamplers.py:35,143-152"""
    x = causal_conv3d(sd, "decoder.conv_in.", z)
    x = mid_block(sd, "decoder.mid_block.", x, groups)
    i = 0
    while f"decoder.up_blocks.{i}.convs.0.norm1.weight" in sd:
        pre = f"decoder.up_blocks.{i}."
        for j in range(_n_convs(sd, pre)):
            x = res_block(sd, pre + f"convs.{j}.", x, groups)
        if pre + "upsampler.conv.weight" in sd:
            x = F.interpolate(x, scale_factor=(1, 2, 2), mode="nearest")
            x = causal_conv3d(sd, pre + "upsampler.conv.", x)
            if temporal_up[i] and x.shape[2] > 1:
                first, rest = x[:, :, :1], x[:, :, 1:]
                rest = F.interpolate(rest, scale_factor=(2, 1, 1), mode="nearest")
                x = torch.cat([first, rest], dim=2)
        i += 1
    x = F.silu(group_norm_per_frame(sd, "decoder.conv_norm_out.", x, groups))
    return causal_conv3d(sd, "decoder.conv_out.", x)


def vae_encode_moments(sd: SD, x, groups: int):
    """AutoencoderKLMagvit.encode up to the moments, autoencoder_magvit.py:256-262"""
    h = encoder(sd, x, groups)
    return F.conv3d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def vae_decode(sd: SD, z, groups: int):
    """AutoencoderKLMagvit._decode, autoencoder_magvit.py:281-282"""
    z = F.conv3d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder(sd, z, groups)
