"""EasyAnimateTransformer3DModel (V5 / V5.1 MMDiT denoiser) -- same constructor, config attributes,
forward signature, return convention and state-dict keys as
/root/reference/easyanimate/models/transformer3d.py:1347-1689; arithmetic in libea_mi355x.so."""
from __future__ import annotations

import glob
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
from torch import nn

from . import ops
from ._params import bf16_weight, f32
from .attention import EasyAnimateDiTBlock
from .config import ConfigMixin, register_to_config
from .norm import EasyAnimateRMSNorm


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class Timesteps(nn.Module):
    """diffusers Timesteps: parameter-free sinusoid (kernel: ea_timestep_sinusoid)."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float, scale: int = 1):
        super().__init__()
        if not flip_sin_to_cos or downscale_freq_shift != 0 or scale != 1:
            raise NotImplementedError("only flip_sin_to_cos=True, freq_shift=0 (the EasyAnimate setting)")
        self.num_channels = num_channels


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding(in, time_embed_dim, 'silu'): keys linear_1.*, linear_2.*"""

    def __init__(self, in_channels: int, time_embed_dim: int, act_fn: str = "silu"):
        super().__init__()
        if act_fn != "silu":
            raise NotImplementedError(act_fn)
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class AdaLayerNorm(nn.Module):
    """diffusers AdaLayerNorm(embedding_dim, output_dim, chunk_dim=1): keys linear.*, norm.*"""

    def __init__(self, embedding_dim: int, output_dim: int, norm_elementwise_affine: bool, norm_eps: float,
                 chunk_dim: int = 1):
        super().__init__()
        assert chunk_dim == 1
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)

    def forward(self, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
        d = self.norm.normalized_shape[0]
        tab = ops.linear_small_m(temb.float().contiguous(), bf16_weight(self.linear.weight), f32(self.linear.bias), act_in=1)
        aff = self.norm.elementwise_affine
        # chunk_dim == 1: shift first, then scale
        return ops.layernorm_modulate(x, f32(self.norm.weight) if aff else None, f32(self.norm.bias) if aff else None,
                                      tab[:, d:2 * d], tab[:, 0:d], self.norm.eps)


class EasyAnimateTransformer3DModel(nn.Module, ConfigMixin):
    _supports_gradient_checkpointing = False
    config_name = "config.json"

    @register_to_config
    def __init__(
        self,
        num_attention_heads: int = 30,
        attention_head_dim: int = 64,
        in_channels: Optional[int] = None,
        out_channels: Optional[int] = None,
        patch_size: Optional[int] = None,
        sample_width: int = 90,
        sample_height: int = 60,
        ref_channels: int = None,
        clip_channels: int = None,
        activation_fn: str = "gelu-approximate",
        timestep_activation_fn: str = "silu",
        freq_shift: int = 0,
        num_layers: int = 30,
        mmdit_layers: int = 10000,
        swa_layers: list = None,
        dropout: float = 0.0,
        time_embed_dim: int = 512,
        add_norm_text_encoder: bool = False,
        text_embed_dim: int = 4096,
        text_embed_dim_t5: int = 4096,
        norm_eps: float = 1e-5,
        norm_elementwise_affine: bool = True,
        flip_sin_to_cos: bool = True,
        time_position_encoding_type: str = "3d_rope",
        after_norm=False,
        resize_inpaint_mask_directly: bool = False,
        enable_clip_in_inpaint: bool = True,
        position_of_clip_embedding: str = "full",
        enable_text_attention_mask: bool = True,
        add_noise_in_inpaint_model: bool = False,
        add_ref_latent_in_control_model: bool = False,
    ):
        super().__init__()
        if patch_size != 2:
            raise NotImplementedError("patch_size must be 2 (V5/V5.1)")
        if attention_head_dim != 64:
            raise NotImplementedError("attention_head_dim must be 64 (V5/V5.1)")
        self.num_heads = num_attention_heads
        self.inner_dim = num_attention_heads * attention_head_dim
        self.resize_inpaint_mask_directly = resize_inpaint_mask_directly
        self.patch_size = patch_size
        self.post_patch_height = sample_height // patch_size
        self.post_patch_width = sample_width // patch_size

        self.time_proj = Timesteps(self.inner_dim, flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(self.inner_dim, time_embed_dim, timestep_activation_fn)
        self.proj = nn.Conv2d(in_channels, self.inner_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=True)
        if not add_norm_text_encoder:
            self.text_proj = nn.Linear(text_embed_dim, self.inner_dim)
            if text_embed_dim_t5 is not None:
                self.text_proj_t5 = nn.Linear(text_embed_dim_t5, self.inner_dim)
        else:
            self.text_proj = nn.Sequential(EasyAnimateRMSNorm(text_embed_dim), nn.Linear(text_embed_dim, self.inner_dim))
            if text_embed_dim_t5 is not None:
                self.text_proj_t5 = nn.Sequential(EasyAnimateRMSNorm(text_embed_dim), nn.Linear(text_embed_dim_t5, self.inner_dim))
        if ref_channels is not None:
            # ref-latent branch (transformer3d.py:1420-1428): its own patch embedding + a fixed 2-D sin/cos table
            self.ref_proj = nn.Conv2d(ref_channels, self.inner_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=True)
            from .embeddings import get_2d_sincos_pos_embed
            self.register_buffer("ref_pos_embedding", get_2d_sincos_pos_embed(self.inner_dim, (self.post_patch_height, self.post_patch_width)),
                                 persistent=False)
        if clip_channels is not None:
            self.clip_proj = nn.Linear(clip_channels, self.inner_dim)
        self.swa_layers = swa_layers
        self.transformer_blocks = nn.ModuleList([
            EasyAnimateDiTBlock(
                dim=self.inner_dim, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                time_embed_dim=time_embed_dim, dropout=dropout, activation_fn=activation_fn,
                norm_elementwise_affine=norm_elementwise_affine, norm_eps=norm_eps, after_norm=after_norm,
                is_mmdit_block=True if index < mmdit_layers else False,
                is_swa=bool(swa_layers is not None and index in swa_layers))
            for index in range(num_layers)])
        self.norm_final = nn.LayerNorm(self.inner_dim, norm_eps, norm_elementwise_affine)
        self.norm_out = AdaLayerNorm(embedding_dim=time_embed_dim, output_dim=2 * self.inner_dim,
                                     norm_elementwise_affine=norm_elementwise_affine, norm_eps=norm_eps, chunk_dim=1)
        self.proj_out = nn.Linear(self.inner_dim, patch_size * patch_size * out_channels)
        self.teacache = None
        self.gradient_checkpointing = False
        self.sequence_parallel = None  # set by easyanimate_amd.sequence_parallel.enable()

    # ------------------------------------------------------------------------------------------
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_teacache(self, num_steps: int, rel_l1_thresh: float,
                        coefficients=[-10.47857366, 8.33844143, -0.78477557, 0.68798618, 0.0136149]):
        """reference: transformer3d.py:1485-1491."""
        from .teacache import TeaCache
        self.teacache = TeaCache(coefficients, num_steps, rel_l1_thresh=rel_l1_thresh)

    def _text_proj(self, mod, x: torch.Tensor) -> torch.Tensor:
        x = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        x = x.contiguous()
        if isinstance(mod, nn.Sequential):
            x = mod[0](x)
            lin = mod[1]
        else:
            lin = mod
        K = lin.weight.shape[1]
        kp = ops.round_up(K, 64)
        if kp != K:
            x = torch.nn.functional.pad(x, (0, kp - K))
        return ops.gemm(x, bf16_weight(lin.weight, kp), f32(lin.bias), ops.EPI_BIAS)

    def _ref_tokens(self, ref_latents: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
        """ref_proj (Conv2d k=s=2 per frame == gather + GEMM) + the position table, trilinearly resized from the
        (post_patch_height, post_patch_width) grid it was built on to this call's (gh, gw) grid (transformer3d.py:1538-1556).
        The resize touches one [gh*gw, d] table per forward: host-side table construction like the RoPE tables."""
        rb, rc, rf, rh, rw = ref_latents.shape
        lat = ref_latents if ref_latents.dtype in (torch.bfloat16, torch.float32) else ref_latents.float()
        k_pad = ops.round_up(rc * 4, 64)
        tok = ops.gemm(ops.patchify(lat.contiguous(), None, k_pad), bf16_weight(self.ref_proj.weight, k_pad), f32(self.ref_proj.bias),
                       ops.EPI_BIAS)                                        # [B, rf * rh/2 * rw/2, d]
        d = self.inner_dim
        pe = self.ref_pos_embedding.view(1, 1, self.post_patch_height, self.post_patch_width, d).permute(0, 4, 1, 2, 3)
        pe = torch.nn.functional.interpolate(pe, size=[1, gh, gw], mode="trilinear", align_corners=False)
        pe = pe.permute(0, 2, 3, 4, 1).reshape(1, -1, d).to(torch.bfloat16)
        if pe.shape[1] != tok.shape[1]:
            raise ValueError(f"ref_latents give {tok.shape[1]} tokens, the position table {pe.shape[1]} (one frame of {gh}x{gw} expected)")
        return ops.bf16_add_(tok, pe.expand(rb, -1, -1).contiguous())

    def time_embed(self, timestep: torch.Tensor, batch_size: int, bf16_round: bool = True) -> torch.Tensor:
        """transformer3d.py:1519-1520: sinusoid (fp32, rounded to the latent dtype) -> Linear -> SiLU -> Linear."""
        t = timestep.reshape(-1).to(device=self.device, dtype=torch.float32)
        if t.numel() == 1 and batch_size > 1:
            t = t.expand(batch_size)
        sin = ops.timestep_sinusoid(t.contiguous(), self.inner_dim, round_bf16=bf16_round)
        l1, l2 = self.time_embedding.linear_1, self.time_embedding.linear_2
        h = ops.linear_small_m(sin, bf16_weight(l1.weight), f32(l1.bias), act_in=0, act_out=1)
        return ops.linear_small_m(h, bf16_weight(l2.weight), f32(l2.bias))

    def forward(
        self,
        hidden_states,
        timestep,
        timestep_cond=None,
        encoder_hidden_states: Optional[torch.Tensor] = None,
        text_embedding_mask: Optional[torch.Tensor] = None,
        encoder_hidden_states_t5: Optional[torch.Tensor] = None,
        text_embedding_mask_t5: Optional[torch.Tensor] = None,
        image_meta_size=None,
        style=None,
        image_rotary_emb: Optional[torch.Tensor] = None,
        inpaint_latents: Optional[torch.Tensor] = None,
        control_latents: Optional[torch.Tensor] = None,
        ref_latents: Optional[torch.Tensor] = None,
        clip_encoder_hidden_states: Optional[torch.Tensor] = None,
        clip_attention_mask: Optional[torch.Tensor] = None,
        added_cond_kwargs: Dict[str, torch.Tensor] = None,
        return_dict=True,
    ):
        if clip_encoder_hidden_states is not None and ref_latents is None:
            raise ValueError("clip_encoder_hidden_states needs ref_latents (transformer3d.py:1559 concatenates the two)")
        if not hidden_states.is_cuda:
            raise RuntimeError("EasyAnimateTransformer3DModel (MI355X): inputs must be on the GPU; there is no CPU fallback")
        batch_size, channels, video_length, height, width = hidden_states.size()
        in_dtype = hidden_states.dtype
        p = self.patch_size
        sp = self.sequence_parallel
        if sp is not None:
            # multi-GPU: this rank's batch slice (CFG axis) -- the token shard (sequence axis) is cut after patchify
            b0, b1 = sp.begin(batch_size)
            if (b0, b1) != (0, batch_size):
                cut = lambda x: None if x is None else x[b0:b1]
                hidden_states, encoder_hidden_states = cut(hidden_states), cut(encoder_hidden_states)
                timestep = timestep if timestep.numel() == 1 else timestep.reshape(-1)[b0:b1]
                encoder_hidden_states_t5, inpaint_latents, control_latents = cut(encoder_hidden_states_t5), cut(inpaint_latents), cut(control_latents)
                ref_latents, clip_encoder_hidden_states = cut(ref_latents), cut(clip_encoder_hidden_states)
                batch_size = b1 - b0

        # 1. time embedding
        temb = self.time_embed(timestep, batch_size, bf16_round=in_dtype != torch.float32)

        # 2. patch embedding: channel concat + Conv2d(k=s=2) == gather + GEMM (transformer3d.py:1523-1531)
        extra = None
        if inpaint_latents is not None and control_latents is not None:
            extra = torch.cat([inpaint_latents, control_latents], 1)
        elif inpaint_latents is not None:
            extra = inpaint_latents
        elif control_latents is not None:
            extra = control_latents
        lat = hidden_states if hidden_states.dtype in (torch.bfloat16, torch.float32) else hidden_states.float()
        lat = lat.contiguous()
        if extra is not None:
            extra = extra.to(lat.dtype).contiguous()
        cin = channels + (extra.shape[1] if extra is not None else 0)
        if cin != self.proj.weight.shape[1]:
            raise ValueError(f"patch embedding expects {self.proj.weight.shape[1]} channels, got {cin}")
        k_pad = ops.round_up(cin * p * p, 64)
        cols = ops.patchify(lat, extra, k_pad)
        if sp is not None:
            cols = sp.shard_tokens(cols)
        hs = ops.gemm(cols, bf16_weight(self.proj.weight, k_pad), f32(self.proj.bias), ops.EPI_BIAS)

        # 3. text projection (transformer3d.py:1533-1536)
        enc = self._text_proj(self.text_proj, encoder_hidden_states.to(hs.device))
        if encoder_hidden_states_t5 is not None:
            enc_t5 = self._text_proj(self.text_proj_t5, encoder_hidden_states_t5.to(hs.device))
            enc = torch.cat([enc, enc_t5], dim=1).contiguous()

        # 3b. ref-latent / CLIP conditioning (transformer3d.py:1538-1561): the projected reference latents (+ a fixed 2-D
        # sin/cos table resized to the latent grid) REPLACE the text stream; CLIP tokens are prepended to them
        if ref_latents is not None:
            enc = self._ref_tokens(ref_latents.to(hs.device), height // p, width // p)
            if clip_encoder_hidden_states is not None:
                clip = self._text_proj(self.clip_proj, clip_encoder_hidden_states.to(hs.device))
                enc = torch.cat([clip, enc], dim=1).contiguous()

        rope = image_rotary_emb
        if rope is not None and sp is not None:
            rope = sp.shard_rope(rope, hs.device)
        elif rope is not None:
            from .processor import rope_to_device
            rope = rope_to_device(rope, hs.device)   # once per forward (the pipeline passes device tables already)

        # TeaCache (transformer3d.py:1564-1590): skip the blocks while the accumulated, rescaled rel-L1 change of the
        # first block's modulated input stays under the threshold; a skipped step re-applies the cached residual
        should_calc = True
        if self.teacache is not None:
            if hs.dtype != torch.bfloat16:
                raise NotImplementedError("TeaCache: the device path keeps bf16 state (the model dtype of every V5.1 config)")
            modulated_inp = self.transformer_blocks[0].norm1(hs, enc, temb)[0]
            residual = self.teacache.previous_residual
            should_calc = self.teacache.should_calc(modulated_inp, sp)
            if not should_calc:
                hs = ops.bf16_add_(hs, residual)

        if should_calc:
            ori_hs = hs
            # 4. transformer blocks
            for block in self.transformer_blocks:
                hs, enc = block(hidden_states=hs, encoder_hidden_states=enc, temb=temb, image_rotary_emb=rope,
                                num_frames=video_length, height=height // p, width=width // p, sp=sp)

            # 5. final norms (transformer3d.py:1673-1680); LayerNorm is per-row: only the video rows are normalised
            aff = self.norm_final.elementwise_affine
            hs = ops.layernorm_modulate(hs, f32(self.norm_final.weight) if aff else None,
                                        f32(self.norm_final.bias) if aff else None, None, None, self.norm_final.eps)
            hs = self.norm_out(hs, temb=temb)
            if self.teacache is not None:
                self.teacache.previous_residual = ops.bf16_sub(hs, ori_hs)   # :1635, kept in HBM
        hs = ops.gemm(hs, bf16_weight(self.proj_out.weight), f32(self.proj_out.bias), ops.EPI_BIAS)
        if sp is not None:
            hs = sp.gather_tokens(hs)

        # 6. un-patchify (transformer3d.py:1683-1685)
        out_dtype = in_dtype if in_dtype in (torch.bfloat16, torch.float32) else torch.float32
        output = ops.unpatchify(hs, channels, video_length, height // p, width // p, out_dtype)
        if output.dtype != in_dtype:
            output = output.to(in_dtype)
        if not return_dict:
            return (output,)
        return Transformer2DModelOutput(sample=output)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, transformer_additional_kwargs={},
                           low_cpu_mem_usage=False, torch_dtype=torch.bfloat16):
        """reference: transformer3d.py:1692-1809 (HF directory layout, strict=False, shape-mismatch skip,
        `proj.weight` input-channel padding)."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file) as f:
            config = json.load(f)
        config = {k: v for k, v in config.items() if not k.startswith("_")}
        model = cls.from_config(config, **{k: v for k, v in dict(transformer_additional_kwargs).items()
                                           if k != "transformer_type"})
        from safetensors.torch import load_file
        state_dict = {}
        files = sorted(glob.glob(os.path.join(pretrained_model_path, "*.safetensors")))
        if files:
            for fpath in files:
                state_dict.update(load_file(fpath))
        else:
            bin_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
            if not os.path.isfile(bin_file):
                raise RuntimeError(f"no weights found under {pretrained_model_path}")
            state_dict = torch.load(bin_file, map_location="cpu", weights_only=True)
        own = model.state_dict()
        if "proj.weight" in state_dict and own["proj.weight"].shape != state_dict["proj.weight"].shape:
            w_new, w_old = own["proj.weight"].clone(), state_dict["proj.weight"]
            if w_new.shape[1] > w_old.shape[1]:
                w_new[:, :w_old.shape[1]] = w_old
                w_new[:, w_old.shape[1]:] = 0
            else:
                w_new = w_old[:, :w_new.shape[1]]
            state_dict["proj.weight"] = w_new
        filtered = {k: v for k, v in state_dict.items() if k in own and own[k].shape == v.shape}
        missing, unexpected = model.load_state_dict(filtered, strict=False)
        print(f"### missing keys: {len(missing)}; \n### unexpected keys: {len(unexpected)};")
        return model.to(torch_dtype)
