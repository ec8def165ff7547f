"""AutoencoderKLMagvit -- the V5/V5.1 causal 3-D VAE, with the constructor / config / encode / decode surface of
/root/reference/easyanimate/models/autoencoder_magvit.py:59-317,478-505 and the Encoder / Decoder trunks of
/root/reference/easyanimate/vae/ldm/models/omnigen_enc_dec.py:25-677 (same state-dict keys).  Arithmetic: HIP
kernels on channels-last activations (ea_conv3d_cl_bf16, ea_groupnorm_*, ea_gemm_bf16, ea_softmax_rows_*)."""
from __future__ import annotations

import glob
import json
import os
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
from torch import nn

from . import ops
from ._params import f32
from .config import ConfigMixin, register_to_config
from .vae_modules import CausalConv3d, _gn, conv_cl, get_down_block, get_mid_block, get_up_block


def str_eval(item):
    return eval(item) if isinstance(item, str) else item


class DiagonalGaussianDistribution:
    """diffusers DiagonalGaussianDistribution: mean/logvar chunk, clamp(-30, 20)."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        gdev = generator.device if generator is not None else self.parameters.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.parameters.dtype)
        return self.mean + self.std * noise.to(self.parameters.device)

    def mode(self) -> torch.Tensor:
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


@dataclass
class DecoderOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class Encoder(nn.Module):
    """reference: omnigen_enc_dec.py:25-337.  forward takes/returns channels-last clips [T,H,W,C]."""

    def __init__(self, in_channels=3, out_channels=8, down_block_types=("SpatialDownBlock3D",), ch=128, ch_mult=[1, 2, 4, 4],
                 block_out_channels=[128, 256, 512, 512], use_gc_blocks=None, mid_block_type="MidBlock3D",
                 mid_block_use_attention=True, mid_block_attention_type="3d", mid_block_num_attention_heads=1,
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", num_attention_heads=1, double_z=True,
                 slice_mag_vae=False, slice_compression_vae=False, cache_compression_vae=False, cache_mag_vae=False,
                 spatial_group_norm=False, mini_batch_encoder=9, verbose=False):
        super().__init__()
        if block_out_channels is None:
            block_out_channels = [ch * i for i in ch_mult]
        assert len(down_block_types) == len(block_out_channels)
        self.conv_in = CausalConv3d(in_channels, block_out_channels[0], kernel_size=3)
        self.down_blocks = nn.ModuleList([])
        output_channels = block_out_channels[0]
        for i, t in enumerate(down_block_types):
            input_channels, output_channels = output_channels, block_out_channels[i]
            self.down_blocks.append(get_down_block(
                t, in_channels=input_channels, out_channels=output_channels, num_layers=layers_per_block, act_fn=act_fn,
                norm_num_groups=norm_num_groups, norm_eps=1e-6, num_attention_heads=num_attention_heads,
                add_gc_block=bool(use_gc_blocks[i]) if use_gc_blocks else False,
                add_downsample=i != len(block_out_channels) - 1))
        self.mid_block = get_mid_block(mid_block_type, in_channels=block_out_channels[-1], num_layers=layers_per_block,
                                       act_fn=act_fn, norm_num_groups=norm_num_groups, norm_eps=1e-6,
                                       add_attention=mid_block_use_attention, attention_type=mid_block_attention_type,
                                       num_attention_heads=mid_block_num_attention_heads)
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[-1], num_groups=norm_num_groups, eps=1e-6)
        self.conv_out = CausalConv3d(block_out_channels[-1], 2 * out_channels if double_z else out_channels, kernel_size=3)
        self.cache_mag_vae, self.spatial_group_norm, self.mini_batch_encoder = cache_mag_vae, spatial_group_norm, mini_batch_encoder

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv_in(x)
        for blk in self.down_blocks:
            x = blk(x)
        x = self.mid_block(x)
        x = _gn(self.conv_norm_out, x, act=True)
        return self.conv_out(x)


class Decoder(nn.Module):
    """reference: omnigen_enc_dec.py:339-677."""

    def __init__(self, in_channels=8, out_channels=3, up_block_types=("SpatialUpBlock3D",), ch=128, ch_mult=[1, 2, 4, 4],
                 block_out_channels=[128, 256, 512, 512], use_gc_blocks=None, mid_block_type="MidBlock3D",
                 mid_block_use_attention=True, mid_block_attention_type="3d", mid_block_num_attention_heads=1,
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", num_attention_heads=1, slice_mag_vae=False,
                 slice_compression_vae=False, cache_compression_vae=False, cache_mag_vae=False, spatial_group_norm=False,
                 mini_batch_decoder=3, verbose=False):
        super().__init__()
        if block_out_channels is None:
            block_out_channels = [ch * i for i in ch_mult]
        assert len(up_block_types) == len(block_out_channels)
        self.conv_in = CausalConv3d(in_channels, block_out_channels[-1], kernel_size=3)
        self.mid_block = get_mid_block(mid_block_type, in_channels=block_out_channels[-1], num_layers=layers_per_block,
                                       act_fn=act_fn, norm_num_groups=norm_num_groups, norm_eps=1e-6,
                                       add_attention=mid_block_use_attention, attention_type=mid_block_attention_type,
                                       num_attention_heads=mid_block_num_attention_heads)
        self.up_blocks = nn.ModuleList([])
        rev = list(reversed(block_out_channels))
        output_channels = rev[0]
        for i, t in enumerate(up_block_types):
            input_channels, output_channels = output_channels, rev[i]
            self.up_blocks.append(get_up_block(
                t, in_channels=input_channels, out_channels=output_channels, num_layers=layers_per_block + 1, act_fn=act_fn,
                norm_num_groups=norm_num_groups, norm_eps=1e-6, num_attention_heads=num_attention_heads,
                add_gc_block=bool(use_gc_blocks[i]) if use_gc_blocks else False,
                add_upsample=i != len(block_out_channels) - 1))
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=1e-6)
        self.conv_out = CausalConv3d(block_out_channels[0], out_channels, kernel_size=3)
        self.cache_mag_vae, self.spatial_group_norm, self.mini_batch_decoder = cache_mag_vae, spatial_group_norm, mini_batch_decoder

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv_in(x)
        x = self.mid_block(x)
        # the virtual temporal x2 travels as a flag on the tensor (vae_modules.is_virtual): a flag an intervening op dropped would
        # silently turn 2T - 1 logical frames into T.  The logical frame count is therefore checked after every block against
        # what the block's up-sampler must produce.
        from . import vae_parallel
        from .vae_modules import SpatialTemporalUpsampler3D, is_virtual
        expect = x.shape[0]
        for blk in self.up_blocks:
            x = blk(x)
            if any(isinstance(mod, SpatialTemporalUpsampler3D) for mod in blk.modules()):
                expect = 2 * expect - 1
            logical = 2 * x.shape[0] - 1 if is_virtual(x) else x.shape[0]
            if vae_parallel.current() is None and logical != expect:     # (a temporal split owns a frame RANGE and never goes virtual)
                raise RuntimeError(f"VAE decoder: {logical} logical frames behind {type(blk).__name__}, expected {expect} -- the "
                                   f"virtual-duplication flag of a temporally up-sampled clip was lost on the way")
        if getattr(x, "tvirt", False):
            # a temporal up-sampler in the LAST up block (not a V5 / V5.1 layout): nobody left to address the duplicated
            # frames virtually -- materialise them
            idx = (torch.arange(2 * x.shape[0] - 1, device=x.device) + 1) >> 1
            x = x[idx].contiguous()
        x = _gn(self.conv_norm_out, x, act=True)
        return self.conv_out(x)


class AutoencoderKLMagvit(nn.Module, ConfigMixin):
    """reference: autoencoder_magvit.py:59-317.  encode(x[B,3,F,H,W]) -> posterior over [B,16,F',H/8,W/8];
    decode(z[B,16,F',h,w]) -> [B,3,F,8h,8w]."""
    config_name = "config.json"

    @register_to_config
    def __init__(self, in_channels: int = 3, out_channels: int = 3, ch=128, ch_mult=[1, 2, 4, 4],
                 block_out_channels=[128, 256, 512, 512], use_gc_blocks=None, down_block_types: tuple = None,
                 up_block_types: tuple = None, mid_block_type: str = "MidBlock3D", mid_block_use_attention: bool = True,
                 mid_block_attention_type: str = "3d", mid_block_num_attention_heads: int = 1, layers_per_block: int = 2,
                 act_fn: str = "silu", num_attention_heads: int = 1, latent_channels: int = 4, norm_num_groups: int = 32,
                 scaling_factor: float = 0.1825, force_upcast: float = True, slice_mag_vae=True,
                 slice_compression_vae=False, cache_compression_vae=False, cache_mag_vae=False, use_tiling=False,
                 use_tiling_encoder=False, use_tiling_decoder=False, mini_batch_encoder=9, mini_batch_decoder=3,
                 upcast_vae=False, spatial_group_norm=False, tile_sample_min_size=384, tile_overlap_factor=0.25):
        super().__init__()
        down_block_types = str_eval(down_block_types)
        up_block_types = str_eval(up_block_types)
        if not spatial_group_norm:
            raise NotImplementedError(
                "AutoencoderKLMagvit (MI355X) implements the V5/V5.1 setting spatial_group_norm=True (per-frame "
                "GroupNorm, nearest temporal up-sampling); only then is whole-clip evaluation equal to the reference's "
                "chunked evaluation")
        if not cache_mag_vae or mini_batch_decoder != 1 or mini_batch_encoder % 2 != 0:
            # whole-clip causal evaluation equals the reference's chunked evaluation only for its cached mode with an even
            # encoder chunk and one latent frame per decoder chunk (SURVEY 8c property 1; the V5 / V5.1 YAML values are
            # cache_mag_vae: true, mini_batch_encoder 4, mini_batch_decoder 1): other settings chunk differently there
            # (stride-2 misalignment, per-chunk attention) and would silently differ
            raise NotImplementedError(
                f"AutoencoderKLMagvit (MI355X): cache_mag_vae={cache_mag_vae}, mini_batch_encoder={mini_batch_encoder}, "
                f"mini_batch_decoder={mini_batch_decoder} -- only cache_mag_vae=True with an even mini_batch_encoder and "
                "mini_batch_decoder=1 (the V5 / V5.1 configuration) is evaluated identically to the reference")
        if upcast_vae:
            # the reference's upcast_vae moves the encoder / decoder to fp32 inside encode / _decode (autoencoder_magvit.py:244-248,
            # 272-275); this library computes in bf16 with fp32 accumulation and has no fp32 convolution path
            raise NotImplementedError("AutoencoderKLMagvit (MI355X): upcast_vae=True asks for an fp32 VAE; the HIP library is bf16-only "
                                      "(fp32 accumulation, fp32 GroupNorm statistics). Load the checkpoint with upcast_vae=False")
        common = dict(ch=ch, ch_mult=ch_mult, block_out_channels=block_out_channels, use_gc_blocks=use_gc_blocks,
                      mid_block_type=mid_block_type, mid_block_use_attention=mid_block_use_attention,
                      mid_block_attention_type=mid_block_attention_type,
                      mid_block_num_attention_heads=mid_block_num_attention_heads, layers_per_block=layers_per_block,
                      norm_num_groups=norm_num_groups, act_fn=act_fn, num_attention_heads=num_attention_heads,
                      slice_mag_vae=slice_mag_vae, slice_compression_vae=slice_compression_vae,
                      cache_compression_vae=cache_compression_vae, cache_mag_vae=cache_mag_vae,
                      spatial_group_norm=spatial_group_norm)
        self.encoder = Encoder(in_channels=in_channels, out_channels=latent_channels, down_block_types=down_block_types,
                               double_z=True, mini_batch_encoder=mini_batch_encoder, **common)
        self.decoder = Decoder(in_channels=latent_channels, out_channels=out_channels, up_block_types=up_block_types,
                               mini_batch_decoder=mini_batch_decoder, **common)
        self.quant_conv = nn.Conv3d(2 * latent_channels, 2 * latent_channels, kernel_size=1)
        self.post_quant_conv = nn.Conv3d(latent_channels, latent_channels, kernel_size=1)
        self.slice_mag_vae, self.slice_compression_vae = slice_mag_vae, slice_compression_vae
        self.cache_compression_vae, self.cache_mag_vae = cache_compression_vae, cache_mag_vae
        self.cache_compression_vae_copy, self.cache_mag_vae_copy = cache_compression_vae, cache_mag_vae
        self.mini_batch_encoder, self.mini_batch_decoder = mini_batch_encoder, mini_batch_decoder
        # use_slicing (one batch element at a time, autoencoder_magvit.py:256-258,299-301) is how this class always works
        self.use_slicing = False
        # spatial tiling (autoencoder_magvit.py:249-254,276-279,339-448): host loops over the whole-tile kernels + ea_tile_blend.
        # Never needed for memory here (288 GB) and it CHANGES results (blended overlaps) -- kept because a checkpoint's config.json
        # may ask for it and the reference then returns the tiled result
        self.use_tiling, self.use_tiling_encoder, self.use_tiling_decoder = use_tiling, use_tiling_encoder, use_tiling_decoder
        self.upcast_vae = upcast_vae
        self.tile_sample_min_size, self.tile_overlap_factor = tile_sample_min_size, tile_overlap_factor
        self.tile_latent_min_size = int(tile_sample_min_size / (2 ** (len(ch_mult) - 1)))
        self.scaling_factor = scaling_factor
        self.latent_channels = latent_channels
        self.out_channels = out_channels
        self.temporal_parallel = None   # set by enable_temporal_parallel()

    # ------------------------------------------------------------------------------------------
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def _clear_conv_cache(self):
        pass  # whole-clip evaluation keeps no chunk caches

    def enable_temporal_parallel(self, group=None, spatial: int = 1):
        """Exact multi-GPU encode / decode: every rank of `group` holds the same weights and passes the same input; each
        evaluates a contiguous range of frames -- and, with spatial = P_s > 1, H / P_s rows of them: 8 ranks on 13 latent
        frames run as 4 (time) x 2 (rows) with every rank busy -- and all ranks return the whole result
        (easyanimate_amd/vae_parallel.py)."""
        from .vae_parallel import TemporalParallel
        self.temporal_parallel = TemporalParallel(group, spatial=spatial)
        return self.temporal_parallel

    def enable_temporal_self_loop(self, group=None, virtual: int = 2):
        """One-GPU bring-up of the split decode: see vae_parallel.SelfLoopTemporal."""
        from .vae_parallel import SelfLoopTemporal
        self.temporal_parallel = SelfLoopTemporal(group, virtual=virtual)
        return self.temporal_parallel

    def disable_temporal_parallel(self):
        self.temporal_parallel = None

    def enable_cache_in_vae(self):
        self.cache_compression_vae, self.cache_mag_vae = self.cache_compression_vae_copy, self.cache_mag_vae_copy

    def disable_cache_in_vae(self):
        self.cache_compression_vae, self.cache_mag_vae = False, False

    def _check(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKLMagvit (MI355X): inputs must be on the GPU; there is no CPU fallback")
        if x.dim() != 5:
            raise ValueError("expected [B, C, F, H, W]")

    @property
    def _enc_c_pad(self):
        """Channel padding of the encoder's input layout: <= 8 input channels (RGB) travel as one 16-byte chunk per voxel, the
        form the 8-channel implicit-GEMM kernel gathers (vae_modules._conv_cl_local)."""
        ci = self.encoder.conv_in.in_channels
        return 8 if ci <= 8 else None

    def _encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """quant_conv(encoder(x)) of a whole clip (or one spatial tile of it): [B,3,F,H,W] -> moments [B,2C,F',H/8,W/8] in x's dtype."""
        if x.shape[2] != 1 and (x.shape[2] - 1) % self.mini_batch_encoder != 0:
            raise ValueError(f"encode: {x.shape[2]} frames -- the reference's chunked encoder only matches whole-clip evaluation for "
                             f"1 + k * mini_batch_encoder ({self.mini_batch_encoder}) frames (predict_t2v.py:288-291 trims the video to that)")
        in_dtype = x.dtype
        odt = in_dtype if in_dtype in (torch.float32, torch.bfloat16) else torch.float32
        tp = self.temporal_parallel
        n_temporal = sum(1 for blk in self.encoder.down_blocks if getattr(blk, "temporal_downsample_factor", 1) == 2)
        moments = []
        for b in range(x.shape[0]):
            xb = x[b]
            if xb.dtype not in (torch.float32, torch.bfloat16):
                xb = xb.float()
            if tp is None:
                h = self.encoder(ops.ncdhw_to_ndhwc(xb.contiguous(), self._enc_c_pad))
                m = conv_cl(self.quant_conv, h)  # 1x1x1
                moments.append(ops.ndhwc_to_ncdhw(m, self.quant_conv.out_channels, odt))
                continue
            from . import vae_parallel
            t_lat = (xb.shape[1] - 1) // (2 ** n_temporal) + 1
            ranges = tp.plan(t_lat)
            fr = ranges[tp.rank]
            for _ in range(n_temporal):
                fr = tp.finer(fr)
            m_loc = None
            tp.rows(xb.shape[2] // 8)                  # the latent rows must divide over the spatial ranks
            r0, r1 = tp.rows(xb.shape[2])
            if tp.is_active:
                with vae_parallel.activate(tp):
                    h = self.encoder(ops.ncdhw_to_ndhwc(xb[:, fr[0]:fr[1], r0:r1].contiguous(), self._enc_c_pad))
                    m_loc = ops.ndhwc_to_ncdhw(conv_cl(self.quant_conv, h), self.quant_conv.out_channels, odt)
            like = torch.empty((self.quant_conv.out_channels, 1, xb.shape[2] // 8, xb.shape[3] // 8), dtype=odt, device=xb.device)
            moments.append(tp.gather_frames(m_loc, ranges, 1, like, row_dim=2))
        return torch.stack(moments).to(in_dtype)

    def _blend_and_stitch(self, rows, blend_extent: int, row_limit: int) -> torch.Tensor:
        """The second half of tiled_encode / tiled_decode (autoencoder_magvit.py:362-375, 408-421): every tile is blended IN PLACE
        with the (already blended) tile above it and the one left of it -- the reference's order, its in-place semantics -- then
        cropped to row_limit and concatenated."""
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    ops.tile_blend_(rows[i - 1][j], tile, blend_extent, 3)
                if j > 0:
                    ops.tile_blend_(row[j - 1], tile, blend_extent, 4)
                result_row.append(tile[:, :, :, :row_limit, :row_limit])
            result_rows.append(torch.cat(result_row, dim=4))
        return torch.cat(result_rows, dim=3)

    def tiled_encode(self, x: torch.Tensor, return_dict: bool = True):
        """reference: autoencoder_magvit.py:339-381.  Overlapping tile_sample_min_size^2 tiles, each encoded as a clip of its own
        by the whole-tile kernels, seams blended over tile_latent_min_size * tile_overlap_factor latents."""
        if self.temporal_parallel is not None:
            raise NotImplementedError("spatial tiling together with the multi-GPU VAE split is not supported")
        overlap_size = int(self.tile_sample_min_size * (1 - self.tile_overlap_factor))
        blend_extent = int(self.tile_latent_min_size * self.tile_overlap_factor)
        row_limit = self.tile_latent_min_size - blend_extent
        t = self.tile_sample_min_size
        rows = [[self._encode_moments(x[:, :, :, i:i + t, j:j + t].contiguous()).contiguous() for j in range(0, x.shape[4], overlap_size)]
                for i in range(0, x.shape[3], overlap_size)]
        posterior = DiagonalGaussianDistribution(self._blend_and_stitch(rows, blend_extent, row_limit))
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)

    def encode(self, x: torch.Tensor, return_dict: bool = True) -> Union[AutoencoderKLOutput, Tuple[DiagonalGaussianDistribution]]:
        self._check(x)
        if (self.use_tiling or self.use_tiling_encoder) and (x.shape[-1] > self.tile_sample_min_size or x.shape[-2] > self.tile_sample_min_size):
            return self.tiled_encode(x, return_dict=return_dict)
        posterior = DiagonalGaussianDistribution(self._encode_moments(x))
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)

    def _decode_local(self, z: torch.Tensor, out_dtype, post: int = 0) -> torch.Tensor:
        zc = z.contiguous()
        if zc.dtype not in (torch.float32, torch.bfloat16):
            zc = zc.float()
        h = conv_cl(self.post_quant_conv, ops.ncdhw_to_ndhwc(zc))
        h = h[..., :self.post_quant_conv.out_channels].contiguous() if h.shape[-1] != self.post_quant_conv.out_channels else h
        y = self.decoder(h)
        return ops.ndhwc_to_ncdhw(y, self.out_channels, out_dtype, post)

    def _decode_one(self, z: torch.Tensor, out_dtype, post: int = 0) -> torch.Tensor:
        tp = self.temporal_parallel
        if tp is None:
            return self._decode_local(z, out_dtype, post)
        from . import vae_parallel
        ranges = tp.plan(z.shape[1])
        from .vae_modules import SpatialTemporalUpsampler3D
        n_temporal = sum(1 for blk in self.decoder.up_blocks if isinstance(getattr(blk, "upsampler", None), SpatialTemporalUpsampler3D))
        out_ranges = ranges
        for _ in range(n_temporal):
            out_ranges = [tp.finer(r) for r in out_ranges]
        y = None
        if isinstance(tp, vae_parallel.SelfLoopTemporal):
            # bring-up: this one rank plays the temporal ranks in turn, halos through the group's send / recv to itself
            parts = []
            for v in range(tp.active_ranks):
                tp.enter(v)
                a, b = ranges[v]
                with vae_parallel.activate(tp):
                    parts.append(self._decode_local(z[:, a:b], out_dtype, post))
            return torch.cat(parts, dim=1)
        r0, r1 = tp.rows(z.shape[2])
        if tp.is_active:
            a, b = ranges[tp.rank]
            with vae_parallel.activate(tp):
                y = self._decode_local(z[:, a:b, r0:r1], out_dtype, post)
        s_ = 2 ** (len(self.decoder.up_blocks) - 1)
        like = torch.empty((self.out_channels, 1, z.shape[2] * s_, z.shape[3] * s_), dtype=out_dtype, device=z.device)
        return tp.gather_frames(y, out_ranges, 1, like, row_dim=2)

    def _decode(self, z: torch.Tensor, postprocess: bool = False) -> torch.Tensor:
        odt = z.dtype if z.dtype in (torch.float32, torch.bfloat16) else torch.float32
        return torch.stack([self._decode_one(z[b], odt, int(postprocess)) for b in range(z.shape[0])]).to(z.dtype)

    def tiled_decode(self, z: torch.Tensor, return_dict: bool = True):
        """reference: autoencoder_magvit.py:383-448.  Overlapping tile_latent_min_size^2 latent tiles decoded as clips of their own,
        seams blended over tile_sample_min_size * tile_overlap_factor pixels, then the lower-right tile_latent_min_size^2 latents
        decoded once more and mixed into the corner."""
        if self.temporal_parallel is not None:
            raise NotImplementedError("spatial tiling together with the multi-GPU VAE split is not supported")
        overlap_size = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        blend_extent = int(self.tile_sample_min_size * self.tile_overlap_factor)
        row_limit = self.tile_sample_min_size - blend_extent
        t = self.tile_latent_min_size
        rows = [[self._decode(z[:, :, :, i:i + t, j:j + t].contiguous()).contiguous() for j in range(0, z.shape[4], overlap_size)]
                for i in range(0, z.shape[3], overlap_size)]
        dec = self._blend_and_stitch(rows, blend_extent, row_limit).contiguous()
        corner = self._decode(z[:, :, :, -t:, -t:].contiguous()).contiguous()
        ops.tile_corner_blend_(corner, dec)
        if not return_dict:
            return (dec,)
        return DecoderOutput(sample=dec)

    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None, postprocess: bool = False
               ) -> Union[DecoderOutput, Tuple[torch.Tensor]]:
        """postprocess=True additionally applies clamp(-1,1) -> /2+0.5 -> clamp(0,1) (pipeline decode_latents) in the
        final layout kernel."""
        self._check(z)
        if (self.use_tiling or self.use_tiling_decoder) and (z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size):
            dec = self.tiled_decode(z, return_dict=False)[0]
            if postprocess:     # (the fused form lives in the layout kernel of the untiled path; tiles must be blended first)
                dec = (dec.clamp(-1, 1) / 2 + 0.5).clamp(0, 1)
        else:
            dec = self._decode(z, postprocess)
        if not return_dict:
            return (dec,)
        return DecoderOutput(sample=dec)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True, generator=None):
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        dec = self.decode(z).sample
        return DecoderOutput(sample=dec) if return_dict else (dec,)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **vae_additional_kwargs):
        """reference: autoencoder_magvit.py:478-505"""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file) as f:
            config = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        vae_additional_kwargs.pop("vae_type", None)
        model = cls.from_config(config, **vae_additional_kwargs)
        from safetensors.torch import load_file
        files = sorted(glob.glob(os.path.join(pretrained_model_path, "*.safetensors")))
        state_dict = {}
        if files:
            for fp in files:
                state_dict.update(load_file(fp))
        else:
            bin_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
            if not os.path.isfile(bin_file):
                raise RuntimeError(f"no weights found under {pretrained_model_path}")
            state_dict = torch.load(bin_file, map_location="cpu", weights_only=True)
        own = model.state_dict()
        filtered = {k: v for k, v in state_dict.items() if k in own and own[k].shape == v.shape}
        m, u = model.load_state_dict(filtered, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        return model
