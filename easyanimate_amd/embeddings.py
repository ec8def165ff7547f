"""Host-side table construction for the 3-D rotary embedding (once per pipeline call, on the host exactly
like the reference: pipeline_easyanimate.py:999-1011 and diffusers get_3d_rotary_pos_embed).
The tables are consumed on the device by ea_qknorm_rope_bf16."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """reference: easyanimate/pipeline/pipeline_easyanimate.py:82-97"""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        resize_height = th
        resize_width = int(round(th / h * w))
    else:
        resize_width = tw
        resize_height = int(round(tw / w * h))
    crop_top = int(round((th - resize_height) / 2.0))
    crop_left = int(round((tw - resize_width) / 2.0))
    return (crop_top, crop_left), (crop_top + resize_height, crop_left + resize_width)


def _rope_1d(dim: int, pos: np.ndarray, theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin of pos x theta^(-2i/dim), each value repeated twice (interleaved pairs); fp32 [len(pos), dim]."""
    pos_t = torch.from_numpy(np.asarray(pos, dtype=np.float32))
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos_t, freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def get_3d_rotary_pos_embed(embed_dim: int, crops_coords, grid_size, temporal_size: int, theta: int = 10000,
                            use_real: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(cos, sin) fp32 [temporal_size*gh*gw, embed_dim], channel split [t | h | w] = [d/4 | 3d/8 | 3d/8]."""
    if not use_real:
        raise ValueError("only use_real=True is supported")
    start, stop = crops_coords
    gh, gw = grid_size
    grid_h = np.linspace(start[0], stop[0], gh, endpoint=False, dtype=np.float32)
    grid_w = np.linspace(start[1], stop[1], gw, endpoint=False, dtype=np.float32)
    grid_t = np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32)
    dim_t, dim_h, dim_w = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3
    ct, st = _rope_1d(dim_t, grid_t, theta)
    ch, sh = _rope_1d(dim_h, grid_h, theta)
    cw, sw = _rope_1d(dim_w, grid_w, theta)

    def combine(ft, fh, fw):
        ft = ft[:, None, None, :].expand(-1, gh, gw, -1)
        fh = fh[None, :, None, :].expand(temporal_size, -1, gw, -1)
        fw = fw[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([ft, fh, fw], dim=-1).reshape(temporal_size * gh * gw, -1).contiguous()

    return combine(ct, ch, cw), combine(st, sh, sw)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size, interpolation_scale: float = 1.0, base_size: int = 16) -> torch.Tensor:
    """Fixed 2-D sin/cos table of the ref-latent branch (reference: transformer3d.py:1424 calls diffusers
    get_2d_sincos_pos_embed; SURVEY Appendix A "[diffusers, restated]").  Returns float64 [gh*gw, embed_dim]:
    first half of the channels encodes the coordinate that runs fastest in memory ("w goes first" in diffusers), each half
    as [sin | cos] over embed_dim/4 frequencies 10000^(-i / (embed_dim/4)); coordinates are scaled to a 16-cell base grid."""
    gh, gw = (grid_size, grid_size) if isinstance(grid_size, int) else grid_size
    ch = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    cw = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    mesh = np.stack(np.meshgrid(cw, ch), axis=0).reshape(2, -1)            # [2, gh*gw]: row 0 = w coordinate, row 1 = h
    quarter = embed_dim // 4
    freq = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / quarter)
    parts = []
    for coord in mesh:                                                     # float32 coordinates x float64 frequencies
        ang = coord.astype(np.float32)[:, None] * freq[None, :]
        parts += [np.sin(ang), np.cos(ang)]
    return torch.from_numpy(np.concatenate(parts, axis=1))
