"""Leaf modules of the MAGVIT causal 3-D VAE -- same class names, constructor arguments and parameter names as
/root/reference/easyanimate/vae/ldm/modules/vaemodules/{common,down_blocks,up_blocks,mid_blocks,downsamplers,
upsamplers,attention}.py (the classes the V5/V5.1 config instantiates).

Internal activation layout is channels-last, one sample: bf16 [T, H, W, C].  The whole clip is processed at once
with causal (replicated-first-frame) temporal addressing inside the convolution kernel; under the V5 settings
(per-frame GroupNorm, nearest temporal up-sampling) this equals the reference's chunked/cached evaluation
(padding_flag 3/4, common.py:97-141) -- SURVEY.md 8c property 1 -- and with 288 GB of HBM no chunking is needed.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
from torch import nn

from . import ops
from ._params import derived, f32


def _pack_conv_weight(w: torch.Tensor, k_pad: Optional[int] = None, n_pad: Optional[int] = None) -> torch.Tensor:
    """[Cout, Cin, kt, kh, kw] -> bf16 [Cout(_pad), kt*kh*kw*Cin (_pad)], tap-major / channel-minor."""
    co = w.shape[0]
    p = w.permute(0, 2, 3, 4, 1).reshape(co, -1).to(torch.bfloat16)
    if k_pad is not None and k_pad != p.shape[1]:
        p = torch.nn.functional.pad(p, (0, k_pad - p.shape[1]))
    if n_pad is not None and n_pad != co:
        p = torch.nn.functional.pad(p, (0, 0, 0, n_pad - co))
    return p.contiguous()


def _pack_subpixel_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3, 3] -> bf16 [4, Cout, 12 * Cin]: the four parity classes (a, b) of "nearest x2, then 3x3x3" as 3 x 2 x 2
    convolutions on the source grid (ea_conv3d_cl_subpixel_bf16).  Output pixel (2i + a, 2j + b) reads up-sampled rows
    2i + a + {-1, 0, 1} = source rows {i-1, i, i} for a = 0 and {i, i, i+1} for a = 1: row tap 0 / 1 of class a = 0 is the sum of
    kh {0} / {1, 2}, of a = 1 of kh {0, 1} / {2}; columns likewise.  Summed in fp32, rounded to bf16 once; taps ordered
    (dt, row tap, column tap), channel-minor."""
    groups = (((0,), (1, 2)), ((0, 1), (2,)))          # [a][tap] -> original kernel indices
    wf = w.float()
    out = []
    for a in range(2):
        for b in range(2):
            taps = []
            for r in range(2):
                wr = sum(wf[:, :, :, kh] for kh in groups[a][r])            # [Co, Ci, 3, 3(kw)]
                for c in range(2):
                    taps.append(sum(wr[:, :, :, kw] for kw in groups[b][c]))   # [Co, Ci, 3(dt)]
            m = torch.stack(taps, -1).view(w.shape[0], w.shape[1], 3, 2, 2)      # [Co, Ci, dt, r, c]
            out.append(m.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1))
    return torch.stack(out).to(torch.bfloat16).contiguous()


def _pack_tmerge_weight(w: torch.Tensor, n_pad: int) -> torch.Tensor:
    """[Cout, Cin, 3, 3, 3] -> bf16 [2, Cout(_pad), 18 * Cin]: the temporal taps of a layer that reads a VIRTUALLY duplicated clip
    (logical frame f = physical (f + 1) >> 1) merged onto the two physical frames it touches (ea_conv3d_cl_bf16, tdup bit 8):
    even output frames see physical (p-1, p, p) -> {W_dt0, W_dt1 + W_dt2}, odd ones (p-1, p-1, p) -> {W_dt0 + W_dt1, W_dt2}; frame
    0 (all three taps on physical frame 0) comes out of the even class with the clamped p - 1.  Summed in fp32, rounded once."""
    wf = w.float()
    even = torch.stack([wf[:, :, 0], wf[:, :, 1] + wf[:, :, 2]], 2)      # [Co, Ci, 2, 3, 3]
    odd = torch.stack([wf[:, :, 0] + wf[:, :, 1], wf[:, :, 2]], 2)
    out = torch.stack([m.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1) for m in (even, odd)])
    if n_pad != w.shape[0]:
        out = torch.nn.functional.pad(out, (0, 0, 0, n_pad - w.shape[0]))
    return out.to(torch.bfloat16).contiguous()


TEMPORAL_TAP_MERGE = True  # False: a layer behind a virtual temporal x2 runs its 27 taps with the frame map in the addressing (A/B, tests)
SUBPIXEL_UPSAMPLE = True   # False: the up-samplers run as 27-tap convolutions with the x2 folded into the addressing (A/B, tests)


def _pad_bias(b: Optional[torch.Tensor], n_pad: int) -> Optional[torch.Tensor]:
    if b is None:
        return None
    b = b.float()
    if b.shape[0] != n_pad:
        b = torch.nn.functional.pad(b, (0, n_pad - b.shape[0]))
    return b.contiguous()


FLASH_MID_BLOCK = True   # False: mid-block attention as Q K^T GEMM -> fp32 logits -> row softmax -> P V GEMM per frame (A/B, tests)
# True: a GroupNorm whose one consumer is a 3x3x3 layer served by the four-wave row-slab kernels writes its output channel-blocked
# ([C / 32, T, H, W, 32]) and the convolution fills its slabs from 1 KiB pieces (bit-identical; False / EA_VAE_BLOCKED=0: A/B, tests)
BLOCKED_GN_OUTPUT = os.environ.get("EA_VAE_BLOCKED", "1") != "0"
VIRTUAL_TDUP = True   # False: SpatialTemporalUpsampler3D materialises its duplicated frames (A/B, tests)


def is_virtual(x: Optional[torch.Tensor]) -> bool:
    """x holds the physical frames of a virtually duplicated clip (logical frame f = physical (f + 1) >> 1)."""
    return bool(getattr(x, "tvirt", False))


def conv_cl(conv: nn.Conv3d, x: torch.Tensor, res: Optional[torch.Tensor] = None, ups: bool = False,
            tdup: bool = False) -> torch.Tensor:
    """Apply a (Causal)Conv3d module's parameters to a channels-last clip x [T,H,W,Cin] -> [T',H',W',Cout_pad8].
    Under a temporal split (vae_parallel) a rank that is not the first prepends its left neighbour's last frames and drops
    the outputs that belong to them: every retained output sees exactly the inputs of the whole-clip evaluation.
    x / res may be VIRTUAL clips (is_virtual): a 1x1x1 layer then runs on the physical frames and its output stays virtual, a
    3x3x3 layer addresses the logical frames inside the kernel and returns a real clip."""
    from . import vae_parallel
    tp = vae_parallel.current()
    if tp is None or conv.weight.shape[2] == 1:
        if conv.weight.shape[2] == 1 and is_virtual(x):
            assert res is None and not ups and not tdup
            y = _conv_cl_local(conv, x, None, False, False)
            y.tvirt = True
            return y
        return _conv_cl_local(conv, x, res, ups, tdup)
    assert not is_virtual(x) and not is_virtual(res), "virtual clips are not produced under a temporal split"
    st, ss = conv.stride[0], conv.stride[1]
    # ---- time: the left neighbour's last frames in front (none on the first temporal rank)
    drop = 0
    n = tp.halo_frames(st)
    halo = tp.exchange(x, n)
    if halo is not None:
        if res is not None:
            res = torch.cat([res.new_zeros((n,) + tuple(res.shape[1:])), res])
        x = torch.cat([halo, x])
        drop = tp.dropped_outputs(st, n, tdup)
    # ---- rows (spatial split): one input row from the neighbours above / below (stride 2: from below only), convolved
    # with the kernel's own zero padding at the outer edges; the rows this rank owns are kept
    keep = None
    if tp.ps > 1:
        h_loc = x.shape[1]
        above, below = tp.exchange_rows(x, 0 if ss == 2 else 1, 1)
        ra = above.shape[1] if above is not None else 0
        rb = below.shape[1] if below is not None else 0
        x = torch.cat([t for t in (above, x, below) if t is not None], dim=1)
        if res is not None:
            z = lambda r: res.new_zeros((res.shape[0], r) + tuple(res.shape[2:]))
            res = torch.cat([t for t in (z(ra) if ra else None, res, z(rb) if rb else None) if t is not None], dim=1).contiguous()
        if ss == 1:
            f = 2 if ups else 1
            keep = (f * ra, f * h_loc)
    y = _conv_cl_local(conv, x.contiguous(), res, ups, tdup)
    fused = getattr(y, "gn_partial", None)
    out = y[drop:]
    if keep is not None:
        out = out[:, keep[0]:keep[0] + keep[1]].contiguous()      # the fused statistics covered the halo rows: dropped
    elif fused is not None and tp.ps == 1:   # the per-frame partial sums of the retained frames
        partial, nblk = fused
        out.gn_partial = (partial.view(-1)[drop * nblk * (y.shape[-1] // 4) * 2:], nblk)
    return out


def _pack_conv_weight_c8(w: torch.Tensor, n_pad: int) -> torch.Tensor:
    """[Cout, Cin <= 8, 3, 3, 3] -> bf16 [n_pad, 256]: 32 tap slots x 8 channels (tap-major), zeros in the five unused tap
    slots, the padded channels and the padded rows -- the layout conv3d_cl_kernel<C8> reads (ea_conv3d_cl_bf16, C_in == 8)."""
    co, ci = w.shape[0], w.shape[1]
    out = torch.zeros((n_pad, 32, 8), dtype=torch.bfloat16, device=w.device)
    out[:co, :27, :ci] = w.permute(0, 2, 3, 4, 1).reshape(co, 27, ci).to(torch.bfloat16)
    return out.view(n_pad, 256)


def _wants_blocked(conv: nn.Conv3d, x: torch.Tensor) -> bool:
    """Will `conv` applied to the GroupNorm of x (voxel-major [T, H, W, C], possibly virtual) read a channel-blocked input?  Mirrors the
    route _conv_cl_local takes for a plain 3x3x3 / stride 1 / pad 1 layer (no up-sampling, no narrow output) with no VAE split."""
    from . import vae_parallel
    if not BLOCKED_GN_OUTPUT or vae_parallel.current() is not None or x.dim() != 4:
        return False
    co, ci, kt, kh, kw = conv.weight.shape
    if not (kt == kh == kw == 3 and tuple(conv.stride) == (1, 1, 1) and conv.padding[1] == 1 and ci % 64 == 0 and co > 4 and x.shape[-1] == ci):
        return False
    T = x.shape[0]
    Tl = 2 * T - 1 if (is_virtual(x) and T > 1) else T
    return ops.conv3d_blocked_ok(Tl, x.shape[1], x.shape[2], ci, ops.round_up(co, 8))


def _conv_cl_local(conv: nn.Conv3d, x: torch.Tensor, res: Optional[torch.Tensor] = None, ups: bool = False,
                   tdup: bool = False) -> torch.Tensor:
    co, ci, kt, kh, kw = conv.weight.shape
    assert kt == kh == kw and kt in (1, 3)
    st, sh, sw = conv.stride
    assert sh == sw
    pad = conv.padding[1] if kt == 3 else 0
    n_pad = ops.round_up(co, 8)
    if getattr(x, "cblocked", False):
        # a channel-blocked GroupNorm output [ci / 32, T, H, W, 32] (_gn(..., consumer=conv) asked _wants_blocked first)
        assert x.dim() == 5 and x.shape[0] * 32 == ci and kt == 3 and (st, sh, pad) == (1, 1, 1) and not ups and not tdup
        w = derived(conv.weight, f"cl{n_pad}", lambda t: _pack_conv_weight(t, None, n_pad))
        b = derived(conv.bias, f"b{n_pad}", lambda t: _pad_bias(t, n_pad)) if conv.bias is not None else None
        vin = is_virtual(x)
        T, H, W = x.shape[1:4]
        if vin and TEMPORAL_TAP_MERGE and T > 1 and ops.conv3d_tmerge_ok(2 * T - 1, H, W, ci, n_pad):
            wm = derived(conv.weight, f"tmerge{n_pad}", lambda t: _pack_tmerge_weight(t, n_pad))
            return ops.conv3d_cl(x, wm, b, kt, st, sh, pad, res=res, vin=True, vres=is_virtual(res), tmerge=True, blocked=True)
        return ops.conv3d_cl(x, w, b, kt, st, sh, pad, res=res, vin=vin, vres=is_virtual(res), blocked=True)
    c8 = ci <= 8 and kt == 3 and x.shape[-1] == 8 and res is None and not ups and not tdup
    if is_virtual(x) or is_virtual(res):
        assert ci % 64 == 0 and not (co <= 4 and kt == 3), "virtual clips feed the wide 3x3x3 / 1x1x1 layers of the up blocks only"
    if x.shape[-1] != ci and not c8:
        raise ValueError(f"conv expects {ci} input channels, got {x.shape[-1]}")
    if c8:
        # <= 8 input channels, handed over padded to 8 (one 16-byte chunk per voxel; the encoder's conv_in on RGB): the
        # implicit-GEMM kernel gathers eight taps per K tile -- no im2col buffer
        w = derived(conv.weight, f"c8_{n_pad}", lambda t: _pack_conv_weight_c8(t, n_pad))
        b = derived(conv.bias, f"b{n_pad}", lambda t: _pad_bias(t, n_pad)) if conv.bias is not None else None
        return ops.conv3d_cl(x, w, b, kt, st, sh, pad)
    if (ci % 64 == 0 and co <= 4 and kt == 3 and (st, sh, pad) == (1, 1, 1) and res is None and not ups and not tdup
            and (x.shape[0] * x.shape[1] * x.shape[2]) % 8 == 0):
        # narrow-N convolution (decoder conv_out 128 -> 3): one GEMM over the input voxels + a 27-tap gather
        wz = derived(conv.weight, "narrow", lambda t: t.permute(2, 3, 4, 0, 1).reshape(27 * co, ci).to(torch.bfloat16).contiguous())
        b = derived(conv.bias, "f32c", lambda t: t.float().contiguous()) if conv.bias is not None else None
        return ops.conv3d_narrow(x, wz, b, co, n_pad)
    if (SUBPIXEL_UPSAMPLE and ups and kt == 3 and (st, sh, pad) == (1, 1, 1) and res is None and ci % 64 == 0 and co % 256 == 0
            and x.shape[2] % 256 == 0 and not is_virtual(x)):
        # nearest x2 + 3x3x3 as four 12-tap parity classes on the source grid: 44 % of the MFMA work
        w4 = derived(conv.weight, "subpixel", _pack_subpixel_weight)
        b = derived(conv.bias, f"b{n_pad}", lambda t: _pad_bias(t, n_pad)) if conv.bias is not None else None
        return ops.conv3d_subpixel(x, w4, b, tdup=tdup)
    if ci % 64 == 0:
        w = derived(conv.weight, f"cl{n_pad}", lambda t: _pack_conv_weight(t, None, n_pad))
        b = derived(conv.bias, f"b{n_pad}", lambda t: _pad_bias(t, n_pad)) if conv.bias is not None else None
        vin = is_virtual(x) and kt == 3
        if (vin and TEMPORAL_TAP_MERGE and (st, sh, pad) == (1, 1, 1) and not ups and not tdup and x.shape[0] > 1
                and ops.conv3d_tmerge_ok(2 * x.shape[0] - 1, x.shape[1], x.shape[2], ci, n_pad)):
            # the duplicated frames make two of the three temporal taps coincide: 18 merged taps instead of 27
            wm = derived(conv.weight, f"tmerge{n_pad}", lambda t: _pack_tmerge_weight(t, n_pad))
            return ops.conv3d_cl(x, wm, b, kt, st, sh, pad, res=res, vin=True, vres=is_virtual(res), tmerge=True)
        return ops.conv3d_cl(x, w, b, kt, st, sh, pad, ups=ups, tdup=tdup, res=res, vin=vin, vres=is_virtual(res))
    assert not is_virtual(x) and not is_virtual(res)
    # small C_in: explicit im2col + GEMM
    assert res is None and not ups and not tdup
    k = kt * kh * kw * ci
    k_pad = ops.round_up(k, 64)
    w = derived(conv.weight, f"cl{n_pad}k{k_pad}", lambda t: _pack_conv_weight(t, k_pad, n_pad))
    b = derived(conv.bias, f"b{n_pad}", lambda t: _pad_bias(t, n_pad)) if conv.bias is not None else None
    cols, (To, Ho, Wo) = ops.im2col3d(x, kt, st, sh, pad, k_pad)
    y = ops.gemm(cols, w, b, ops.EPI_BIAS)
    return y.view(To, Ho, Wo, n_pad)


class CausalConv3d(nn.Conv3d):
    """reference: vaemodules/common.py:31-179.  Parameters as nn.Conv3d ([Cout,Cin,kt,kh,kw]); temporal padding is
    causal (kt-1 replicated leading frames), spatial padding `padding` (0 for the strided down-samplers)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size=3, stride=1, padding=1, dilation=1, **kwargs):
        kernel_size = kernel_size if isinstance(kernel_size, tuple) else (kernel_size,) * 3
        stride = stride if isinstance(stride, tuple) else (stride,) * 3
        dilation = dilation if isinstance(dilation, tuple) else (dilation,) * 3
        if dilation != (1, 1, 1):
            raise NotImplementedError("dilation != 1")
        if not isinstance(padding, int):
            raise NotImplementedError("padding must be an int (V5 configs use 0 or 1)")
        self.t_stride = stride[0]
        self.temporal_padding = kernel_size[0] - 1
        self.padding_flag = 0
        self.prev_features = None
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation,
                         padding=(0, padding, padding), **kwargs)

    def _clear_conv_cache(self):
        self.prev_features = None

    def forward(self, x: torch.Tensor, res: Optional[torch.Tensor] = None, ups: bool = False, tdup: bool = False):
        return conv_cl(self, x, res=res, ups=ups, tdup=tdup)


def _gn(norm: nn.GroupNorm, x: torch.Tensor, act: bool, consumer: Optional[nn.Conv3d] = None) -> torch.Tensor:
    """Per-frame GroupNorm: a virtual clip is normalised on its physical frames (duplicated frames have equal statistics).
    consumer: the ONE layer that reads the result; if it is a convolution that fills its slabs faster from a channel-blocked
    tensor, the result is written that way (y.cblocked)."""
    from . import vae_parallel
    tp = vae_parallel.current()
    if consumer is not None and _wants_blocked(consumer, x):
        y = ops.groupnorm_silu(x, f32(norm.weight), f32(norm.bias), norm.num_groups, norm.eps, act=act, blocked=True)
        y.cblocked = True
        if is_virtual(x):
            y.tvirt = True
        return y
    if tp is not None and tp.ps > 1:
        # spatial split: the statistics are over the WHOLE frame -- all-reduce the additive half over the frame's ranks
        sums = tp.all_reduce_rows(ops.groupnorm_local_sums(x, norm.num_groups))
        n = (x.numel() // (x.shape[0] * x.shape[-1])) * tp.ps * (x.shape[-1] // norm.num_groups)
        stats = ops.groupnorm_stats_from_sums(sums, n, norm.eps)
        return ops.groupnorm_apply(x, stats, f32(norm.weight), f32(norm.bias), norm.num_groups, act=act)
    y = ops.groupnorm_silu(x, f32(norm.weight), f32(norm.bias), norm.num_groups, norm.eps, act=act)
    if is_virtual(x):
        y.tvirt = True
    return y


class ResidualBlock3D(nn.Module):
    """reference: vaemodules/common.py:254-323 (per-frame GroupNorm, i.e. set_3dgroupnorm / spatial_group_norm)."""

    def __init__(self, in_channels: int, out_channels: int, non_linearity: str = "silu", norm_num_groups: int = 32,
                 norm_eps: float = 1e-6, dropout: float = 0.0, output_scale_factor: float = 1.0):
        super().__init__()
        if non_linearity not in ("silu", "swish") or output_scale_factor != 1.0:
            raise NotImplementedError("only SiLU / output_scale_factor 1 (V5 config)")
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=norm_eps, affine=True)
        self.conv1 = CausalConv3d(in_channels, out_channels, kernel_size=3)
        self.norm2 = nn.GroupNorm(num_groups=norm_num_groups, num_channels=out_channels, eps=norm_eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = CausalConv3d(out_channels, out_channels, kernel_size=3)
        if in_channels != out_channels:
            self.shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1)
        else:
            self.shortcut = nn.Identity()
        self.set_3dgroupnorm = True

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shortcut = x if isinstance(self.shortcut, nn.Identity) else conv_cl(self.shortcut, x)
        h = _gn(self.norm1, x, act=True, consumer=self.conv1)
        h = self.conv1(h)
        h = _gn(self.norm2, h, act=True, consumer=self.conv2)
        return self.conv2(h, res=shortcut)  # (x + shortcut) / 1.0 fused in the conv epilogue


# ---- samplers -------------------------------------------------------------------------------------
class SpatialDownsampler3D(nn.Module):
    """downsamplers.py:24-47: F.pad(x,(0,1,0,1)) + CausalConv3d(k3, stride (1,2,2), padding 0)."""

    def __init__(self, in_channels: int, out_channels):
        super().__init__()
        self.conv = CausalConv3d(in_channels, out_channels or in_channels, kernel_size=3, stride=(1, 2, 2), padding=0)

    def forward(self, x):
        return self.conv(x)


class SpatialTemporalDownsampler3D(nn.Module):
    """downsamplers.py:72-94: stride (2,2,2)."""

    def __init__(self, in_channels: int, out_channels):
        super().__init__()
        self.conv = CausalConv3d(in_channels, out_channels or in_channels, kernel_size=3, stride=(2, 2, 2), padding=0)

    def forward(self, x):
        return self.conv(x)


class SpatialUpsampler3D(nn.Module):
    """upsamplers.py:21-37: nearest x2 (H,W) then conv -- the up-sampling is folded into the conv's addressing."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = CausalConv3d(in_channels, out_channels or in_channels, kernel_size=3)

    def forward(self, x):
        return self.conv(x, ups=True)


class SpatialTemporalUpsampler3D(nn.Module):
    """upsamplers.py:123-153: nearest x2 (H,W), conv, then temporal nearest x2 of every frame but the first
    (set_3dgroupnorm => mode "nearest"); the duplication is a second store in the conv epilogue."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = CausalConv3d(in_channels, out_channels or in_channels, kernel_size=3)
        self.padding_flag = 0
        self.set_3dgroupnorm = True

    def forward(self, x):
        from . import vae_parallel
        if VIRTUAL_TDUP and vae_parallel.current() is None and x.shape[0] > 1:
            # the duplicated frames are never written: the consumers (the next block's GroupNorm, 1x1x1 shortcut, first
            # 3x3x3 convolution and the residual add of its second one) address frame (t + 1) >> 1
            y = self.conv(x, ups=True, tdup=False)
            y.tvirt = True
            return y
        return self.conv(x, ups=True, tdup=True)


# ---- blocks ----------------------------------------------------------------------------------------
def _res_stack(in_channels, out_channels, num_layers, norm_num_groups, norm_eps, dropout):
    return nn.ModuleList([ResidualBlock3D(in_channels if i == 0 else out_channels, out_channels,
                                          norm_num_groups=norm_num_groups, norm_eps=norm_eps, dropout=dropout)
                          for i in range(num_layers)])


class SpatialDownBlock3D(nn.Module):
    """down_blocks.py:156-211"""

    def __init__(self, in_channels, out_channels, num_layers=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-6,
                 dropout=0.0, output_scale_factor=1.0, add_gc_block=False, add_downsample=True):
        super().__init__()
        if add_gc_block:
            raise NotImplementedError("GlobalContextBlock is unused by the V5 config")
        self.convs = _res_stack(in_channels, out_channels, num_layers, norm_num_groups, norm_eps, dropout)
        self.gc_block = None
        self.downsampler = SpatialDownsampler3D(out_channels, out_channels) if add_downsample else None
        self.spatial_downsample_factor = 2 if add_downsample else 1
        self.temporal_downsample_factor = 1

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return self.downsampler(x) if self.downsampler is not None else x


class SpatialTemporalDownBlock3D(nn.Module):
    """down_blocks.py:272-327"""

    def __init__(self, in_channels, out_channels, num_layers=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-6,
                 dropout=0.0, output_scale_factor=1.0, add_gc_block=False, add_downsample=True):
        super().__init__()
        if add_gc_block:
            raise NotImplementedError("GlobalContextBlock is unused by the V5 config")
        self.convs = _res_stack(in_channels, out_channels, num_layers, norm_num_groups, norm_eps, dropout)
        self.gc_block = None
        self.downsampler = SpatialTemporalDownsampler3D(out_channels, out_channels) if add_downsample else None
        self.spatial_downsample_factor = 2 if add_downsample else 1
        self.temporal_downsample_factor = 2 if add_downsample else 1

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return self.downsampler(x) if self.downsampler is not None else x


class SpatialUpBlock3D(nn.Module):
    """up_blocks.py:96-147"""

    def __init__(self, in_channels, out_channels, num_layers=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-6,
                 dropout=0.0, output_scale_factor=1.0, add_gc_block=False, add_upsample=True):
        super().__init__()
        if add_gc_block:
            raise NotImplementedError("GlobalContextBlock is unused by the V5 config")
        self.upsampler = SpatialUpsampler3D(in_channels, in_channels) if add_upsample else None
        self.gc_block = None
        self.convs = _res_stack(in_channels, out_channels, num_layers, norm_num_groups, norm_eps, dropout)

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return self.upsampler(x) if self.upsampler is not None else x


class SpatialTemporalUpBlock3D(nn.Module):
    """up_blocks.py:344-395"""

    def __init__(self, in_channels, out_channels, num_layers=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-6,
                 dropout=0.0, output_scale_factor=1.0, add_gc_block=False, add_upsample=True):
        super().__init__()
        if add_gc_block:
            raise NotImplementedError("GlobalContextBlock is unused by the V5 config")
        self.convs = _res_stack(in_channels, out_channels, num_layers, norm_num_groups, norm_eps, dropout)
        self.gc_block = None
        self.upsampler = SpatialTemporalUpsampler3D(out_channels, out_channels) if add_upsample else None

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return self.upsampler(x) if self.upsampler is not None else x


_DOWN = {"SpatialDownBlock3D": SpatialDownBlock3D, "SpatialTemporalDownBlock3D": SpatialTemporalDownBlock3D}
_UP = {"SpatialUpBlock3D": SpatialUpBlock3D, "SpatialTemporalUpBlock3D": SpatialTemporalUpBlock3D}


def get_down_block(down_block_type, **kw):
    if down_block_type not in _DOWN:
        raise NotImplementedError(f"{down_block_type}: only the V5 block types {sorted(_DOWN)} are built")
    kw.pop("num_attention_heads", None)
    return _DOWN[down_block_type](**kw)


def get_up_block(up_block_type, **kw):
    if up_block_type not in _UP:
        raise NotImplementedError(f"{up_block_type}: only the V5 block types {sorted(_UP)} are built")
    kw.pop("num_attention_heads", None)
    return _UP[up_block_type](**kw)


# ---- mid block ---------------------------------------------------------------------------------------
class SpatialAttention(nn.Module):
    """Single-head per-frame attention of the mid block (vaemodules/attention.py:12-423 `Attention` with
    residual_connection=True + attention_processors.py:76-139).  Keys: group_norm, to_q, to_k, to_v, to_out.
    head_dim = C (512): logits = Q K^T via MFMA GEMM into fp32, row softmax, P V via GEMM."""

    def __init__(self, query_dim: int, nheads: int = 1, head_dim: int = 64, bias: bool = True, upcast_softmax: bool = True,
                 norm_num_groups: int = 32, eps: float = 1e-6, rescale_output_factor: float = 1.0,
                 residual_connection: bool = True, **unused):
        super().__init__()
        if nheads != 1 or rescale_output_factor != 1.0 or not residual_connection:
            raise NotImplementedError("only the V5 mid-block configuration (1 head, residual, no rescale)")
        self.inner_dim = head_dim * nheads
        self.nheads = nheads
        self.scale = head_dim ** -0.5
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_out = nn.Linear(self.inner_dim, query_dim, bias=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ._params import bf16_weight
        T, H, W, C = x.shape
        n = H * W
        xr = x.view(T, n, C)
        xn = _gn(self.group_norm, xr, act=False)
        # the P.V product runs over K = n keys and the GEMMs want K % 64 == 0 / N % 8 == 0: 576x1008 gives n = 9072, 480x720
        # gives 5400.  Keys / values are then padded to n_pad rows of zeros; the pad columns of the logits are set
        # to -inf before the row softmax (probability exactly 0) and the pad columns of V^T are zero (to_v's bias is folded out).
        from . import vae_parallel
        tp = vae_parallel.current()
        xa = xn
        if tp is not None and tp.ps > 1:
            xa = tp.all_gather_rows(xn, 1)      # spatial split: queries = this rank's rows, keys / values = the whole frame
        n_keys = xa.shape[1]
        n_pad = ops.round_up(n_keys, 64)
        if n_pad != n_keys:
            xk = torch.zeros((T, n_pad, C), dtype=xn.dtype, device=xn.device)
            xk[:, :n_keys] = xa
        else:
            xk = xa
        q = ops.gemm(xn, bf16_weight(self.to_q.weight), f32(self.to_q.bias), ops.EPI_BIAS)
        k = ops.gemm(xk, bf16_weight(self.to_k.weight), f32(self.to_k.bias), ops.EPI_BIAS)
        wv = bf16_weight(self.to_v.weight)
        # softmax rows sum to 1, so the value bias passes straight through attention: fold it into the out bias
        wo = bf16_weight(self.to_out.weight)
        b_eff = f32(self.to_out.bias)
        if self.to_v.bias is not None:  # b_eff = W_o b_v + b_o  (GEMV kernel)
            b_eff = ops.linear_small_m(f32(self.to_v.bias).view(1, -1), wo, b_eff).view(-1)
        ones = derived(self.to_out.bias, "ones", lambda t: torch.ones(1, t.shape[0], dtype=torch.float32, device=t.device))
        out = torch.empty_like(xr)
        if self.inner_dim == 512 and C == 512 and FLASH_MID_BLOCK:
            # one flash launch for all frames (ea_attention_d512_fwd_bf16): no [n, n] logits buffer, no softmax pass
            vt = torch.empty((T, C, n_pad), dtype=xn.dtype, device=xn.device)
            for f in range(T):
                ops.gemm(wv, xk[f], None, ops.EPI_BIAS, out=vt[f])                # V^T [C, n_pad] of the frame
            o = ops.attention_d512(q, k, vt, n_keys, self.scale)
            ops.gemm(o, wo, b_eff, ops.EPI_BIAS_GATE_RES, out=out, res=xr, gate=ones.expand(T, C))
            return out.view(T, H, W, C)
        logits = torch.empty((n, n_pad), dtype=torch.float32, device=x.device)
        for f in range(T):  # one frame at a time: [n, n] fp32 logits (1 GiB at 1024^2) stay a reusable buffer
            vt = ops.gemm(wv, xk[f], None, ops.EPI_BIAS)                      # V^T [C, n_pad]
            ops.gemm(q[f], k[f], None, ops.EPI_F32_OUT, out=logits)           # Q K^T
            if n_pad != n_keys:
                logits[:, n_keys:] = float("-inf")
            p = ops.softmax_rows(logits, self.scale)
            o = ops.gemm(p, vt, None, ops.EPI_BIAS)                            # P V  [n, C]
            ops.gemm(o, wo, b_eff, ops.EPI_BIAS_GATE_RES, out=out[f], res=xr[f], gate=ones)
        return out.view(T, H, W, C)


class MidBlock3D(nn.Module):
    """mid_blocks.py:38-196 with attention_type="spatial": convs[0], then (attention, conv) pairs."""

    def __init__(self, in_channels: int, num_layers: int = 1, act_fn: str = "silu", norm_num_groups: int = 32,
                 norm_eps: float = 1e-6, dropout: float = 0.0, add_attention: bool = True, attention_type: str = "3d",
                 attention_head_dim: int = 1, output_scale_factor: float = 1.0):
        super().__init__()
        if add_attention and attention_type != "spatial":
            raise NotImplementedError("only mid_block_attention_type='spatial' (V5 config)")
        self.attention_type = attention_type
        self.convs = nn.ModuleList([ResidualBlock3D(in_channels, in_channels, norm_num_groups=norm_num_groups,
                                                    norm_eps=norm_eps, dropout=dropout)])
        self.attentions = nn.ModuleList([])
        for _ in range(num_layers - 1):
            if add_attention:
                self.attentions.append(SpatialAttention(in_channels, nheads=in_channels // attention_head_dim,
                                                        head_dim=attention_head_dim, bias=True, upcast_softmax=True,
                                                        norm_num_groups=norm_num_groups, eps=norm_eps,
                                                        rescale_output_factor=output_scale_factor, residual_connection=True))
            else:
                self.attentions.append(None)
            self.convs.append(ResidualBlock3D(in_channels, in_channels, norm_num_groups=norm_num_groups,
                                              norm_eps=norm_eps, dropout=dropout))

    def forward(self, x):
        x = self.convs[0](x)
        for attn, resnet in zip(self.attentions, self.convs[1:]):
            if attn is not None:
                x = attn(x)
            x = resnet(x)
        return x


def get_mid_block(mid_block_type, in_channels, num_layers, act_fn, norm_num_groups=32, norm_eps=1e-6, dropout=0.0,
                  add_attention=True, attention_type="3d", num_attention_heads=1, output_scale_factor=1.0):
    if mid_block_type != "MidBlock3D":
        raise ValueError(f"Unknown mid block type: {mid_block_type}")
    return MidBlock3D(in_channels=in_channels, num_layers=num_layers, act_fn=act_fn, norm_num_groups=norm_num_groups,
                      norm_eps=norm_eps, dropout=dropout, add_attention=add_attention, attention_type=attention_type,
                      attention_head_dim=in_channels // num_attention_heads, output_scale_factor=output_scale_factor)
