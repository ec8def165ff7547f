"""TEST INFRASTRUCTURE ONLY -- the parity oracle.  Never imported by easyanimate_amd/ (the product).

A plain-PyTorch CPU restatement of the reference's diffusion-sampling hot path, written as pure
functions over a state dict (reference key names).  Each function cites the
reference lines it follows.  It is pinned in two ways (tests/test_oracle_cpu.py):
  1. against the golden vectors in tests/golden/ that oracle/gen_golden.py produced by executing the
     UNCHANGED reference modules from /root/reference (through oracle/diffusers_shim.py), and
  2. directly against the live reference whenever /root/reference is present.
Caveat ("parity unpinned" at the diffusers boundary): the reference has no tests or golden vectors of
its own, and diffusers 0.30/0.31 is not installable here, so the diffusers pieces
(Attention, FeedForward, apply_rotary_emb, Timesteps, AdaLayerNorm, FlowMatchEulerDiscreteScheduler)
are restated from their published behaviour in the shim and here.

dtype policy: every function computes in the dtype of its inputs with the same rounding points as the
reference modules (F.linear / F.layer_norm on bf16 tensors round their outputs to bf16, FP32LayerNorm
up-casts, RoPE is fp32 then cast), so `dtype=torch.bfloat16` reproduces the reference's bf16 CPU path
and `torch.float32` is the exact oracle.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------------------------
# embeddings / tables
# ---------------------------------------------------------------------------------------------
def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0) as called at
    easyanimate/models/transformer3d.py:1399,1519."""
    half = dim // 2
    e = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * e[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def time_embedding(sd: SD, t: torch.Tensor, inner_dim: int, dtype) -> torch.Tensor:
    """transformer3d.py:1519-1520 (+ diffusers TimestepEmbedding: linear_2(silu(linear_1(x))))."""
    x = timestep_sinusoid(t, inner_dim).to(dtype)
    x = F.linear(x, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    x = F.silu(x)
    return F.linear(x, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """easyanimate/pipeline/pipeline_easyanimate.py:82-97"""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        rh, rw = th, int(round(th / h * w))
    else:
        rw, rh = tw, int(round(tw / w * h))
    top, left = int(round((th - rh) / 2.0)), int(round((tw - rw) / 2.0))
    return (top, left), (top + rh, left + rw)


def rope_3d(embed_dim: int, crops_coords, grid_size, temporal_size: int, theta: float = 10000.0):
    """diffusers get_3d_rotary_pos_embed, called at pipeline_easyanimate.py:1008."""
    start, stop = crops_coords
    gh, gw = grid_size

    def one(dim, pos):
        pos = torch.from_numpy(np.asarray(pos, dtype=np.float32))
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(pos, freqs)
        return ang.cos().repeat_interleave(2, 1).float(), ang.sin().repeat_interleave(2, 1).float()

    gh_ = np.linspace(start[0], stop[0], gh, endpoint=False, dtype=np.float32)
    gw_ = np.linspace(start[1], stop[1], gw, endpoint=False, dtype=np.float32)
    gt_ = np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32)
    ft, fh, fw = one(embed_dim // 4, gt_), one(embed_dim // 8 * 3, gh_), one(embed_dim // 8 * 3, gw_)

    def comb(a, b, c):
        a = a[:, None, None, :].expand(-1, gh, gw, -1)
        b = b[None, :, None, :].expand(temporal_size, -1, gw, -1)
        c = c[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([a, b, c], -1).reshape(temporal_size * gh * gw, -1)

    return comb(ft[0], fh[0], fw[0]), comb(ft[1], fh[1], fw[1])


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """diffusers apply_rotary_emb(use_real=True, unbind_dim=-1), called at processor.py:283-285."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


# ---------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------
def fp32_layernorm(x, w, b, eps):
    """easyanimate/models/norm.py:16-26"""
    return F.layer_norm(x.float(), (x.shape[-1],), None if w is None else w.float(), None if b is None else b.float(),
                        eps).to(x.dtype)


def rmsnorm(x, w, eps=1e-6):
    """easyanimate/models/norm.py:28-42"""
    dt = x.dtype
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(dt)


def layernorm_zero(sd: SD, pre: str, h, e, temb, eps):
    """EasyAnimateLayerNormZero.forward, norm.py:160-166"""
    mod = F.linear(F.silu(temb), sd[pre + "linear.weight"], sd[pre + "linear.bias"])
    shift, scale, gate, e_shift, e_scale, e_gate = mod.chunk(6, dim=1)
    w, b = sd.get(pre + "norm.weight"), sd.get(pre + "norm.bias")
    h = fp32_layernorm(h, w, b, eps) * (1 + scale)[:, None, :] + shift[:, None, :]
    e = fp32_layernorm(e, w, b, eps) * (1 + e_scale)[:, None, :] + e_shift[:, None, :]
    return h, e, gate[:, None, :], e_gate[:, None, :]


# ---------------------------------------------------------------------------------------------
# attention processor / block / transformer
# ---------------------------------------------------------------------------------------------
def attn_processor(sd: SD, pre1: str, pre2: Optional[str], h, e, rope, heads: int, qk_eps: float = 1e-6):
    """EasyAnimateAttnProcessor2_0.__call__, processor.py:222-312 (attention_mask is None on this path)."""
    T = e.shape[1]
    B = h.shape[0]
    if pre2 is None:
        h = torch.cat([e, h], dim=1)

    def qkv(pre, x):
        q = F.linear(x, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"])
        k = F.linear(x, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"])
        v = F.linear(x, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"])
        dh = q.shape[-1] // heads
        q, k, v = [t.view(B, -1, heads, dh).transpose(1, 2) for t in (q, k, v)]
        q = F.layer_norm(q, (dh,), sd[pre + "norm_q.weight"], sd[pre + "norm_q.bias"], qk_eps)
        k = F.layer_norm(k, (dh,), sd[pre + "norm_k.weight"], sd[pre + "norm_k.bias"], qk_eps)
        return q, k, v

    q, k, v = qkv(pre1, h)
    if pre2 is not None:
        qt, kt, vt = qkv(pre2, e)
        q, k, v = torch.cat([qt, q], 2), torch.cat([kt, k], 2), torch.cat([vt, v], 2)
    if rope is not None:
        q = torch.cat([q[:, :, :T], apply_rotary_emb(q[:, :, T:], *rope)], 2)
        k = torch.cat([k[:, :, :T], apply_rotary_emb(k[:, :, T:], *rope)], 2)
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, q.shape[1] * q.shape[-1])
    if pre2 is None:
        o = F.linear(o, sd[pre1 + "to_out.0.weight"], sd[pre1 + "to_out.0.bias"])
        return o[:, T:], o[:, :T]
    oe, oh = o[:, :T], o[:, T:]
    oh = F.linear(oh, sd[pre1 + "to_out.0.weight"], sd[pre1 + "to_out.0.bias"])
    oe = F.linear(oe, sd[pre2 + "to_out.0.weight"], sd[pre2 + "to_out.0.bias"])
    return oh, oe


_SWA_ORDERS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))   # (f h w) -> the six scan orders, :405-411


def swa_attn_processor(sd: SD, pre1: str, pre2: Optional[str], h, e, rope, heads: int, grid, qk_eps: float = 1e-6,
                       cross_attention_size: int = 1024):
    """EasyAnimateSWAttnProcessor2_0.__call__, processor.py:326-459, with flash_attn_func as restated in
    oracle/flash_attn_shim.py (the package is absent: that call is "parity unpinned")."""
    from .flash_attn_shim import flash_attn_func
    Fr, Hh, Ww = grid
    T, B = e.shape[1], h.shape[0]
    if pre2 is None:
        h = torch.cat([e, h], dim=1)

    def qkv(pre, x):
        q = F.linear(x, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"]).view(B, -1, heads, 64).transpose(1, 2)
        k = F.linear(x, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"]).view(B, -1, heads, 64).transpose(1, 2)
        v = F.linear(x, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"]).view(B, -1, heads, 64)
        q = F.layer_norm(q, (64,), sd[pre + "norm_q.weight"], sd[pre + "norm_q.bias"], qk_eps)
        k = F.layer_norm(k, (64,), sd[pre + "norm_k.weight"], sd[pre + "norm_k.bias"], qk_eps)
        return q, k, v

    q, k, v = qkv(pre1, h)
    if pre2 is not None:
        qt, kt, vt = qkv(pre2, e)
        q, k, v = torch.cat([qt, q], 2), torch.cat([kt, k], 2), torch.cat([vt, v], 1)
    if rope is not None:
        q = torch.cat([q[:, :, :T], apply_rotary_emb(q[:, :, T:], *rope)], 2)
        k = torch.cat([k[:, :, :T], apply_rotary_emb(k[:, :, T:], *rope)], 2)
    q, k = q.transpose(1, 2).to(v.dtype), k.transpose(1, 2).to(v.dtype)          # [B, S, heads, 64]
    N = q.shape[1] - T
    interval = max(N // (cross_attention_size - T), 1)                             # :393
    cross = flash_attn_func(q, torch.cat([k[:, :T], k[:, T::interval]], 1), torch.cat([v[:, :T], v[:, T::interval]], 1))
    base = torch.arange(N).view(Fr, Hh, Ww)
    qs, ks, vs = [torch.tensor_split(t[:, T:], 6, 2) for t in (q, k, v)]
    srcs = [base.permute(*o).reshape(-1) for o in _SWA_ORDERS]
    qn, kn, vn = [torch.cat([part[:, src] for part, src in zip(parts, srcs)], dim=2) for parts in (qs, ks, vs)]
    win = flash_attn_func(qn, kn, vn, window_size=(Hh * Ww, Hh * Ww))              # :420
    outs = []
    for part, src in zip(torch.tensor_split(win, 6, 2), srcs):
        back = torch.empty_like(part)
        back[:, src] = part
        outs.append(back)
    o = torch.cat([cross[:, :T], torch.cat(outs, dim=2)], dim=1) + cross            # :435
    o = o.reshape(B, -1, heads * 64)
    if pre2 is None:
        o = F.linear(o, sd[pre1 + "to_out.0.weight"], sd[pre1 + "to_out.0.bias"])
        return o[:, T:], o[:, :T]
    return (F.linear(o[:, T:], sd[pre1 + "to_out.0.weight"], sd[pre1 + "to_out.0.bias"]),
            F.linear(o[:, :T], sd[pre2 + "to_out.0.weight"], sd[pre2 + "to_out.0.bias"]))


def feed_forward(sd: SD, pre: str, x):
    """diffusers FeedForward('gelu-approximate'): net.0.proj -> gelu(tanh) -> net.2 (attention.py:1082-1098)"""
    x = F.gelu(F.linear(x, sd[pre + "net.0.proj.weight"], sd[pre + "net.0.proj.bias"]), approximate="tanh")
    return F.linear(x, sd[pre + "net.2.weight"], sd[pre + "net.2.bias"])


def dit_block(sd: SD, pre: str, h, e, temb, rope, heads: int, norm_eps: float, return_parts: bool = False,
              after_norm: bool = False, swa_grid=None):
    """EasyAnimateDiTBlock.forward, easyanimate/models/attention.py:1107-1163 (not SWA; after_norm is detected from the
    norm3.* keys, or forced for an affine-free norm3)."""
    mmdit = (pre + "attn2.to_q.weight") in sd
    nh, ne, gate, egate = layernorm_zero(sd, pre + "norm1.", h, e, temb, norm_eps)
    if swa_grid is not None:      # is_swa block (attention.py:1124-1132)
        ah, ae = swa_attn_processor(sd, pre + "attn1.", pre + "attn2." if mmdit else None, nh, ne, rope, heads, swa_grid)
    else:
        ah, ae = attn_processor(sd, pre + "attn1.", pre + "attn2." if mmdit else None, nh, ne, rope, heads)
    h = h + gate * ah
    e = e + egate * ae
    nh, ne, gate_ff, egate_ff = layernorm_zero(sd, pre + "norm2.", h, e, temb, norm_eps)
    fh = feed_forward(sd, pre + "ff.", nh)
    fe = feed_forward(sd, pre + ("txt_ff." if (pre + "txt_ff.net.2.weight") in sd else "ff."), ne)
    if (pre + "norm3.weight") in sd or after_norm:      # after_norm, attention.py:1150-1155 (FP32LayerNorm)
        fh = fp32_layernorm(fh, sd.get(pre + "norm3.weight"), sd.get(pre + "norm3.bias"), norm_eps)
        fe = fp32_layernorm(fe, sd.get(pre + "norm3.weight"), sd.get(pre + "norm3.bias"), norm_eps)
    h2 = h + gate_ff * fh
    e2 = e + egate_ff * fe
    if return_parts:
        return h2, e2, dict(attn_h=ah, attn_e=ae, ff_h=fh, ff_e=fe)
    return h2, e2


class TeaCache:
    """The step-skip heuristic of transformer3d.py:90-121 and its use at :1564-1590, 1635.
    rel-L1 distance between consecutive `transformer_blocks[0].norm1` outputs (video stream, whole CFG batch),
    rescaled by np.poly1d(coefficients) and accumulated; a step is skipped while the sum stays below the threshold.
    The reference evaluates the distance in the MODEL dtype (bf16 tensors: |cur-prev| is rounded per element, each
    mean and the quotient are rounded to bf16) before `.item()`; with fp32 tensors nothing is rounded."""

    def __init__(self, coefficients, num_steps: int, rel_l1_thresh: float = 0.0):
        self.coefficients, self.num_steps, self.rel_l1_thresh = list(coefficients), num_steps, rel_l1_thresh
        self.cnt = 0
        self.accumulated = 0.0
        self.prev_mod = None
        self.prev_residual = None

    def rescale(self, x: float) -> float:
        return float(np.poly1d(self.coefficients)(x))

    @staticmethod
    def rel_l1(prev: torch.Tensor, cur: torch.Tensor) -> float:
        return ((cur - prev).abs().mean() / prev.abs().mean()).item()

    def should_calc(self, mod: torch.Tensor):
        """-> (should_calc, rel_l1 or None); advances the step counter exactly as :1569-1584."""
        dist = None
        if self.cnt == 0 or self.cnt == self.num_steps - 1:
            calc = True
            self.accumulated = 0.0
        else:
            dist = self.rel_l1(self.prev_mod, mod)
            self.accumulated += self.rescale(dist)
            calc = not (self.accumulated < self.rel_l1_thresh)
            if calc:
                self.accumulated = 0.0
        self.prev_mod = mod
        self.cnt += 1
        if self.cnt == self.num_steps:
            self.cnt, self.prev_mod, self.prev_residual = 0, None, None   # reset(): `accumulated` is kept
        return calc, dist


def text_projection(sd: SD, pre: str, enc):
    """text_proj / text_proj_t5, transformer3d.py:1405-1418,1533-1535: Linear, or RMSNorm -> Linear (add_norm_text_encoder)."""
    if pre + ".0.weight" in sd:
        return F.linear(rmsnorm(enc, sd[pre + ".0.weight"]), sd[pre + ".1.weight"], sd[pre + ".1.bias"])
    return F.linear(enc, sd[pre + ".weight"], sd[pre + ".bias"])


def sincos_2d(embed_dim: int, gh: int, gw: int, base_size: int = 16):
    """diffusers get_2d_sincos_pos_embed [restated] as called at transformer3d.py:1424: float64 [gh*gw, embed_dim]."""
    ch = np.arange(gh, dtype=np.float32) / (gh / base_size)
    cw = np.arange(gw, dtype=np.float32) / (gw / base_size)
    grid = np.stack(np.meshgrid(cw, ch), axis=0).reshape(2, -1)

    def one(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos, omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    return torch.from_numpy(np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1))


def ref_clip_tokens(sd: SD, cfg: dict, ref_latents, clip_states, gh: int, gw: int, dtype):
    """transformer3d.py:1538-1561: ref_proj patch embedding + the 2-D sin/cos table resized (trilinear) to the latent grid;
    CLIP tokens (clip_proj) are prepended.  The result REPLACES the text stream."""
    inner = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    p = cfg["patch_size"]
    B, C, Fr, H, W = ref_latents.shape
    r = F.conv2d(ref_latents.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W), sd["ref_proj.weight"], sd["ref_proj.bias"], stride=p)
    r = r.reshape(B, Fr, inner, H // p, W // p).permute(0, 2, 1, 3, 4).flatten(2).transpose(1, 2)
    pph, ppw = cfg.get("sample_height", 60) // p, cfg.get("sample_width", 90) // p
    pe = sincos_2d(inner, pph, ppw).to(dtype).view(1, 1, pph, ppw, inner).permute(0, 4, 1, 2, 3)
    pe = F.interpolate(pe, size=[1, gh, gw], mode="trilinear", align_corners=False).permute(0, 2, 3, 4, 1).reshape(1, -1, inner)
    e = r + pe
    if clip_states is not None:
        e = torch.cat([F.linear(clip_states, sd["clip_proj.weight"], sd["clip_proj.bias"]), e], dim=1)
    return e


def transformer_forward(sd: SD, cfg: dict, latents, timestep, enc, rope, inpaint_latents=None, teacache=None,
                        control_latents=None, enc_t5=None, ref_latents=None, clip_states=None):
    """EasyAnimateTransformer3DModel.forward, transformer3d.py:1496-1689."""
    heads, dh = cfg["num_attention_heads"], cfg["attention_head_dim"]
    inner = heads * dh
    p = cfg["patch_size"]
    B, C, Fr, H, W = latents.shape
    dtype = latents.dtype
    temb = time_embedding(sd, timestep, inner, dtype)
    x = latents if inpaint_latents is None else torch.cat([latents, inpaint_latents], 1)      # :1523-1524
    if control_latents is not None:
        x = torch.cat([x, control_latents], 1)                                                  # :1525-1526
    x = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, x.shape[1], H, W)
    x = F.conv2d(x, sd["proj.weight"], sd["proj.bias"], stride=p)
    x = x.reshape(B, Fr, inner, H // p, W // p).permute(0, 2, 1, 3, 4).flatten(2).transpose(1, 2)
    e = text_projection(sd, "text_proj", enc)
    if enc_t5 is not None:                                                                      # :1534-1536
        e = torch.cat([e, text_projection(sd, "text_proj_t5", enc_t5)], dim=1)
    if ref_latents is not None:
        e = ref_clip_tokens(sd, cfg, ref_latents, clip_states, H // p, W // p, dtype)
    calc = True
    if teacache is not None:
        mod_in, _, _, _ = layernorm_zero(sd, "transformer_blocks.0.norm1.", x, e, temb, cfg["norm_eps"])
        nxt = teacache.prev_residual
        calc, _ = teacache.should_calc(mod_in)
        if not calc:
            x = x + nxt                      # :1590 (the residual recorded by the last computed step)
    if calc:
        x_in = x
        for i in range(cfg["num_layers"]):
            x, e = dit_block(sd, f"transformer_blocks.{i}.", x, e, temb, rope, heads, cfg["norm_eps"],
                             after_norm=bool(cfg.get("after_norm", False)),
                             swa_grid=(Fr, H // p, W // p) if i in (cfg.get("swa_layers") or ()) else None)
        T = e.shape[1]
        x = torch.cat([e, x], dim=1)
        x = F.layer_norm(x, (inner,), sd.get("norm_final.weight"), sd.get("norm_final.bias"), cfg["norm_eps"])
        x = x[:, T:]
        mod = F.linear(F.silu(temb), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"])
        shift, scale = mod.chunk(2, dim=1)
        x = F.layer_norm(x, (inner,), sd.get("norm_out.norm.weight"), sd.get("norm_out.norm.bias"), cfg["norm_eps"])
        x = x * (1 + scale[:, None, :]) + shift[:, None, :]
        if teacache is not None:
            teacache.prev_residual = x - x_in            # :1635
    x = F.linear(x, sd["proj_out.weight"], sd["proj_out.bias"])
    out = x.reshape(B, Fr, H // p, W // p, C, p, p).permute(0, 4, 1, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    return out


# ---------------------------------------------------------------------------------------------
# scheduler + denoise loop
# ---------------------------------------------------------------------------------------------
def flow_sigmas(num_inference_steps: int, shift: float = 1.0, use_dynamic_shifting: bool = False,
                mu: Optional[float] = None, num_train_timesteps: int = 1000):
    """diffusers FlowMatchEulerDiscreteScheduler.__init__ + set_timesteps.
    Returns (timesteps fp32 [n], sigmas fp32 [n+1])."""
    ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
    s0 = torch.from_numpy(ts).float() / num_train_timesteps
    if not use_dynamic_shifting:
        s0 = shift * s0 / (1 + (shift - 1) * s0)
    smax, smin = s0[0].item(), s0[-1].item()
    t = np.linspace(smax * num_train_timesteps, smin * num_train_timesteps, num_inference_steps)
    s = t / num_train_timesteps
    if use_dynamic_shifting:
        s = math.exp(mu) / (math.exp(mu) + (1 / s - 1) ** 1.0)
    else:
        s = shift * s / (1 + (shift - 1) * s)
    s = torch.from_numpy(s).to(torch.float32)
    return s * num_train_timesteps, torch.cat([s, torch.zeros(1)])


def euler_step(v, x, sigma, sigma_next):
    """FlowMatchEulerDiscreteScheduler.step: fp32 update, cast to the model-output dtype."""
    return (x.to(torch.float32) + (sigma_next - sigma) * v).to(v.dtype)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """pipeline_easyanimate.py:100-112"""
    std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    return guidance_rescale * (noise_cfg * (std_text / std_cfg)) + (1 - guidance_rescale) * noise_cfg


def denoise_loop(sd: SD, cfg: dict, latents, enc_neg_pos, rope, num_steps: int, guidance_scale: float,
                 inpaint_latents=None, shift: float = 1.0, return_all: bool = False, teacache=None,
                 guidance_rescale: float = 0.0, first_step: int = 0):
    """EasyAnimatePipeline.__call__ hot loop, pipeline_easyanimate.py:1069-1111 (CFG on).  first_step > 0 runs the tail of
    the schedule only (the strength < 1 path of the inpaint pipeline, pipeline_easyanimate_inpaint.py:760-767)."""
    timesteps, sigmas = flow_sigmas(num_steps, shift=shift)
    trace = []
    for i, t in enumerate(timesteps):
        if i < first_step:
            continue
        lat_in = torch.cat([latents] * 2)
        t_expand = torch.tensor([t] * lat_in.shape[0]).to(dtype=lat_in.dtype)
        v = transformer_forward(sd, cfg, lat_in, t_expand, enc_neg_pos, rope, inpaint_latents, teacache=teacache)
        vu, vt = v.chunk(2)
        v = vu + guidance_scale * (vt - vu)
        if guidance_rescale > 0.0:                      # :1106-1108
            v = rescale_noise_cfg(v, vt, guidance_rescale)
        latents = euler_step(v, latents, sigmas[i], sigmas[i + 1])
        if return_all:
            trace.append(latents.clone())
    return (latents, trace) if return_all else latents
