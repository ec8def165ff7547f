"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.pt by executing the UNCHANGED reference modules
(/root/reference, through oracle/diffusers_shim.py) on seeded synthetic weights and inputs.

    python -m oracle.gen_golden            (run in the build container; /root/reference does not travel)

Weights are not stored: they are regenerated from (state-dict name, shape, seed, style) by
easyanimate_amd.synthetic.synth_tensor, so the fixtures stay small; the shapes recorded here are the
reference's own state-dict shapes, which also pins the key names of SURVEY Appendix C.
"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyanimate_amd.synthetic import synth_state_dict  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

TINY = dict(num_attention_heads=2, attention_head_dim=64, in_channels=16, out_channels=16, patch_size=2,
            num_layers=2, time_embed_dim=64, add_norm_text_encoder=True, text_embed_dim=48, text_embed_dim_t5=None,
            norm_eps=1e-5, time_position_encoding_type="3d_rope", enable_text_attention_mask=True)


TINY_VAE = dict(in_channels=3, out_channels=3, block_out_channels=[64, 64, 128, 128],
                down_block_types=("SpatialDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D"),
                up_block_types=("SpatialUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D"),
                mid_block_attention_type="spatial", latent_channels=16, norm_num_groups=16, spatial_group_norm=True,
                cache_mag_vae=True, slice_mag_vae=False, cache_compression_vae=False, slice_compression_vae=False,
                mini_batch_encoder=4, mini_batch_decoder=1, layers_per_block=2)


def _load_sd(module, seed, style):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(shapes, seed, style)
    module.load_state_dict(sd, strict=True)
    return shapes


def _g(seed):
    return torch.Generator().manual_seed(seed)


SECTIONS = {}


def section(name):
    def deco(fn):
        SECTIONS[name] = fn
        return fn
    return deco


@section("rope")
def gen_rope(ns, shim):
    # ---- rope tables + crop regions (pipeline_easyanimate.py:82-97, 999-1011)
    rope = {}
    for (gh, gw, f) in [(4, 4, 3), (16, 16, 1), (24, 42, 1), (6, 10, 4)]:
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((gh, gw), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (gh, gw), f, use_real=True)
        rope[f"{gh}x{gw}x{f}"] = dict(crops=cc, cos=cos, sin=sin)
    torch.save(rope, os.path.join(OUT, "rope.pt"))


@section("scheduler")
def gen_scheduler(ns, shim):
    # ---- scheduler (diffusers FlowMatchEulerDiscreteScheduler restated in the shim)
    sched = {}
    for n, shift in [(2, 1.0), (50, 1.0), (25, 3.0)]:
        s = shim.FlowMatchEulerDiscreteScheduler(shift=shift)
        s.set_timesteps(n, device="cpu", mu=1)
        sched[f"n{n}_shift{shift}"] = dict(timesteps=s.timesteps.clone(), sigmas=s.sigmas.clone())
    torch.save(sched, os.path.join(OUT, "scheduler.pt"))


@section("dit_block")
def gen_dit_block(ns, shim):
    # ---- one DiT block, stress init (attention.py:1028-1163)
    for name, mmdit in (("dit_block_mmdit", True), ("dit_block_shared", False)):
        blk = ns.attention.EasyAnimateDiTBlock(dim=128, num_attention_heads=2, attention_head_dim=64, time_embed_dim=64,
                                               norm_eps=1e-5, is_mmdit_block=mmdit).eval()
        shapes = _load_sd(blk, 11, "stress")
        g = _g(5)
        h = torch.randn(2, 40, 128, generator=g)
        e = torch.randn(2, 7, 128, generator=g)
        temb = torch.randn(2, 64, generator=g)
        cos, sin = shim.get_3d_rotary_pos_embed(64, ((0, 8), (30, 38)), (4, 5), 2, use_real=True)
        ho, eo = blk(h, e, temb, image_rotary_emb=(cos, sin))
        hb, eb = blk.to(torch.bfloat16)(h.bfloat16(), e.bfloat16(), temb.bfloat16(), image_rotary_emb=(cos, sin))
        torch.save(dict(shapes=shapes, seed=11, style="stress", h=h, e=e, temb=temb, cos=cos, sin=sin, h_out=ho, e_out=eo,
                        h_out_bf16=hb.float(), e_out_bf16=eb.float(), heads=2, norm_eps=1e-5),
                   os.path.join(OUT, f"{name}.pt"))


@section("transformer")
def gen_transformer(ns, shim):
    # ---- tiny transformers: T2V (16 ch), InP (33 ch), mixed mmdit/shared blocks
    for name, over, style in (("transformer_t2v", {}, "stress"), ("transformer_inp", dict(in_channels=33), "default"),
                              ("transformer_mixed", dict(mmdit_layers=1), "stress")):
        cfg = dict(TINY, **over)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        shapes = _load_sd(m, 3, style)
        g = _g(7)
        B, Fr, H, W, T = 2, 3, 8, 12, 9
        lat = torch.randn(B, 16, Fr, H, W, generator=g)
        inp = torch.randn(B, 17, Fr, H, W, generator=g) if cfg["in_channels"] == 33 else None
        enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
        t = torch.tensor([999.0, 999.0]).to(torch.bfloat16).float()  # the pipeline rounds t to the latent dtype
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), inpaint_latents=inp, return_dict=False)[0]
        mb = m.to(torch.bfloat16)
        outb = mb(lat.bfloat16(), t.bfloat16(), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=(cos, sin),
                  inpaint_latents=None if inp is None else inp.bfloat16(), return_dict=False)[0]
        torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style=style, latents=lat, inpaint=inp, enc=enc, t=t, cos=cos, sin=sin,
                        out=out, out_bf16=outb.float()), os.path.join(OUT, f"{name}.pt"))


@section("denoise_loop")
def gen_denoise_loop(ns, shim):
    # ---- 2-step CFG denoise loops on the tiny T2V model (pipeline_easyanimate.py:1069-1111), fp32 and bf16
    for name, style in (("denoise_loop", "stress"), ("denoise_loop_default", "default")):
        cfg = dict(TINY)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        shapes = _load_sd(m, 3, style)
        g = _g(43)
        Fr, H, W, T = 2, 8, 8, 6
        latents = torch.randn(1, 16, Fr, H, W, generator=g)
        enc = torch.randn(2, T, cfg["text_embed_dim"], generator=g)
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        traces = {}
        for dt in (torch.float32, torch.bfloat16):
            mm = m.to(dt)
            s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
            s.set_timesteps(2, device="cpu", mu=1)
            x = latents.clone().to(dt)
            trace = []
            for t in s.timesteps:
                li = torch.cat([x] * 2)
                te = torch.tensor([t] * 2).to(dtype=li.dtype)
                v = mm(li, te, encoder_hidden_states=enc.to(dt), image_rotary_emb=(cos, sin), return_dict=False)[0]
                vu, vt = v.chunk(2)
                v = vu + 6.0 * (vt - vu)
                x = s.step(v, t, x, return_dict=False)[0]
                trace.append(x.float().clone())
            traces[dt] = trace
        torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style=style, latents=latents, enc=enc, cos=cos, sin=sin,
                        guidance=6.0, steps=2, trace=traces[torch.float32], trace_bf16=traces[torch.bfloat16]),
                   os.path.join(OUT, f"{name}.pt"))


@section("teacache")
def gen_teacache(ns, shim):
    # ---- TeaCache (transformer3d.py:90-121,1564-1636): 8-step CFG loop on the tiny T2V model with the step-skip
    # heuristic on; records the reference's rel-L1 distances, skip decisions and latents (fp32 and bf16 runs)
    cfg = dict(TINY)
    m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
    shapes = _load_sd(m, 3, "stress")
    g = _g(47)
    Fr, H, W, T = 2, 8, 8, 6
    latents = torch.randn(1, 16, Fr, H, W, generator=g)
    enc = torch.randn(2, T, cfg["text_embed_dim"], generator=g)
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
    cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
    coeff = [-10.47857366, 8.33844143, -0.78477557, 0.68798618, 0.0136149]
    n_steps = 8
    tea = {}
    for thresh in (0.15, 0.3, 0.5):
        for dt in (torch.float32, torch.bfloat16):
            mm = copy.deepcopy(m).to(dt)   # (Module.to converts in place: never round-trip the fp32 weights through bf16)
            mm.enable_teacache(n_steps, thresh, coefficients=coeff)
            dists, calcs, accs = [], [], []
            orig = type(mm.teacache).compute_rel_l1_distance
            calls = {"n": 0}
            hook = mm.transformer_blocks[1].register_forward_hook(lambda *a: calls.__setitem__("n", calls["n"] + 1))

            def logged(prev, cur, _o=orig, _d=dists):
                v = _o(prev, cur)
                _d.append(v)
                return v
            mm.teacache.compute_rel_l1_distance = logged
            s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
            s.set_timesteps(n_steps, device="cpu", mu=1)
            x = latents.clone().to(dt)
            trace = []
            for t in s.timesteps:
                before = calls["n"]
                li = torch.cat([x] * 2)
                te = torch.tensor([t] * 2).to(dtype=li.dtype)
                v = mm(li, te, encoder_hidden_states=enc.to(dt), image_rotary_emb=(cos, sin), return_dict=False)[0]
                calcs.append(calls["n"] > before)
                accs.append(float(mm.teacache.accumulated_rel_l1_distance))
                vu, vt = v.chunk(2)
                v = vu + 6.0 * (vt - vu)
                x = s.step(v, t, x, return_dict=False)[0]
                trace.append(x.float().clone())
            hook.remove()
            mm.teacache = None
            tea[(thresh, "bf16" if dt == torch.bfloat16 else "fp32")] = dict(dists=dists, calcs=calcs, accs=accs, trace=trace)
            print("teacache", thresh, dt, "calc:", calcs, "dists:", [round(d, 4) for d in dists], "acc:", [round(a, 4) for a in accs])
    torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style="stress", latents=latents, enc=enc, cos=cos, sin=sin, guidance=6.0,
                    steps=n_steps, coefficients=coeff, runs=tea), os.path.join(OUT, "teacache_loop.pt"))


@section("vae_tiny")
def gen_vae_tiny(ns, shim):
    # ---- tiny MAGVIT VAE in the reference's real (chunked, cached) inference mode: encode + decode
    vkw = dict(TINY_VAE)
    vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**vkw).eval()
    shapes = _load_sd(vae, 2, "default")
    g = _g(9)
    video = torch.rand(1, 3, 9, 64, 64, generator=g) * 2 - 1
    zlat = torch.randn(1, 16, 3, 8, 8, generator=g)
    moments = vae.encode(video)[0].parameters
    dec = vae.decode(zlat)[0]
    vb = vae.to(torch.bfloat16)
    moments_b = vb.encode(video.bfloat16())[0].parameters.float()
    dec_b = vb.decode(zlat.bfloat16())[0].float()
    torch.save(dict(cfg=vkw, shapes=shapes, seed=2, style="default", video=video, z=zlat, moments=moments, dec=dec,
                    moments_bf16=moments_b, dec_bf16=dec_b), os.path.join(OUT, "vae_tiny.pt"))


@section("i2v")
def gen_i2v(ns, shim):
    # ---- I2V conditioning helpers (pipeline_easyanimate_inpaint.py:116-149, utils/utils.py:128-157)
    from easyanimate.pipeline import pipeline_easyanimate_inpaint as inp
    g = _g(21)
    mask = torch.zeros(1, 1, 9, 32, 48)
    mask[:, :, 1:] = 1.0
    lat = torch.zeros(1, 16, 3, 4, 6)
    rm = {f"first{int(b)}": inp.resize_mask(1 - mask, lat, b) for b in (True, False)}
    mask2 = (torch.rand(1, 1, 13, 16, 16, generator=g) > 0.5).float()
    rm["random_first1"] = inp.resize_mask(1 - mask2, torch.zeros(1, 16, 4, 2, 2), True)
    torch.save(dict(mask=mask, mask2=mask2, resized=rm), os.path.join(OUT, "i2v_resize_mask.pt"))


# =====================================================================================================
# round 2: stated-bar fixtures.  Large inputs are regenerated from seeds by the tests
# (inputs_from_seed below is imported by tests/ so generator and test cannot drift); large outputs are stored fp16.
# =====================================================================================================
FULL_VAE = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512],
                down_block_types=("SpatialDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D"),
                up_block_types=("SpatialUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D"),
                mid_block_attention_type="spatial", latent_channels=16, norm_num_groups=32, spatial_group_norm=True,
                cache_mag_vae=True, slice_mag_vae=False, cache_compression_vae=False, slice_compression_vae=False,
                mini_batch_encoder=4, mini_batch_decoder=1, layers_per_block=2)

FULL_DIT = dict(num_attention_heads=48, attention_head_dim=64, in_channels=16, out_channels=16, patch_size=2,
                num_layers=2, time_embed_dim=512, add_norm_text_encoder=True, text_embed_dim=3584, text_embed_dim_t5=None,
                norm_eps=1e-5, time_position_encoding_type="3d_rope", enable_text_attention_mask=True)


def vae_full_inputs(seed=9, frames=9, size=256):
    """Seeded inputs of the full-width VAE fixture (<= 9 x 256^2)."""
    g = _g(seed)
    video = torch.rand(1, 3, frames, size, size, generator=g) * 2 - 1
    z = torch.randn(1, 16, (frames - 1) // 4 + 1, size // 8, size // 8, generator=g)
    return video, z


def dit_full_inputs(cfg, seed, B, Fr, H, W, T):
    g = _g(seed)
    lat = torch.randn(B, 16, Fr, H, W, generator=g)
    extra = torch.randn(B, cfg["in_channels"] - 16, Fr, H, W, generator=g) if cfg["in_channels"] > 16 else None
    enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
    return lat, extra, enc


def _mse(a, b):
    return ((a.double() - b.double()) ** 2).mean().item()


@section("vae_full")
def gen_vae_full(ns, shim):
    # ---- full-width MAGVIT VAE (128/256/512/512, mid-block head_dim 512), 9 x 256^2, the reference's chunked/cached mode
    vkw = dict(FULL_VAE)
    vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**vkw).eval()
    shapes = _load_sd(vae, 2, "default")
    video, zlat = vae_full_inputs()
    import time
    t0 = time.time()
    moments = vae.encode(video)[0].parameters
    print("  encode fp32", time.time() - t0, flush=True)
    t0 = time.time()
    dec = vae.decode(zlat)[0]
    print("  decode fp32", time.time() - t0, flush=True)
    out = dict(cfg=vkw, shapes=shapes, seed=2, style="default", input_seed=9, frames=9, size=256,
               video_sum=video.double().sum().item(), z_sum=zlat.double().sum().item(),
               moments=moments, dec_f16=dec.to(torch.float16), dec_std=dec.std().item(), moments_std=moments.std().item())
    torch.save(out, os.path.join(OUT, "vae_full_9x256.pt"))
    if os.environ.get("EA_GOLDEN_BF16_FLOOR", "1") == "1":   # the reference's own bf16 path: scalars only (slow on CPU)
        vb = vae.to(torch.bfloat16)
        t0 = time.time()
        mb = vb.encode(video.bfloat16())[0].parameters.float()
        db = vb.decode(zlat.bfloat16())[0].float()
        print("  bf16 pass", time.time() - t0, flush=True)
        out.update(moments_floor_mse=_mse(mb, moments), dec_floor_mse=_mse(db, dec))
        torch.save(out, os.path.join(OUT, "vae_full_9x256.pt"))
    print("  moments std", out["moments_std"], "dec std", out["dec_std"], {k: v for k, v in out.items() if k.endswith("floor_mse")})


@section("vae_dec_512")
def gen_vae_dec_512(ns, shim):
    # ---- full-width decoder at 5 x 512^2: rows of 512 voxels, the shape class where the C_out = 128 layers run on the
    # 512-voxel row-slab kernel (the kernel that carries 39 % of a 49 x 1024^2 decode).  Output stored at every second
    # pixel (fp16): the comparison is an MSE, and the fixture stays at 2 MB.
    vkw = dict(FULL_VAE)
    vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**vkw).eval()
    shapes = _load_sd(vae, 2, "default")
    _, zlat = vae_full_inputs(seed=10, frames=5, size=512)
    import time
    t0 = time.time()
    dec = vae.decode(zlat)[0]
    print("  decode fp32 5x512^2", time.time() - t0, flush=True)
    out = dict(cfg=vkw, shapes=shapes, seed=2, style="default", input_seed=10, frames=5, size=512, z_sum=zlat.double().sum().item(),
               dec_sub_f16=dec[..., ::2, ::2].to(torch.float16).contiguous(), dec_std=dec.std().item(), dec_shape=tuple(dec.shape))
    torch.save(out, os.path.join(OUT, "vae_dec_5x512.pt"))
    print("  dec std", out["dec_std"], "shape", out["dec_shape"])


def vae_ragged_inputs(seed=12, frames=5, height=96, width=168):
    g = _g(seed)
    video = torch.rand(1, 3, frames, height, width, generator=g) * 2 - 1
    z = torch.randn(1, 16, (frames - 1) // 4 + 1, height // 8, width // 8, generator=g)
    return video, z


@section("vae_ragged")
def gen_vae_ragged(ns, shim):
    # ---- round 3: full-width VAE at a quarter of the reference's published 384 x 672 shape: 5 x 96 x 168, latents 12 x 21 --
    # an odd latent width, rows that are no multiple of any tile (21 / 42 / 84 / 168 voxels), 252 mid-block keys (padded to
    # 256 by the product).  The reference in its chunked / cached mode, encode and decode, plus its own bf16 floor.
    vkw = dict(FULL_VAE)
    vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**vkw).eval()
    shapes = _load_sd(vae, 2, "default")
    video, zlat = vae_ragged_inputs()
    moments = vae.encode(video)[0].parameters
    dec = vae.decode(zlat)[0]
    out = dict(cfg=vkw, shapes=shapes, seed=2, style="default", input_seed=12, frames=5, height=96, width=168,
               video_sum=video.double().sum().item(), z_sum=zlat.double().sum().item(),
               moments=moments, dec_f16=dec.to(torch.float16), dec_std=dec.std().item(), moments_std=moments.std().item())
    vb = vae.to(torch.bfloat16)
    mb = vb.encode(video.bfloat16())[0].parameters.float()
    db = vb.decode(zlat.bfloat16())[0].float()
    out.update(moments_floor_mse=_mse(mb, moments), dec_floor_mse=_mse(db, dec))
    torch.save(out, os.path.join(OUT, "vae_full_ragged_5x96x168.pt"))
    print("  moments", tuple(moments.shape), "std", out["moments_std"], "dec", tuple(dec.shape), "std", out["dec_std"],
          {k: v for k, v in out.items() if k.endswith("floor_mse")})


@section("vae_dec_wide")
def gen_vae_dec_wide(ns, shim):
    # ---- round 4 (VERDICT r3 next #3b): full-width decoder at 5 x 256 x 1024 -- output rows of 1024 voxels = TWO 512-voxel tiles
    # per row, the geometry of every full-resolution layer of a 49 x 1024^2 decode; source rows of 256 / 512 voxels for the two
    # large up-samplers (the sub-pixel kernels) and 3 -> 5 frames behind a virtual x2 (the merged-tap layers).  The reference in
    # its chunked / cached mode, fp32; every second pixel stored (fp16), plus the reference's own bf16 floor.
    vkw = dict(FULL_VAE)
    vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**vkw).eval()
    shapes = _load_sd(vae, 2, "default")
    _, zlat = vae_ragged_inputs(seed=14, frames=5, height=256, width=1024)
    import time
    t0 = time.time()
    with torch.no_grad():
        dec = vae.decode(zlat)[0]
    print("  decode fp32 5x256x1024", time.time() - t0, flush=True)
    out = dict(cfg=vkw, shapes=shapes, seed=2, style="default", input_seed=14, frames=5, height=256, width=1024,
               z_sum=zlat.double().sum().item(), dec_sub_f16=dec[..., ::2, ::2].to(torch.float16).contiguous(), dec_std=dec.std().item(),
               dec_shape=tuple(dec.shape))
    torch.save(out, os.path.join(OUT, "vae_dec_5x256x1024.pt"))
    t0 = time.time()
    with torch.no_grad():
        db = vae.to(torch.bfloat16).decode(zlat.bfloat16())[0].float()
    out["dec_floor_mse"] = _mse(db, dec)
    torch.save(out, os.path.join(OUT, "vae_dec_5x256x1024.pt"))
    print("  bf16 pass", time.time() - t0, "dec std", out["dec_std"], "shape", out["dec_shape"], "floor", out["dec_floor_mse"], flush=True)



def vae_config4_inputs(seed=15, frames=49, size=1024):
    """Seeded inputs of BASELINE config 4: U(-1, 1) RGB [1, 3, 49, 1024, 1024] and the latents the pipeline hands to
    vae.decode, randn[1, 16, 13, 128, 128] / scaling_factor (pipeline_easyanimate.py decode_latents)."""
    g = _g(seed)
    z = torch.randn(1, 16, (frames - 1) // 4 + 1, size // 8, size // 8, generator=g) / 0.1825
    video = torch.rand(1, 3, frames, size, size, generator=g) * 2 - 1
    return video, z


C4_PIX_STRIDE = 7      # coprime with every tile / sub-pixel period of the decoder: the samples visit all phases of 2, 4, 8, 16, 512
C4_LAT_STRIDE = 3


@section("vae_config4")
def gen_vae_config4(ns, shim):
    # ---- round 5 (VERDICT r4 next #2a): BASELINE config 4 AT THE SHAPE IT IS QUOTED ON -- 49 x 1024^2 through the unchanged
    # reference (fp32, its chunked / cached mode: omnigen_enc_dec.py:279-337, 617-677; autoencoder_magvit.py:229-317), decode
    # and encode.  ~1 h on 8 cores.  Stored: every 7th pixel (fp16) of the decode, every 3rd latent site (fp32) of the moments.
    # EA_GOLDEN_C4_PART = "dec" | "enc" restricts the run to one half (each half writes its own file).
    import time
    part = os.environ.get("EA_GOLDEN_C4_PART", "dec,enc").split(",")
    vkw = dict(FULL_VAE)
    vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**vkw).eval()
    shapes = _load_sd(vae, 2, "default")
    video, zlat = vae_config4_inputs()
    base = dict(cfg=vkw, shapes=shapes, seed=2, style="default", input_seed=15, frames=49, size=1024)
    if "dec" in part:
        t0 = time.time()
        with torch.no_grad():
            dec = vae.decode(zlat)[0]
        dt = time.time() - t0
        print("  decode fp32 49x1024^2", dt, tuple(dec.shape), flush=True)
        s = C4_PIX_STRIDE
        out = dict(base, z_sum=zlat.double().sum().item(), dec_shape=tuple(dec.shape), dec_std=dec.std().item(), stride=s,
                   dec_sub_f16=dec[..., ::s, ::s].to(torch.float16).contiguous(), ref_cpu_seconds=dt,
                   dec_frame_means=dec.double().mean(dim=(0, 1, 3, 4)))
        torch.save(out, os.path.join(OUT, "vae_config4_dec_49x1024.pt"))
        print("  dec std", out["dec_std"], flush=True)
        del dec
    if "enc" in part:
        t0 = time.time()
        with torch.no_grad():
            moments = vae.encode(video)[0].parameters
        dt = time.time() - t0
        print("  encode fp32 49x1024^2", dt, tuple(moments.shape), flush=True)
        s = C4_LAT_STRIDE
        out = dict(base, video_sum=video.double().sum().item(), moments_shape=tuple(moments.shape), moments_std=moments.std().item(),
                   stride=s, moments_sub=moments[..., ::s, ::s].contiguous(), ref_cpu_seconds=dt,
                   moments_frame_means=moments.double().mean(dim=(0, 1, 3, 4)))
        torch.save(out, os.path.join(OUT, "vae_config4_enc_49x1024.pt"))
        print("  moments std", out["moments_std"], flush=True)


def _ref_loop(ns, shim, m, latents, enc, rope, steps, guidance, dt, keep=None):
    """The reference's sampling loop (pipeline_easyanimate.py:1069-1111) over the shim-hosted reference transformer."""
    mm = copy.deepcopy(m).to(dt)
    s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
    s.set_timesteps(steps, device="cpu", mu=1)
    x = latents.clone().to(dt)
    kept = {}
    for i, t in enumerate(s.timesteps):
        li = torch.cat([x] * 2)
        te = torch.tensor([t] * 2).to(dtype=li.dtype)
        v = mm(li, te, encoder_hidden_states=enc.to(dt), image_rotary_emb=rope, return_dict=False)[0]
        vu, vt = v.chunk(2)
        v = vu + guidance * (vt - vu)
        x = s.step(v, t, x, return_dict=False)[0]
        if keep and (i + 1) in keep:
            kept[i + 1] = x.float().clone()
    return x.float(), kept


@section("denoise_loop_50")
def gen_denoise_loop_50(ns, shim):
    # ---- the stated bar at the stated schedule: 50 Flow steps, CFG 6, tiny T2V model; reference fp32 and bf16 traces
    # "_bf16" styles: weights, initial latents and text embeddings are bf16-representable values, i.e. the fp32 reference
    # and a bf16 implementation start from bit-identical inputs (what a released bf16 checkpoint gives); without it the
    # fp32 trace uses weights the bf16 model cannot hold and the comparison also measures the checkpoint's rounding.
    for name, style in (("denoise_loop_50", "stress"), ("denoise_loop_50_default", "default"),
                        ("denoise_loop_50_bf16in", "stress_bf16"), ("denoise_loop_50_default_bf16in", "default_bf16")):
        cfg = dict(TINY)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        shapes = _load_sd(m, 3, style)
        g = _g(43)
        Fr, H, W, T = 2, 8, 8, 6
        latents = torch.randn(1, 16, Fr, H, W, generator=g)
        enc = torch.randn(2, T, cfg["text_embed_dim"], generator=g)
        if style.endswith("_bf16"):
            latents, enc = latents.bfloat16().float(), enc.bfloat16().float()
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        rope = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        keep = (1, 10, 25, 50)
        xf, kf = _ref_loop(ns, shim, m, latents, enc, rope, 50, 6.0, torch.float32, keep)
        xb, kb = _ref_loop(ns, shim, m, latents, enc, rope, 50, 6.0, torch.bfloat16, keep)
        print(f"  {name}: final latent std {xf.std().item():.3f}; reference bf16-vs-fp32 MSE by step",
              {k: _mse(kb[k], kf[k]) for k in keep})
        torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style=style, latents=latents, enc=enc, cos=rope[0], sin=rope[1],
                        guidance=6.0, steps=50, trace=kf, trace_bf16=kb), os.path.join(OUT, f"{name}.pt"))


@section("transformer_r2")
def gen_transformer_r2(ns, shim):
    # ---- branches of the same forward that round 1 left untested: the V5 two-encoder text path (text_proj_t5,
    # transformer3d.py:1533-1536; add_norm_text_encoder False as config/easyanimate_video_v5_magvit_multi_text_encoder.yaml
    # and True), control_latents alone and together with inpaint_latents (:1523-1526)
    cases = (("transformer_t5", dict(add_norm_text_encoder=False, text_embed_dim=48, text_embed_dim_t5=40), 0, "stress"),
             ("transformer_t5_norm", dict(add_norm_text_encoder=True, text_embed_dim=48, text_embed_dim_t5=48), 0, "stress"),
             ("transformer_control", dict(in_channels=32), 16, "stress"),
             ("transformer_inp_control", dict(in_channels=49), 16, "default"))
    for name, over, n_ctrl, style in cases:
        cfg = dict(TINY, **over)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        shapes = _load_sd(m, 3, style)
        g = _g(17)
        B, Fr, H, W, T, T5 = 2, 3, 8, 12, 9, 5
        lat = torch.randn(B, 16, Fr, H, W, generator=g)
        n_inp = cfg["in_channels"] - 16 - n_ctrl
        inp = torch.randn(B, n_inp, Fr, H, W, generator=g) if n_inp else None
        ctrl = torch.randn(B, n_ctrl, Fr, H, W, generator=g) if n_ctrl else None
        enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
        enc5 = torch.randn(B, T5, cfg["text_embed_dim_t5"], generator=g) * 3 if cfg["text_embed_dim_t5"] else None
        t = torch.tensor([603.0, 603.0]).to(torch.bfloat16).float()
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        bf = lambda x: None if x is None else x.bfloat16()
        out = m(lat, t, encoder_hidden_states=enc, encoder_hidden_states_t5=enc5, image_rotary_emb=(cos, sin),
                inpaint_latents=inp, control_latents=ctrl, return_dict=False)[0]
        mb = copy.deepcopy(m).to(torch.bfloat16)
        outb = mb(bf(lat), t.bfloat16(), encoder_hidden_states=bf(enc), encoder_hidden_states_t5=bf(enc5),
                  image_rotary_emb=(cos, sin), inpaint_latents=bf(inp), control_latents=bf(ctrl), return_dict=False)[0]
        print(f"  {name}: out std {out.std().item():.3f}, floor {_mse(outb.float(), out):.3e}")
        torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style=style, latents=lat, inpaint=inp, control=ctrl, enc=enc, enc_t5=enc5,
                        t=t, cos=cos, sin=sin, out=out, out_bf16=outb.float()), os.path.join(OUT, f"{name}.pt"))


@section("transformer_full")
def gen_transformer_full(ns, shim):
    # ---- full-width (d = 3072 = 48 x 64, ff 12288, text 3584, time 512) 2-layer forwards, SURVEY 8d-(i):
    # T2V 16 channels at 5 x 64 x 64 latents (N = 5120 video + 256 text tokens) and InP 33 channels at 3 x 32 x 48
    # (N = 1152, T = 77: unaligned text).  Inputs are regenerated from seeds by the tests (dit_full_inputs).
    import time
    for name, over, dims, style, floor in (("transformer_full_t2v", {}, (2, 5, 64, 64, 256), "stress", False),
                                           ("transformer_full_inp", dict(in_channels=33), (2, 3, 32, 48, 77), "stress", True)):
        cfg = dict(FULL_DIT, **over)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        shapes = _load_sd(m, 5, style)
        B, Fr, H, W, T = dims
        lat, extra, enc = dit_full_inputs(cfg, 23, *dims)
        t = torch.tensor([799.0] * B).to(torch.bfloat16).float()
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        t0 = time.time()
        out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), inpaint_latents=extra, return_dict=False)[0]
        print(f"  {name}: fp32 forward {time.time() - t0:.1f} s, out std {out.std().item():.3f}", flush=True)
        rec = dict(cfg=cfg, shapes=shapes, seed=5, style=style, input_seed=23, dims=dims, t=t, crops=cc,
                   lat_sum=lat.double().sum().item(), enc_sum=enc.double().sum().item(), out=out)
        if floor:
            t0 = time.time()
            mb = m.to(torch.bfloat16)
            outb = mb(lat.bfloat16(), t.bfloat16(), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=(cos, sin),
                      inpaint_latents=None if extra is None else extra.bfloat16(), return_dict=False)[0]
            rec["floor_mse"] = _mse(outb.float(), out)
            print(f"  {name}: bf16 forward {time.time() - t0:.1f} s, floor {rec['floor_mse']:.3e}", flush=True)
        torch.save(rec, os.path.join(OUT, f"{name}.pt"))


@section("transformer_full_ragged")
def gen_transformer_full_ragged(ns, shim):
    # ---- round 3: full width at the patch grid of the reference's own published shapes (README.md:143: 384 x 672 -> latents
    # 48 x 84 -> 24 x 42 patches), two latent frames: N = 2016 video tokens = 31.5 x 64 -- not a multiple of the 64-key tile, the
    # 256-row query block or the GEMM tiles -- with T = 256 text tokens.  Inputs regenerated from seeds (dit_full_inputs).
    import time
    cfg = dict(FULL_DIT)
    m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
    shapes = _load_sd(m, 5, "stress")
    dims = (2, 2, 48, 84, 256)
    B, Fr, H, W, T = dims
    lat, extra, enc = dit_full_inputs(cfg, 29, *dims)
    t = torch.tensor([433.0] * B).to(torch.bfloat16).float()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
    cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
    t0 = time.time()
    out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), inpaint_latents=extra, return_dict=False)[0]
    print(f"  transformer_full_ragged: fp32 forward {time.time() - t0:.1f} s, out std {out.std().item():.3f}", flush=True)
    torch.save(dict(cfg=cfg, shapes=shapes, seed=5, style="stress", input_seed=29, dims=dims, t=t, crops=cc,
                    lat_sum=lat.double().sum().item(), enc_sum=enc.double().sum().item(), out=out),
               os.path.join(OUT, "transformer_full_ragged.pt"))


DIT_7B = dict(FULL_DIT, num_layers=28)     # "7B-class" declared dims, SURVEY Appendix B


def config1_inputs():
    """1 frame 256 x 256 -> latents [1,16,1,32,32]; text embeddings randn[1,256,3584] for the
    negative and the positive prompt (generator seed 1); latents from torch.Generator("cpu").manual_seed(43).
    Values are rounded to bf16 so that the fp32 reference and the bf16 product start from identical inputs."""
    enc = torch.randn(2, 256, 3584, generator=_g(1)).bfloat16().float()
    latents = torch.randn(1, 16, 1, 32, 32, generator=_g(43)).bfloat16().float()
    return latents, enc


def _meta_build(ctor, seed, style):
    """Instantiate a (large) reference module without allocating / initialising it twice: parameters are created on the
    meta device and replaced by the synthetic tensors (load_state_dict(assign=True))."""
    with torch.device("meta"):
        m = ctor()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth_state_dict(shapes, seed, style), strict=True, assign=True)
    for n, b in list(m.named_buffers()):
        assert not b.is_meta, n
    return m.eval(), shapes


@section("config1")
def gen_config1(ns, shim):
    # ---- BASELINE.json configs[0] at its declared dims (SURVEY 8d "Config 1"): v5.1 yaml, 7B-class DiT (L=28, d=3072),
    # 1 frame 256x256, 2 Flow steps, CFG 6, fp32, then the full-width VAE decode of the one frame.  Reference modules on CPU.
    import time
    t0 = time.time()
    m, shapes = _meta_build(lambda: ns.transformer3d.EasyAnimateTransformer3DModel(**DIT_7B), 0, "default_bf16")
    print(f"  7B-class reference transformer built in {time.time() - t0:.0f} s, {sum(p.numel() for p in m.parameters()) / 1e9:.2f} B parameters", flush=True)
    latents, enc = config1_inputs()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((16, 16), 45, 30)
    rope = shim.get_3d_rotary_pos_embed(64, cc, (16, 16), 1, use_real=True)
    s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
    s.set_timesteps(2, device="cpu", mu=1)
    x = latents.clone()
    trace = []
    for t in s.timesteps:
        t0 = time.time()
        li = torch.cat([x] * 2)
        v = m(li, torch.tensor([t] * 2).to(li.dtype), encoder_hidden_states=enc, image_rotary_emb=rope, return_dict=False)[0]
        vu, vt = v.chunk(2)
        x = s.step(vu + 6.0 * (vt - vu), t, x, return_dict=False)[0]
        trace.append(x.clone())
        print(f"  step: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads", flush=True)
    # the reference's own bf16 run of the same loop (the noise floor of this 2-step, d_sigma = 0.5, CFG-6 configuration)
    mb = m.to(torch.bfloat16)
    s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
    s.set_timesteps(2, device="cpu", mu=1)
    xb = latents.clone().bfloat16()
    trace_b = []
    for t in s.timesteps:
        t0 = time.time()
        li = torch.cat([xb] * 2)
        v = mb(li, torch.tensor([t] * 2).to(li.dtype), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=rope, return_dict=False)[0]
        vu, vt = v.chunk(2)
        xb = s.step(vu + 6.0 * (vt - vu), t, xb, return_dict=False)[0]
        trace_b.append(xb.float().clone())
        print(f"  bf16 step: {time.time() - t0:.1f} s; reference bf16-vs-fp32 latent MSE {_mse(trace_b[-1], trace[len(trace_b) - 1]):.3e}", flush=True)
    del m, mb
    vae, vshapes = _meta_build(lambda: ns.autoencoder_magvit.AutoencoderKLMagvit(**FULL_VAE), 2, "default_bf16")
    t0 = time.time()
    dec = vae.decode(x / 0.1825)[0]
    frames = (dec.clamp(-1, 1) / 2 + 0.5).clamp(0, 1)    # pipeline_easyanimate.py:731-739
    print(f"  vae decode {time.time() - t0:.1f} s; latents std {x.std().item():.3f}, frames mean {frames.mean().item():.3f}", flush=True)
    torch.save(dict(dit_cfg=DIT_7B, vae_cfg=FULL_VAE, dit_seed=0, vae_seed=2, style="default_bf16", steps=2, guidance=6.0,
                    height=256, width=256, video_length=1, latents_sum=latents.double().sum().item(),
                    enc_sum=enc.double().sum().item(), trace=trace, trace_bf16=trace_b, frames=frames), os.path.join(OUT, "config1_7b_256.pt"))


@section("transformer_r2b")
def gen_transformer_r2b(ns, shim):
    # ---- more branches of the hot forward: after_norm (attention.py:1102-1105,1150-1155) and the
    # ref-latent / CLIP conditioning of the control checkpoints (transformer3d.py:1420-1431,1538-1561)
    cases = (("transformer_after_norm", dict(after_norm=True), False, "stress"),
             ("transformer_ref_clip", dict(in_channels=32, ref_channels=16, clip_channels=24, sample_height=12, sample_width=20), True, "stress"),
             ("transformer_ref", dict(in_channels=32, ref_channels=16, sample_height=8, sample_width=12), True, "default"))
    for name, over, use_ref, style in cases:
        cfg = dict(TINY, **over)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval().to(torch.float32)   # (the sin/cos buffer is float64 until cast)
        shapes = _load_sd(m, 3, style)
        g = _g(19)
        B, Fr, H, W, T = 2, 3, 8, 12, 9
        lat = torch.randn(B, 16, Fr, H, W, generator=g)
        ctrl = torch.randn(B, cfg["in_channels"] - 16, Fr, H, W, generator=g) if cfg["in_channels"] > 16 else None
        enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
        ref = torch.randn(B, 16, 1, H, W, generator=g) if use_ref else None
        clip = torch.randn(B, 5, cfg["clip_channels"], generator=g) if cfg.get("clip_channels") else None
        t = torch.tensor([411.0, 411.0]).to(torch.bfloat16).float()
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        bf = lambda x: None if x is None else x.bfloat16()
        out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), control_latents=ctrl, ref_latents=ref,
                clip_encoder_hidden_states=clip, return_dict=False)[0]
        mb = copy.deepcopy(m).to(torch.bfloat16)
        outb = mb(bf(lat), t.bfloat16(), encoder_hidden_states=bf(enc), image_rotary_emb=(cos, sin), control_latents=bf(ctrl),
                  ref_latents=bf(ref), clip_encoder_hidden_states=bf(clip), return_dict=False)[0]
        print(f"  {name}: out std {out.std().item():.3f}, floor {_mse(outb.float(), out):.3e}")
        torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style=style, latents=lat, control=ctrl, enc=enc, ref=ref, clip=clip, t=t,
                        cos=cos, sin=sin, out=out, out_bf16=outb.float()), os.path.join(OUT, f"{name}.pt"))


@section("swa")
def gen_swa(ns, shim):
    # ---- sliding-window attention blocks (processor.py:320-459, attention.py:1045-1046,1065,1124-1132):
    # the reference's own EasyAnimateSWAttnProcessor2_0 with flash_attn_func restated (oracle/flash_attn_shim.py).
    # 3 x 16 x 48 patches = 2304 video tokens > 2 * (1024 - T): the strided cross keys use interval 2; window = 768.
    from oracle import flash_attn_shim
    ns.processor.flash_attn_func = flash_attn_shim.flash_attn_func
    for name, over, style in (("transformer_swa", dict(num_attention_heads=6, num_layers=2, swa_layers=[0, 1]), "stress"),
                              ("transformer_swa_mixed", dict(num_attention_heads=6, num_layers=3, swa_layers=[1], mmdit_layers=2), "default")):
        cfg = dict(TINY, **over)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        assert sum(type(b.attn1.processor).__name__ == "EasyAnimateSWAttnProcessor2_0" for b in m.transformer_blocks) == len(cfg["swa_layers"])
        shapes = _load_sd(m, 3, style)
        g = _g(31)
        B, Fr, H, W, T = 2, 3, 32, 96, 24
        lat = torch.randn(B, 16, Fr, H, W, generator=g)
        enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
        t = torch.tensor([707.0, 707.0]).to(torch.bfloat16).float()
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), return_dict=False)[0]
        mb = copy.deepcopy(m).to(torch.bfloat16)
        outb = mb(lat.bfloat16(), t.bfloat16(), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=(cos, sin), return_dict=False)[0]
        print(f"  {name}: out std {out.std().item():.3f}, floor {_mse(outb.float(), out):.3e}")
        torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style=style, input_seed=31, dims=(B, Fr, H, W, T), t=t, crops=cc,
                        lat_sum=lat.double().sum().item(), out=out.to(torch.float16), out_std=out.std().item(),
                        floor_mse=_mse(outb.float(), out)), os.path.join(OUT, f"{name}.pt"))


def swa_inputs(cfg, seed, B, Fr, H, W, T):
    g = _g(seed)
    lat = torch.randn(B, 16, Fr, H, W, generator=g)
    enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
    return lat, enc


@section("fp8")
def gen_fp8(ns, shim):
    # ---- fp8 weight storage, the reference's default GPU_memory_mode (predict_t2v.py:37,104,266; utils/fp8_optimization.py:
    # 17-35): every parameter stored as float8_e4m3fn and up-cast to the compute dtype per call.  The reference's own
    # wrappers run here on the CPU: bf16 compute through convert_weight_dtype_wrapper, and (the exact oracle) fp32 compute
    # on the same fp8-rounded values.
    from easyanimate.utils.fp8_optimization import convert_model_weight_to_float8, convert_weight_dtype_wrapper
    cfg = dict(TINY)
    m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
    shapes = _load_sd(m, 3, "stress")
    g = _g(37)
    B, Fr, H, W, T = 2, 3, 8, 12, 9
    lat = torch.randn(B, 16, Fr, H, W, generator=g)
    enc = torch.randn(B, T, cfg["text_embed_dim"], generator=g) * 3
    t = torch.tensor([555.0, 555.0]).to(torch.bfloat16).float()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
    cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
    m8 = copy.deepcopy(m).to(torch.bfloat16)            # predict_t2v.py loads with torch_dtype=float8_e4m3fn from a bf16 checkpoint
    convert_model_weight_to_float8(m8)
    assert all(p.dtype == torch.float8_e4m3fn for p in m8.parameters())
    mq = copy.deepcopy(m)                               # fp32 compute on the fp8-representable values
    mq.load_state_dict({k: v.float() for k, v in m8.state_dict().items()})
    convert_weight_dtype_wrapper(m8, torch.bfloat16)
    out8 = m8(lat.bfloat16(), t.bfloat16(), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=(cos, sin), return_dict=False)[0]
    outq = mq(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), return_dict=False)[0]
    out0 = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), return_dict=False)[0]
    print(f"  fp8 storage: reference bf16-through-wrappers vs fp32-on-fp8-values {_mse(out8.float(), outq):.3e}; the fp8 rounding itself "
          f"moves the fp32 output by {_mse(outq, out0):.3e}")
    torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style="stress", latents=lat, enc=enc, t=t, cos=cos, sin=sin, out=outq,
                    out_bf16=out8.float(), out_unquantised=out0), os.path.join(OUT, "transformer_fp8_storage.pt"))


@section("config1_v0")
def gen_config1_v0(ns, shim):
    # ---- VERDICT r2 next #6a: the quantity that explains config 1's latent MSE -- the velocity of the FIRST forward (CFG
    # pair, un-combined) of the declared 7B-class model at 1 x 256^2: reference fp32, and the reference's own bf16 forward
    # from the same inputs (its per-forward noise).  Kept in its own small file (config1_7b_256.pt is unchanged).
    import time
    m, _ = _meta_build(lambda: ns.transformer3d.EasyAnimateTransformer3DModel(**DIT_7B), 0, "default_bf16")
    latents, enc = config1_inputs()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((16, 16), 45, 30)
    rope = shim.get_3d_rotary_pos_embed(64, cc, (16, 16), 1, use_real=True)
    s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
    s.set_timesteps(2, device="cpu", mu=1)
    t = s.timesteps[0]
    li = torch.cat([latents] * 2)
    t0 = time.time()
    v = m(li, torch.tensor([t] * 2).to(li.dtype), encoder_hidden_states=enc, image_rotary_emb=rope, return_dict=False)[0]
    print(f"  fp32 forward {time.time() - t0:.1f} s; velocity std {v.std().item():.3f}", flush=True)
    mb = m.to(torch.bfloat16)
    t0 = time.time()
    vb = mb(li.bfloat16(), torch.tensor([t] * 2).to(torch.bfloat16), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=rope,
            return_dict=False)[0].float()
    cfg_f, cfg_b = v[0] + 6.0 * (v[1] - v[0]), vb[0] + 6.0 * (vb[1] - vb[0])
    print(f"  bf16 forward {time.time() - t0:.1f} s; reference bf16-vs-fp32: per-forward velocity MSE {_mse(vb, v):.3e}, after the CFG-6 "
          f"combine {_mse(cfg_b, cfg_f):.3e}", flush=True)
    torch.save(dict(timestep=float(t), v=v, v_bf16=vb, floor_mse=_mse(vb, v), floor_cfg_mse=_mse(cfg_b, cfg_f)),
               os.path.join(OUT, "config1_7b_256_v0.pt"))


def config2_inputs():
    """BASELINE.json configs[1]: 7B-class DiT, 49 frames 512 x 512 -> latents [1,16,13,64,64] (13 * 32 * 32 = 13 312 video tokens
    + 256 text tokens = S 13 568); text embeddings for the negative / positive prompt.  bf16-representable values, so the fp32
    reference and the bf16 product start from identical inputs."""
    enc = (torch.randn(2, 256, 3584, generator=_g(61)) * 3).bfloat16().float()
    latents = torch.randn(1, 16, 13, 64, 64, generator=_g(62)).bfloat16().float()
    return latents, enc


@section("config2_forward")
def gen_config2_forward(ns, shim):
    # ---- VERDICT r3 next #3a: depth x length on a BASELINE config -- ONE forward of the declared 7B-class model (L = 28,
    # d = 3072) at config 2's full shape (S = 13 568), CFG pair, the first timestep of the 50-step Flow schedule; the unchanged
    # reference in fp32 on the host cores (~3e14 FLOP).  The two batch elements run one after the other (B = 1 each: the forward
    # has no cross-batch term, and a B = 2 fp32 pass of this size does not fit beside the 25 GB of fp32 weights in 62 GB of RAM).
    # Then the reference's own bf16 forward from the same inputs (its per-forward noise floor).  Stored as fp16 (|v| < 10).
    import time
    t0 = time.time()
    m, _ = _meta_build(lambda: ns.transformer3d.EasyAnimateTransformer3DModel(**DIT_7B), 0, "default_bf16")
    print(f"  7B-class reference transformer built in {time.time() - t0:.0f} s", flush=True)
    latents, enc = config2_inputs()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((32, 32), 45, 30)
    rope = shim.get_3d_rotary_pos_embed(64, cc, (32, 32), 13, use_real=True)
    s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
    s.set_timesteps(50, device="cpu", mu=1)
    t = s.timesteps[0]
    tt = torch.tensor([t]).to(torch.bfloat16).float()     # the pipeline hands the model a bf16 timestep (pipeline_easyanimate.py:1079-1081)
    outs = []
    with torch.no_grad():
        for b in range(2):
            t0 = time.time()
            outs.append(m(latents, tt, encoder_hidden_states=enc[b:b + 1], image_rotary_emb=rope, return_dict=False)[0])
            print(f"  fp32 forward, batch element {b}: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads; std {outs[-1].std().item():.3f}", flush=True)
        v = torch.cat(outs)
        mb = m.to(torch.bfloat16)
        outs = []
        for b in range(2):
            t0 = time.time()
            outs.append(mb(latents.bfloat16(), tt.bfloat16(), encoder_hidden_states=enc[b:b + 1].bfloat16(), image_rotary_emb=rope,
                           return_dict=False)[0].float())
            print(f"  bf16 forward, batch element {b}: {time.time() - t0:.1f} s", flush=True)
        vb = torch.cat(outs)
    print(f"  reference bf16-vs-fp32 velocity MSE {_mse(vb, v):.3e}; |v| max {v.abs().max().item():.2f}", flush=True)
    torch.save(dict(cfg=DIT_7B, seed=0, style="default_bf16", timestep=float(tt), crops=cc, v=v.half(), floor_mse=_mse(vb, v),
                    rel_l2_floor=((vb - v).double().norm() / v.double().norm()).item(),
                    fp16_storage_mse=_mse(v.half().float(), v), latents_sum=latents.double().sum().item(),
                    enc_sum=enc.double().sum().item()), os.path.join(OUT, "config2_7b_49x512_v0.pt"))


@section("transformer_full_length")
def gen_transformer_full_length(ns, shim):
    # ---- round 4: config 3's REAL sequence length through the unchanged reference -- full width (d = 3072, 48 heads), 2 layers,
    # latents [1,16,13,128,128] = 53 248 video + 256 text tokens (S = 53 504: 49 f x 1024^2), one batch element, fp32 on the host
    # cores (~1e14 FLOP; the 12B model's 48 layers at this length are hours of CPU time).  This is the one place the attention
    # kernel's 836-tile online softmax, the 208-tile GEMM grids and the RoPE tables of the benchmark shape meet a reference
    # output.  Stored at every second latent pixel (fp16).  Inputs regenerated from seeds by the test (dit_full_inputs).
    import time
    cfg = dict(FULL_DIT)
    m, shapes = _meta_build(lambda: ns.transformer3d.EasyAnimateTransformer3DModel(**cfg), 5, "stress")
    dims = (1, 13, 128, 128, 256)
    B, Fr, H, W, T = dims
    lat, extra, enc = dit_full_inputs(cfg, 29, *dims)
    t = torch.tensor([799.0] * B).to(torch.bfloat16).float()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
    cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
    t0 = time.time()
    with torch.no_grad():
        out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), return_dict=False)[0]
    print(f"  transformer_full_length: fp32 forward {time.time() - t0:.1f} s, out std {out.std().item():.3f}, |out| max {out.abs().max().item():.2f}", flush=True)
    torch.save(dict(cfg=cfg, shapes=shapes, seed=5, style="stress", input_seed=29, dims=dims, t=t, crops=cc,
                    lat_sum=lat.double().sum().item(), enc_sum=enc.double().sum().item(), out_sub_f16=out[..., ::2, ::2].half().contiguous(),
                    out_std=out.std().item(), out_shape=tuple(out.shape)), os.path.join(OUT, "transformer_full_length_s53504.pt"))


FULL_LOOP_DIMS = (3, 40, 56, 256)      # latent frames, latent H, W, text tokens -> 3 * 20 * 28 = 1680 video tokens


@section("loop_full_width")
def gen_loop_full_width(ns, shim):
    # ---- VERDICT r2 next #6b: a multi-step loop at FULL WIDTH (d = 3072, 48 heads, ff 12288; 2 layers), 1680 video + 256 text
    # tokens, 10 Flow steps, CFG 6: the reference's loop in fp32 and in its own bf16 bookkeeping.
    import time
    for name, style in (("loop_full_width", "stress_bf16"), ("loop_full_width_default", "default_bf16")):
        cfg = dict(FULL_DIT)
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
        shapes = _load_sd(m, 11, style)
        Fr, H, W, T = FULL_LOOP_DIMS
        g = _g(47)
        latents = torch.randn(1, 16, Fr, H, W, generator=g).bfloat16().float()
        enc = torch.randn(2, T, cfg["text_embed_dim"], generator=g).bfloat16().float()
        cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
        rope = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
        keep = (1, 5, 10)
        t0 = time.time()
        xf, kf = _ref_loop(ns, shim, m, latents, enc, rope, 10, 6.0, torch.float32, keep)
        t1 = time.time()
        xb, kb = _ref_loop(ns, shim, m, latents, enc, rope, 10, 6.0, torch.bfloat16, keep)
        print(f"  {name}: fp32 loop {t1 - t0:.0f} s, bf16 loop {time.time() - t1:.0f} s; final latent std {xf.std().item():.3f}; reference "
              f"bf16-vs-fp32 MSE by step", {k: _mse(kb[k], kf[k]) for k in keep}, flush=True)
        torch.save(dict(cfg=cfg, shapes=shapes, seed=11, style=style, input_seed=47, dims=FULL_LOOP_DIMS, crops=cc, guidance=6.0, steps=10,
                        latents=latents, enc_sum=enc.double().sum().item(), trace=kf, trace_bf16=kb), os.path.join(OUT, f"{name}.pt"))


def loop_full_width_inputs(cfg, seed=47):
    Fr, H, W, T = FULL_LOOP_DIMS
    g = _g(seed)
    latents = torch.randn(1, 16, Fr, H, W, generator=g).bfloat16().float()
    enc = torch.randn(2, T, cfg["text_embed_dim"], generator=g).bfloat16().float()
    return latents, enc


@section("swa_long")
def gen_swa_long(ns, shim):
    # ---- VERDICT r2 next #6c: sliding-window blocks on a grid where the window TRUNCATES in all six scan orders:
    # 9 x 16 x 16 patches = 2304 video tokens, window = 16 * 16 = 256 positions either side (F*H*W = 9 * H*W >> 2 * H*W), the
    # strided cross keys use interval 2.  Same reference processor, flash_attn_func restated (oracle/flash_attn_shim.py).
    from oracle import flash_attn_shim
    ns.processor.flash_attn_func = flash_attn_shim.flash_attn_func
    cfg = dict(TINY, num_attention_heads=6, num_layers=2, swa_layers=[0, 1])
    m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg).eval()
    shapes = _load_sd(m, 3, "stress")
    B, Fr, H, W, T = 2, 9, 32, 32, 24
    lat, enc = swa_inputs(cfg, 33, B, Fr, H, W, T)
    t = torch.tensor([303.0, 303.0]).to(torch.bfloat16).float()
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((H // 2, W // 2), 45, 30)
    cos, sin = shim.get_3d_rotary_pos_embed(64, cc, (H // 2, W // 2), Fr, use_real=True)
    out = m(lat, t, encoder_hidden_states=enc, image_rotary_emb=(cos, sin), return_dict=False)[0]
    mb = copy.deepcopy(m).to(torch.bfloat16)
    outb = mb(lat.bfloat16(), t.bfloat16(), encoder_hidden_states=enc.bfloat16(), image_rotary_emb=(cos, sin), return_dict=False)[0]
    print(f"  transformer_swa_long: out std {out.std().item():.3f}, floor {_mse(outb.float(), out):.3e}")
    torch.save(dict(cfg=cfg, shapes=shapes, seed=3, style="stress", input_seed=33, dims=(B, Fr, H, W, T), t=t, crops=cc,
                    lat_sum=lat.double().sum().item(), out=out.to(torch.float16), out_std=out.std().item(),
                    floor_mse=_mse(outb.float(), out)), os.path.join(OUT, "transformer_swa_long.pt"))


def vae_tiled_inputs(seed=19, frames=5, height=512, width=448):
    """Inputs of the tiled-VAE fixture: a 5 x 512 x 448 clip (two tile rows, two tile columns of different widths: 384 + 224 rows,
    384 + 160 columns) and a [1,16,2,64,56] latent."""
    g = _g(seed)
    video = (torch.rand(1, 3, frames, height, width, generator=g) * 2 - 1).bfloat16().float()
    z = torch.randn(1, 16, (frames - 1) // 4 + 1, height // 8, width // 8, generator=g).bfloat16().float()
    return video, z


@section("vae_tiled")
def gen_vae_tiled(ns, shim):
    # ---- the reference's spatial tiling (autoencoder_magvit.py:249-254,276-279,319-448) with its default tile
    # geometry (tile_sample_min_size 384, overlap 0.25): use_tiling=True, full-width VAE, fp32 on the host; encode moments and
    # decode output of the UNCHANGED reference, plus the untiled results' distance (tiling changes the result: that it does, and by
    # how much, is part of the fixture).
    import time
    vae, shapes = _meta_build(lambda: ns.autoencoder_magvit.AutoencoderKLMagvit(**dict(FULL_VAE, use_tiling=True)), 2, "default_bf16")
    video, z = vae_tiled_inputs()
    t0 = time.time()
    mom = vae.encode(video)[0].parameters
    t1 = time.time()
    dec = vae.decode(z)[0]
    print(f"  vae_tiled: tiled encode {t1 - t0:.0f} s, tiled decode {time.time() - t1:.0f} s", flush=True)
    vae.use_tiling = False
    mom_u = vae.encode(video)[0].parameters
    dec_u = vae.decode(z)[0]
    print(f"  vae_tiled: tiled-vs-untiled MSE: moments {_mse(mom, mom_u):.3e}, decode {_mse(dec, dec_u):.3e}", flush=True)
    torch.save(dict(cfg=dict(FULL_VAE, use_tiling=True), seed=2, style="default_bf16", input_seed=19, frames=5, height=512, width=448,
                    video_sum=video.double().sum().item(), z_sum=z.double().sum().item(), moments=mom.half(), dec=dec.half(),
                    untiled_mse=(_mse(mom, mom_u), _mse(dec, dec_u)), moments_fp16_mse=_mse(mom.half().float(), mom),
                    dec_fp16_mse=_mse(dec.half().float(), dec)), os.path.join(OUT, "vae_tiled_5x512x448.pt"))


DIT_12B = dict(FULL_DIT, num_layers=48)                       # BASELINE configs[2]: 12B (v5 MMDiT), SURVEY Appendix B
DIT_12B_INP = dict(FULL_DIT, num_layers=48, in_channels=33)   # BASELINE configs[4]: 12B InP (image-latent concat)
DEPTH_TAPS = (1, 12, 24, 48)          # residual streams kept after these many blocks
TAP_VIDEO_STRIDE, TAP_TEXT_STRIDE = 256, 4


def config3_inputs():
    """BASELINE.json configs[2]: 12B DiT at 49 x 1024 x 1024 -> latents [1,16,13,128,128] (53 248 video + 256 text tokens,
    S = 53 504); the text embedding of the CONDITIONAL sample (the CFG pair's second element).  bf16-representable values."""
    enc = (torch.randn(1, 256, 3584, generator=_g(71)) * 3).bfloat16().float()
    latents = torch.randn(1, 16, 13, 128, 128, generator=_g(72)).bfloat16().float()
    return latents, None, enc


def config5_inputs():
    """BASELINE.json configs[4]: 12B InP DiT at 49 x 768 x 768 -> latents [1,16,13,96,96] (29 952 video + 256 text tokens,
    S = 30 208) with the 17 conditioning channels pipeline_easyanimate_inpaint.py:1383 concatenates: the resized mask (0 on the
    first latent frame, 1 elsewhere: the I2V case of utils/utils.py:152-157) in front of 16 channels of masked-video latents
    (synthetic N(0, 0.5^2); the real ones are a VAE encode, config 4's subject).  bf16-representable values."""
    enc = (torch.randn(1, 256, 3584, generator=_g(81)) * 3).bfloat16().float()
    latents = torch.randn(1, 16, 13, 96, 96, generator=_g(82)).bfloat16().float()
    mask = torch.ones(1, 1, 13, 96, 96)
    mask[:, :, 0] = 0
    inp = torch.cat([mask, (torch.randn(1, 16, 13, 96, 96, generator=_g(83)) * 0.5).bfloat16().float()], 1)
    return latents, inp, enc


def _streamed_reference(ns, cfg, seed, style, dtype, taps):
    """The UNCHANGED reference EasyAnimateTransformer3DModel with its blocks' weights STREAMED: the module tree is built on the
    meta device, everything outside `transformer_blocks` is filled once, and each EasyAnimateDiTBlock is filled from
    synth_tensor by a forward-pre hook and returned to the meta device by a forward hook -- the 47 GB of fp32 block weights of
    the 12B model never coexist (one block = 0.9 GB).  The reference's own forward (transformer3d.py:1496-1689) drives the
    blocks; nothing of it is restated.  taps: {n: (video rows, text rows)} of the residual streams after n blocks."""
    with torch.device("meta"):
        m = ns.transformer3d.EasyAnimateTransformer3DModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    head = {k: s for k, s in shapes.items() if not k.startswith("transformer_blocks.")}
    sd = {k: v.to(dtype) for k, v in synth_state_dict(head, seed, style).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and all(k.startswith("transformer_blocks.") for k in missing)

    def fill(i):
        def pre(blk, args, kwargs):
            pre_ = f"transformer_blocks.{i}."
            sub = {k[len(pre_):]: s for k, s in shapes.items() if k.startswith(pre_)}
            vals = synth_state_dict({pre_ + k: s for k, s in sub.items()}, seed, style)
            blk.load_state_dict({k: vals[pre_ + k].to(dtype) for k in sub}, strict=True, assign=True)
        return pre

    def drop(i):
        def post(blk, args, kwargs, out):
            blk.to("meta")
            if i + 1 in taps:
                h, e = out
                taps[i + 1] = (h[0, ::TAP_VIDEO_STRIDE].float().clone(), e[0, ::TAP_TEXT_STRIDE].float().clone())
            if (i + 1) % 4 == 0:
                print(f"    block {i + 1}/{len(m.transformer_blocks)}: residual std video {out[0].float().std().item():.4f} "
                      f"text {out[1].float().std().item():.4f}", flush=True)
        return post

    for i, blk in enumerate(m.transformer_blocks):
        blk.register_forward_pre_hook(fill(i), with_kwargs=True)
        blk.register_forward_hook(drop(i), with_kwargs=True)
    return m.eval(), shapes


def _depth_golden(ns, shim, name, cfg, inputs, grid, note):
    import time
    lat, inp, enc = inputs
    Fr, gh, gw = grid
    cc = ns.pipeline_easyanimate.get_resize_crop_region_for_grid((gh, gw), 45, 30)
    rope = shim.get_3d_rotary_pos_embed(64, cc, (gh, gw), Fr, use_real=True)
    s = shim.FlowMatchEulerDiscreteScheduler(shift=1.0)
    s.set_timesteps(50, device="cpu", mu=1)
    tt = torch.tensor([s.timesteps[0]]).to(torch.bfloat16).float()       # pipeline_easyanimate.py:1079-1081
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        taps = {n: None for n in DEPTH_TAPS}
        m, shapes = _streamed_reference(ns, cfg, 0, "default_bf16", dtype, taps)
        c = lambda x: None if x is None else x.to(dtype)
        t0 = time.time()
        with torch.no_grad():
            v = m(c(lat), c(tt), encoder_hidden_states=c(enc), image_rotary_emb=rope, inpaint_latents=c(inp), return_dict=False)[0].float()
        print(f"  {name}: {dtype} forward {time.time() - t0:.0f} s on {torch.get_num_threads()} threads; v std {v.std().item():.4f} "
              f"|v| max {v.abs().max().item():.2f}", flush=True)
        res[dtype] = (v, taps)
        del m
    v, taps = res[torch.float32]
    vb, taps_b = res[torch.bfloat16]
    floor_taps = {n: (_mse(taps_b[n][0], taps[n][0]), _mse(taps_b[n][1], taps[n][1])) for n in DEPTH_TAPS}
    print(f"  {name}: reference bf16-vs-fp32 velocity MSE {_mse(vb, v):.3e}; residual-stream floors (video, text) by depth {floor_taps}", flush=True)
    rec = dict(cfg=cfg, seed=0, style="default_bf16", timestep=float(tt), crops=cc, grid=grid, note=note,
               v_sub_f16=v[..., ::2, ::2].half().contiguous(), v_shape=tuple(v.shape), v_std=v.std().item(),
               v_frame_sums=v.double().sum(dim=(0, 1, 3, 4)), v_abs_max=v.abs().max().item(),
               floor_mse=_mse(vb, v), rel_l2_floor=((vb - v).double().norm() / v.double().norm()).item(),
               floor_mse_sub=_mse(vb[..., ::2, ::2], v[..., ::2, ::2]),
               fp16_storage_mse=_mse(v[..., ::2, ::2].half().float(), v[..., ::2, ::2]),
               taps={n: (taps[n][0].half(), taps[n][1].half()) for n in DEPTH_TAPS},
               tap_std={n: (taps[n][0].std().item(), taps[n][1].std().item()) for n in DEPTH_TAPS},
               tap_floor_mse=floor_taps, tap_strides=(TAP_VIDEO_STRIDE, TAP_TEXT_STRIDE),
               latents_sum=lat.double().sum().item(), enc_sum=enc.double().sum().item(),
               inp_sum=None if inp is None else inp.double().sum().item())
    torch.save(rec, os.path.join(OUT, f"{name}.pt"))


@section("config3_forward")
def gen_config3_forward(ns, shim):
    # ---- BASELINE configs[2] at its DECLARED DEPTH x LENGTH -- the 12B model (L = 48, d = 3072) at
    # 49 x 1024^2 (S = 53 504), the conditional sample (B = 1), first timestep of the 50-step Flow schedule: the unchanged
    # reference in fp32 (2.3e15 FLOP, > 1 h of the 8 host cores), block-streamed, then its own bf16 forward (the floor).
    _depth_golden(ns, shim, "config3_12b_49x1024_v0", DIT_12B, config3_inputs(), (13, 64, 64),
                  "12B L=48 d=3072, 49f x 1024^2, S=53504, B=1 (conditional sample), block-streamed unchanged reference")


@section("config5_forward")
def gen_config5_forward(ns, shim):
    # ---- BASELINE configs[4]: the 12B InP model (33 input channels) at 49 x 768^2 (S = 30 208), same protocol.
    _depth_golden(ns, shim, "config5_12b_inp_49x768_v0", DIT_12B_INP, config5_inputs(), (13, 48, 48),
                  "12B InP L=48 d=3072 Cin=33, 49f x 768^2, S=30208, B=1 (conditional sample), block-streamed unchanged reference")


@torch.no_grad()
def main(only=None):
    os.makedirs(OUT, exist_ok=True)
    ns = ref_loader.load()
    shim = ns.shim
    for name, fn in SECTIONS.items():
        if only and name not in only:
            continue
        print(f"== {name}", flush=True)
        fn(ns, shim)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main(sys.argv[1:] or None)
