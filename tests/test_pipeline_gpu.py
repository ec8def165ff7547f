"""End-to-end: EasyAnimatePipeline.__call__ (T2V) and EasyAnimateInpaintPipeline.__call__ (I2V, predict_i2v.py path) on
the GPU -- tiny DiT + tiny MAGVIT VAE with synthetic weights -- against the oracle restatement of the same chain on CPU
(prepare latents -> [VAE-encode the masked video, resize the mask] -> CFG Flow loop -> VAE decode -> [0,1] frames).
This is SURVEY 8d "config 1" (the plumbing case) at test size, plus row P4."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _models(in_channels):
    from easyanimate_amd import AutoencoderKLMagvit, EasyAnimateTransformer3DModel
    from easyanimate_amd.synthetic import synth_state_dict
    gt = torch.load(os.path.join(GOLD, "transformer_t2v.pt" if in_channels == 16 else "transformer_inp.pt"), weights_only=False)
    gv = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    sd_t = synth_state_dict(gt["shapes"], gt["seed"], gt["style"])
    sd_v = synth_state_dict(gv["shapes"], gv["seed"], gv["style"])
    m = EasyAnimateTransformer3DModel.from_config(gt["cfg"])
    m.load_state_dict(sd_t, strict=True)
    vae = AutoencoderKLMagvit.from_config(gv["cfg"])
    vae.load_state_dict(sd_v, strict=True)
    return m.to(torch.bfloat16).to(DEV).eval(), vae.to(torch.bfloat16).to(DEV).eval(), sd_t, sd_v, gt["cfg"], gv["cfg"]


def _report(name, got, ref):
    got, ref = torch.as_tensor(got).double(), ref.double()
    mse = ((got - ref) ** 2).mean().item()
    mx = (got - ref).abs().max().item()
    print(f"[parity] {name}: frames in [0,1]: MSE={mse:.3e} max_abs={mx:.3e} ref_std={ref.std().item():.3f}")
    return mse, mx


def _oracle_frames(sd_t, cfg_t, sd_v, cfg_v, latents, enc_np, rope, steps, guidance, inpaint=None):
    from oracle import restatement as R
    from oracle import restatement_vae as RV
    with torch.no_grad():
        z = R.denoise_loop(sd_t, cfg_t, latents, enc_np, rope, steps, guidance, inpaint_latents=inpaint)
        video = RV.vae_decode(sd_v, z / 0.1825 if "scaling_factor" not in cfg_v else z / cfg_v["scaling_factor"],
                              cfg_v["norm_num_groups"])
    return (video.clamp(-1, 1) / 2 + 0.5).clamp(0, 1)


def test_t2v_pipeline_end_to_end():
    from easyanimate_amd import EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler
    m, vae, sd_t, sd_v, cfg_t, cfg_v = _models(16)
    pipe = EasyAnimatePipeline(vae=vae, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    g = torch.Generator().manual_seed(11)
    F_, H, W, T, steps, guidance = 9, 64, 64, 7, 4, 6.0
    latents = torch.randn(pipe.latent_shape(1, 16, F_, H, W), generator=g)
    assert tuple(latents.shape) == (1, 16, 3, 8, 8)
    pos = torch.randn(1, T, cfg_t["text_embed_dim"], generator=g)
    neg = torch.randn(1, T, cfg_t["text_embed_dim"], generator=g)
    out = pipe(video_length=F_, height=H, width=W, num_inference_steps=steps, guidance_scale=guidance,
               latents=latents.to(torch.bfloat16), prompt_embeds=pos, negative_prompt_embeds=neg, output_type="np")
    frames = out.frames
    assert frames.shape == (1, 3, F_, H, W) and frames.min() >= 0 and frames.max() <= 1
    rope = tuple(t.cpu() for t in pipe.rotary_embedding(H, W, 3))   # resident on the device in the product; the oracle runs on the host
    vcfg = dict(cfg_v, scaling_factor=vae.config.scaling_factor)
    ref = _oracle_frames(sd_t, cfg_t, sd_v, vcfg, latents.bfloat16().float(), torch.cat([neg, pos]).bfloat16().float(), rope,
                         steps, guidance)
    mse, mx = _report("t2v pipeline 9f x 64^2, 4 steps, CFG 6", frames, ref)
    assert mse < 2e-4   # pixel scale [0,1]; the loop's bf16 noise (CFG x11) passed through the decoder


def test_i2v_pipeline_end_to_end():
    """predict_i2v.py path: start image -> get_image_to_video_latent -> mask / masked video -> VAE encode ->
    inpaint_latents [2, 17, f, h, w] -> InP transformer (in_channels 33) loop -> decode."""
    from easyanimate_amd import EasyAnimateInpaintPipeline, FlowMatchEulerDiscreteScheduler
    from easyanimate_amd.pipeline import get_image_to_video_latent, resize_mask
    from oracle import restatement_vae as RV
    m, vae, sd_t, sd_v, cfg_t, cfg_v = _models(33)
    m.resize_inpaint_mask_directly = True    # V5.1 yaml
    pipe = EasyAnimateInpaintPipeline(vae=vae, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    g = torch.Generator().manual_seed(13)
    F_, H, W, T, steps, guidance = 9, 64, 64, 7, 3, 6.0
    image = torch.rand(3, H, W, generator=g)
    video, mask = get_image_to_video_latent(image, F_)
    latents = torch.randn(1, 16, 3, 8, 8, generator=g)
    pos = torch.randn(1, T, cfg_t["text_embed_dim"], generator=g)
    neg = torch.randn(1, T, cfg_t["text_embed_dim"], generator=g)
    out = pipe(video_length=F_, video=video, mask_video=mask, height=H, width=W, num_inference_steps=steps,
               guidance_scale=guidance, latents=latents.to(torch.bfloat16), prompt_embeds=pos, negative_prompt_embeds=neg,
               output_type="np")
    frames = out.frames
    assert frames.shape == (1, 3, F_, H, W)
    # ---- oracle chain on CPU (fp32 arithmetic on the same bf16-representable inputs)
    s = vae.config.scaling_factor
    masked_video, mask_c = EasyAnimateInpaintPipeline.masked_video_and_mask(video, mask)
    with torch.no_grad():
        mom = RV.vae_encode_moments(sd_v, masked_video.bfloat16().float(), cfg_v["norm_num_groups"])
    masked_lat = mom[:, :16] * s                                  # .mode() of the diagonal Gaussian
    mask_lat = resize_mask(1 - mask_c, masked_lat, True) * s
    inpaint = torch.cat([mask_lat, masked_lat], 1)
    inpaint = torch.cat([inpaint] * 2)
    rope = tuple(t.cpu() for t in pipe.rotary_embedding(H, W, 3))   # resident on the device in the product; the oracle runs on the host
    vcfg = dict(cfg_v, scaling_factor=s)
    ref = _oracle_frames(sd_t, cfg_t, sd_v, vcfg, latents.bfloat16().float(), torch.cat([neg, pos]).bfloat16().float(), rope,
                         steps, guidance, inpaint=inpaint.bfloat16().float())
    # conditioning tensor itself
    with torch.no_grad():
        got_c = pipe.inpaint_conditioning(video, mask, torch.bfloat16, DEV, True)
    assert got_c.shape == (2, 17, 3, 8, 8)
    cm, _ = _report("i2v inpaint_latents", got_c.float().cpu(), inpaint)
    assert cm < 1e-4
    mse, mx = _report("i2v pipeline 9f x 64^2, 3 steps, CFG 6", frames, ref)
    assert mse < 2e-4
