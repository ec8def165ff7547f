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
    m = EasyAnimateTransformer3DModel.from_config(dict(gt["cfg"], enable_clip_in_inpaint=False))     # the V5 / V5.1 YAML value
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


class _DeviceTokenizer:
    """Stand-in for the Qwen2 tokenizer files (unreachable offline): chat template + byte-level ids, padded to max_length, with a
    BatchEncoding-like `.to(device)`."""
    model_max_length = 512

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
        return "".join(f"<|im_start|>{m['role']}\n{m['content'][0]['text']}<|im_end|>\n" for m in messages) + "<|im_start|>assistant\n"

    def __call__(self, text=None, padding=None, max_length=None, truncation=None, return_attention_mask=None, padding_side=None,
                 return_tensors=None):
        assert padding == "max_length" and padding_side == "right" and return_tensors == "pt"
        rows, masks = [], []
        for t in text:
            ids = [3 + b % 97 for b in t.encode()][:max_length]
            masks.append([1] * len(ids) + [0] * (max_length - len(ids)))
            rows.append(ids + [0] * (max_length - len(ids)))

        class _Enc:
            def __init__(self, ids, mask):
                self.input_ids, self.attention_mask = ids, mask

            def to(self, device):
                return _Enc(self.input_ids.to(device), self.attention_mask.to(device))
        return _Enc(torch.tensor(rows), torch.tensor(masks))


def test_t2v_pipeline_from_prompt_strings():
    """The text-encoder step in front of the loop (pipeline_easyanimate.py:421-460): the text_encoder slot holds
    the `transformers` class the reference loads (Qwen2VLForConditionalGeneration, tiny random-init configuration, on the GPU in
    bf16), prompts go in as STRINGS.  The embeddings the pipeline derives equal the fp32 CPU run of the same encoder to bf16
    noise, and the frames equal the oracle chain fed with those embeddings."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_host_cpu import tiny_qwen2vl
    from easyanimate_amd import EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler
    m, vae, sd_t, sd_v, cfg_t, cfg_v = _models(16)
    enc_cpu = tiny_qwen2vl(cfg_t["text_embed_dim"])
    enc = tiny_qwen2vl(cfg_t["text_embed_dim"]).to(torch.bfloat16).to(DEV)
    tok = _DeviceTokenizer()
    pipe = EasyAnimatePipeline(vae=vae, text_encoder=enc, tokenizer=tok, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    F_, H, W, steps, guidance = 9, 64, 64, 4, 6.0
    g = torch.Generator().manual_seed(21)
    latents = torch.randn(pipe.latent_shape(1, 16, F_, H, W), generator=g)
    prompt, negative = "a dog shakes its head", "blurry, static, low quality"
    with torch.no_grad():
        pe, ne, pm, nm = pipe.encode_prompt(prompt, DEV, torch.bfloat16, 1, True, negative)
        ref_pipe = EasyAnimatePipeline(vae=None, text_encoder=enc_cpu, tokenizer=tok, transformer=m, scheduler=None)
        pe32, ne32, _, _ = ref_pipe.encode_prompt(prompt, "cpu", torch.float32, 1, True, negative)
    assert pe.shape == (1, 256, cfg_t["text_embed_dim"]) and pe.device.type == "cuda" and int(pm.sum()) < 256
    for a, b in ((pe, pe32), (ne, ne32)):
        err = ((a.float().cpu() - b) ** 2).mean().item() / (b ** 2).mean().item()
        assert err < 1e-3, err          # the encoder forward itself is transformers' code (bf16 on the GPU vs fp32 on the host)
    out = pipe(prompt=prompt, negative_prompt=negative, video_length=F_, height=H, width=W, num_inference_steps=steps,
               guidance_scale=guidance, latents=latents.to(torch.bfloat16), output_type="np")
    out2 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, video_length=F_, height=H, width=W, num_inference_steps=steps,
                guidance_scale=guidance, latents=latents.to(torch.bfloat16), output_type="np")
    assert out.frames.shape == (1, 3, F_, H, W) and (out.frames == out2.frames).all()     # strings and their embeddings: the same call
    rope = tuple(t.cpu() for t in pipe.rotary_embedding(H, W, 3))
    vcfg = dict(cfg_v, scaling_factor=vae.config.scaling_factor)
    ref = _oracle_frames(sd_t, cfg_t, sd_v, vcfg, latents.bfloat16().float(), torch.cat([ne, pe]).float().cpu(), rope, steps, guidance)
    mse, mx = _report("t2v pipeline from prompt strings (Qwen2VL class in the text_encoder slot)", out.frames, ref)
    assert mse < 2e-4


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


# ---- the remaining branches of the two pipelines (VERDICT r1 "missing" 7) ---------------------------------
def test_guidance_rescale_kernel_and_loop():
    """rescale_noise_cfg (pipeline_easyanimate.py:100-112,1106-1108): the fused device reduction + step against fp64, and
    the whole loop with guidance_rescale = 0.7 against the oracle loop."""
    from easyanimate_amd import EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler, ops
    from oracle import restatement as R
    g = torch.Generator().manual_seed(3)
    for dt in (torch.float32, torch.bfloat16):
        v = torch.randn(2, 16, 3, 16, 24, generator=g).to(dt)
        x = torch.randn(1, 16, 3, 16, 24, generator=g).to(dt)
        vd, xd = v.double(), x.double()
        cfg = vd[0:1] + 6.0 * (vd[1:2] - vd[0:1])
        if dt == torch.bfloat16:
            cfg = cfg.to(dt).double()
        ref = xd + (-0.25) * R.rescale_noise_cfg(cfg, vd[1:2], 0.7)
        xg = x.to(DEV).contiguous()
        ops.cfg_rescale_euler_step(v.to(DEV).contiguous(), xg, 6.0, -0.25, 0.7)
        err = (xg.double().cpu() - ref).abs().max().item()
        print(f"[parity] cfg_rescale_euler_step {dt}: max abs err {err:.3e}")
        assert err < (2e-2 if dt == torch.bfloat16 else 2e-5)
    m, vae, sd_t, sd_v, cfg_t, cfg_v = _models(16)
    pipe = EasyAnimatePipeline(vae=None, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    latents = torch.randn(1, 16, 3, 8, 8, generator=g).bfloat16().float()
    enc = torch.randn(2, 7, cfg_t["text_embed_dim"], generator=g).bfloat16().float()
    out = pipe(video_length=9, height=64, width=64, num_inference_steps=6, guidance_scale=6.0, guidance_rescale=0.7,
               latents=latents.clone(), prompt_embeds=enc[1:2], negative_prompt_embeds=enc[0:1]).frames
    rope = tuple(t.cpu() for t in pipe.rotary_embedding(64, 64, 3))
    with torch.no_grad():
        ref = R.denoise_loop(sd_t, cfg_t, latents, enc, rope, 6, 6.0, guidance_rescale=0.7)
        ref0 = R.denoise_loop(sd_t, cfg_t, latents, enc, rope, 6, 6.0)
    mse = ((out.float().cpu() - ref) ** 2).mean().item()
    print(f"[parity] 6-step loop with guidance_rescale 0.7: latent MSE vs oracle {mse:.3e} (the rescale moves the result by "
          f"{((ref - ref0) ** 2).mean().item():.3e})")
    assert mse < 1e-4 and not torch.equal(ref, ref0)


def test_inpaint_pipeline_refuses_clip_conditioning():
    """An inpaint checkpoint whose config says enable_clip_in_inpaint (the constructor default) would receive CLIP tokens -- zeros
    without a clip_image -- in every forward of the reference (pipeline_easyanimate_inpaint.py:1296-1311,1509-1513): not built,
    and never skipped silently."""
    from easyanimate_amd import EasyAnimateInpaintPipeline, EasyAnimateTransformer3DModel, FlowMatchEulerDiscreteScheduler
    gt = torch.load(os.path.join(GOLD, "transformer_inp.pt"), weights_only=False)
    with torch.device("meta"):
        m = EasyAnimateTransformer3DModel.from_config(gt["cfg"])
    assert m.config.get("enable_clip_in_inpaint", True) is True and m.config.in_channels == 33
    pipe = EasyAnimateInpaintPipeline(vae=None, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    with pytest.raises(NotImplementedError, match="enable_clip_in_inpaint"):
        pipe(prompt_embeds=torch.zeros(1, 7, 48), negative_prompt_embeds=torch.zeros(1, 7, 48), video_length=9, height=64, width=64,
             num_inference_steps=2)


def test_inpaint_pipeline_branches():
    """EasyAnimateInpaintPipeline: the all-255 mask zero-latent shortcut (:1322-1336), add_noise_in_inpaint_model
    (:153-167,799), the VAE-encoded mask of resize_inpaint_mask_directly=False (:1364-1377), strength < 1 (:760-767,
    :862-893), callbacks and the output_type convention (:1592-1606)."""
    from easyanimate_amd import EasyAnimateInpaintPipeline, FlowMatchEulerDiscreteScheduler
    from easyanimate_amd.pipeline import add_noise_to_reference_video, get_image_to_video_latent
    from oracle import restatement as R
    from oracle import restatement_vae as RV
    m, vae, sd_t, sd_v, cfg_t, cfg_v = _models(33)
    m.resize_inpaint_mask_directly = True
    pipe = EasyAnimateInpaintPipeline(vae=vae, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    g = torch.Generator().manual_seed(29)
    F_, H, W = 9, 64, 64
    s = vae.config.scaling_factor
    video, mask = get_image_to_video_latent(torch.rand(3, H, W, generator=g), F_)
    # 1. all-255 mask (T2V through an InP checkpoint): zero conditioning, no VAE call
    c0 = pipe.inpaint_conditioning(video, torch.full_like(mask, 255.0), torch.bfloat16, DEV, True, latent_shape=(1, 16, 3, 8, 8))
    assert c0.shape == (2, 17, 3, 8, 8) and c0.abs().max().item() == 0
    c1 = pipe.inpaint_conditioning(None, None, torch.bfloat16, DEV, False, latent_shape=(1, 16, 3, 8, 8))
    assert c1.shape == (1, 17, 3, 8, 8) and c1.abs().max().item() == 0
    # 2. noise augmentation: same generator -> same noise as the reference formula; masked (-1) pixels untouched
    masked_video, mask_c = EasyAnimateInpaintPipeline.masked_video_and_mask(video, mask)
    aug = add_noise_to_reference_video(masked_video, ratio=0.0563, generator=torch.Generator().manual_seed(5))
    noise = torch.randn(masked_video.size(), generator=torch.Generator().manual_seed(5)) * 0.0563
    assert torch.equal(aug, masked_video + torch.where(masked_video == -1, torch.zeros_like(noise), noise))
    assert torch.equal(aug[:, :, 1:], masked_video[:, :, 1:]) and not torch.equal(aug[:, :, 0], masked_video[:, :, 0])
    object.__setattr__(m, "_internal_dict", type(m.config)(dict(m.config, add_noise_in_inpaint_model=True)))
    c_aug = pipe.inpaint_conditioning(video, mask, torch.bfloat16, DEV, False, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref_aug = RV.vae_encode_moments(sd_v, aug.bfloat16().float(), cfg_v["norm_num_groups"])[:, :16] * s
    mse = ((c_aug[:, 1:].float().cpu() - ref_aug) ** 2).mean().item()
    print(f"[parity] add_noise_in_inpaint_model: masked-video latents MSE vs oracle {mse:.3e}")
    assert mse < 1e-4
    object.__setattr__(m, "_internal_dict", type(m.config)(dict(m.config, add_noise_in_inpaint_model=False)))
    # 3. resize_inpaint_mask_directly = False: the (tiled, binarised) mask goes through the VAE: 16 + 16 channels
    m.resize_inpaint_mask_directly = False
    c2 = pipe.inpaint_conditioning(video, mask, torch.bfloat16, DEV, False)
    with torch.no_grad():
        ref_m = RV.vae_encode_moments(sd_v, torch.tile(mask_c, [1, 3, 1, 1, 1]), cfg_v["norm_num_groups"])[:, :16] * s
        ref_v = RV.vae_encode_moments(sd_v, masked_video.bfloat16().float(), cfg_v["norm_num_groups"])[:, :16] * s
    assert c2.shape == (1, 32, 3, 8, 8)
    mse = ((c2.float().cpu() - torch.cat([ref_m, ref_v], 1)) ** 2).mean().item()
    print(f"[parity] VAE-encoded mask conditioning (resize_inpaint_mask_directly=False): MSE vs oracle {mse:.3e}")
    assert mse < 1e-4
    with pytest.raises(ValueError, match="input channels"):   # 16 + 32 != 33
        pipe(video_length=F_, video=video, mask_video=mask, height=H, width=W, num_inference_steps=2, guidance_scale=6.0,
             prompt_embeds=torch.zeros(1, 7, cfg_t["text_embed_dim"]), negative_prompt_embeds=torch.zeros(1, 7, cfg_t["text_embed_dim"]))
    m.resize_inpaint_mask_directly = True
    # 4. strength 0.5: the last half of the schedule, starting from scale_noise(encoded video, t_start, noise)
    pos = torch.randn(1, 7, cfg_t["text_embed_dim"], generator=g).bfloat16().float()
    neg = torch.randn(1, 7, cfg_t["text_embed_dim"], generator=g).bfloat16().float()
    seen = []
    vae_keep, pipe.vae = pipe.vae, vae
    out = pipe(video_length=F_, video=video, mask_video=mask, height=H, width=W, num_inference_steps=6, guidance_scale=6.0, strength=0.5,
               generator=torch.Generator().manual_seed(41), prompt_embeds=pos, negative_prompt_embeds=neg, output_type="latent",
               callback_on_step_end=lambda p, i, t, kw: seen.append((i, float(t), kw["latents"].float().cpu())) or {})
    assert isinstance(out.frames, torch.Tensor) and out.frames.shape == (1, 3, F_, H, W)     # "latent" = a tensor of the decoded video
    assert [i for i, _, _ in seen] == [0, 1, 2] and pipe.num_timesteps == 3
    ts, sig = R.flow_sigmas(6)
    assert [round(t, 3) for _, t, _ in seen] == [round(float(x), 3) for x in ts[3:]]
    noise = torch.randn((1, 16, 3, 8, 8), generator=torch.Generator().manual_seed(41), dtype=torch.bfloat16).float()
    with torch.no_grad():
        vlat = (RV.vae_encode_moments(sd_v, (video * 2 - 1).bfloat16().float(), cfg_v["norm_num_groups"])[:, :16] * s)
        x0 = (sig[3] * noise + (1 - sig[3]) * vlat).bfloat16().float()
        inp = torch.cat([torch.cat([pipe_resize(1 - mask_c, vlat) * s, ref_v], 1)] * 2).bfloat16().float()
        rope = tuple(t.cpu() for t in pipe.rotary_embedding(H, W, 3))
        ref = R.denoise_loop(sd_t, cfg_t, x0, torch.cat([neg, pos]), rope, 6, 6.0, inpaint_latents=inp, first_step=3)
    mse = ((seen[-1][2] - ref) ** 2).mean().item()
    print(f"[parity] strength 0.5 (V2V re-noising, 3 of 6 steps): final latent MSE vs oracle {mse:.3e}")
    assert mse < 2e-4


def pipe_resize(mask, latent):
    from easyanimate_amd.pipeline import resize_mask
    return resize_mask(mask, latent, True)


def test_t2v_pipeline_with_fp8_stored_transformer():
    """predict_t2v.py's default GPU_memory_mode hands the pipeline a transformer whose every parameter is
    float8_e4m3fn.  `transformer.dtype` is then a storage type: the pipeline must embed / sample / step in the VAE's (text
    encoder's) bf16 -- `compute_dtype` -- and its result must equal the run of a transformer that holds the same
    fp8-representable values in bf16, bit for bit (the W8 kernels widen exactly); a `torch.Generator` sampling the latents
    must work (randn has no fp8 kernel)."""
    import copy
    from easyanimate_amd import EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler
    m, vae, sd_t, sd_v, cfg_t, cfg_v = _models(16)
    m8, mq = copy.deepcopy(m), copy.deepcopy(m)
    for p8, pq in zip(m8.parameters(), mq.parameters()):
        q = p8.data.to(torch.float8_e4m3fn)
        p8.data = q                                  # convert_model_weight_to_float8 (predict_t2v.py:266)
        pq.data = q.to(torch.bfloat16)
    assert m8.dtype == torch.float8_e4m3fn
    g = torch.Generator().manual_seed(11)
    F_, H, W, T, steps, guidance = 5, 64, 64, 7, 3, 6.0
    pos = torch.randn(1, T, cfg_t["text_embed_dim"], generator=g)
    neg = torch.randn(1, T, cfg_t["text_embed_dim"], generator=g)
    outs = []
    for model in (m8, mq):
        pipe = EasyAnimatePipeline(vae=vae, transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
        assert pipe.compute_dtype == torch.bfloat16
        pipe.enable_model_cpu_offload()              # what the script calls in this mode: everything stays resident
        out = pipe(video_length=F_, height=H, width=W, num_inference_steps=steps, guidance_scale=guidance,
                   generator=torch.Generator(device="cuda").manual_seed(43), prompt_embeds=pos, negative_prompt_embeds=neg,
                   output_type="latent")
        outs.append(out.frames)
    assert outs[0].shape == (1, 3, F_, H, W) and torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
