#!/usr/bin/env python
"""predict_t2v.py on MI355X: the same flow as /root/reference/predict_t2v.py, step for step, over easyanimate_amd.

    reference predict_t2v.py                                   here
    -------------------------------------------------------    ----------------------------------------------------
    :91      OmegaConf.load(config_path)                       config.load_yaml(config_path)            (PyYAML)
    :94-108  name_to_transformer3d[...].from_pretrained_2d     same registry key, same call
    :135-142 name_to_autoencoder_magvit[...].from_pretrained   same registry key, same call, .to(weight_dtype)
    :158-205 tokenizers / text encoders                        OUT OF SCOPE: prompt embeddings are
                                                               inputs (--embeds file, or seeded synthetic ones)
    :219-231 FlowMatchEulerDiscreteScheduler.from_pretrained   same call (scheduler/scheduler_config.json)
    :233-254 Inpaint pipeline iff in_channels != latent ch.    same test, same constructor slots
    :256-273 offload / fp8 wrappers                            not needed: everything is resident in 288 GB of HBM;
                                                               --fp8-storage keeps the checkpoint's fp8 storage mode
    :275-278 transformer.enable_teacache(...)                  same call (--teacache THRESH)
    :280     torch.Generator(device="cuda").manual_seed(seed)  torch.Generator("cpu") -- config 1 draws its latents on
                                                               the host so that the CPU oracle sees the same noise
    :285-317 pipeline(...).frames                              same keywords, plus prompt_embeds / negative_prompt_embeds
    :322-338 save png / mp4                                    frames saved as .npy (and .png for a single frame if PIL
                                                               is importable)

There are no checkpoints offline, so `--make-synthetic {tiny,7b,12b}` first WRITES a checkpoint directory in the HF
layout the reference loads (<dir>/{transformer,vae,scheduler}/config.json + diffusion_pytorch_model.safetensors) holding
seeded synthetic bf16 weights (easyanimate_amd.synthetic); everything after that goes through the loaders.

    python examples/predict_t2v_mi355x.py --make-synthetic tiny --model-dir /tmp/ea_tiny --height 64 --width 64 \
           --video-length 9 --steps 4
    python examples/predict_t2v_mi355x.py --make-synthetic 7b --model-dir /tmp/ea_7b --height 256 --width 256 \
           --video-length 1 --steps 2          # BASELINE.json configs[0] at its declared dims
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the 7 + 8 + 2 keys of /root/reference/config/easyanimate_video_v5.1_magvit_qwen.yaml (data, not code: they are the
# constructor kwargs the reference merges over the checkpoint's config.json, predict_t2v.py:98-108,139-142)
V51_YAML = {
    "transformer_additional_kwargs": {
        "transformer_type": "EasyAnimateTransformer3DModel", "after_norm": False, "time_position_encoding_type": "3d_rope",
        "resize_inpaint_mask_directly": True, "enable_text_attention_mask": True, "enable_clip_in_inpaint": False,
        "add_ref_latent_in_control_model": True},
    "vae_kwargs": {
        "vae_type": "AutoencoderKLMagvit", "mini_batch_encoder": 4, "mini_batch_decoder": 1, "slice_mag_vae": False,
        "slice_compression_vae": False, "cache_compression_vae": False, "cache_mag_vae": True},
    "text_encoder_kwargs": {"enable_multi_text_encoder": False, "replace_t5_to_llm": True},
}

_DIT = dict(num_attention_heads=48, attention_head_dim=64, in_channels=16, out_channels=16, patch_size=2, num_layers=48,
            time_embed_dim=512, add_norm_text_encoder=True, text_embed_dim=3584, text_embed_dim_t5=None, norm_eps=1e-5,
            time_position_encoding_type="3d_rope", enable_text_attention_mask=True)
_VAE = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512],
            down_block_types=["SpatialDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D"],
            up_block_types=["SpatialUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D"],
            mid_block_attention_type="spatial", latent_channels=16, norm_num_groups=32, spatial_group_norm=True,
            cache_mag_vae=True, slice_mag_vae=False, cache_compression_vae=False, slice_compression_vae=False,
            mini_batch_encoder=4, mini_batch_decoder=1, layers_per_block=2, scaling_factor=0.1825)
SYNTHETIC = {
    # declared synthetic architectures, SURVEY Appendix B (the HF config.json files are not in the reference tree)
    "12b": (dict(_DIT), dict(_VAE)),
    "7b": (dict(_DIT, num_layers=28), dict(_VAE)),
    "tiny": (dict(_DIT, num_attention_heads=2, num_layers=2, time_embed_dim=64, text_embed_dim=48),
             dict(_VAE, block_out_channels=[64, 64, 128, 128], norm_num_groups=16)),
}


def write_yaml(path: str, cfg: dict = V51_YAML) -> str:
    import yaml
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f, sort_keys=False)
    return path


def make_synthetic_checkpoint(model_dir: str, which: str = "tiny", dit_seed: int = 0, vae_seed: int = 2,
                              style: str = "default_bf16", in_channels: int | None = None, shard_bytes: int = 4 << 30) -> str:
    """Write <model_dir>/{transformer,vae,scheduler}/ in the layout predict_t2v.py loads: config.json +
    diffusion_pytorch_model*.safetensors with seeded synthetic bf16 weights.  Large models are written as several
    safetensors shards (the glob branch of from_pretrained_2d, transformer3d.py:1763-1770)."""
    from safetensors.torch import save_file

    from easyanimate_amd import AutoencoderKLMagvit, EasyAnimateTransformer3DModel
    from easyanimate_amd.synthetic import synth_tensor
    dit_cfg, vae_cfg = SYNTHETIC[which]
    dit_cfg = dict(dit_cfg, **({"in_channels": in_channels} if in_channels else {}))
    for sub, cls, cfg, seed in (("transformer", EasyAnimateTransformer3DModel, dit_cfg, dit_seed),
                                ("vae", AutoencoderKLMagvit, vae_cfg, vae_seed)):
        d = os.path.join(model_dir, sub)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(dict(cfg, _class_name=cls.__name__, _diffusers_version="0.30.1"), f, indent=1)
        with torch.device("meta"):
            shapes = {k: tuple(v.shape) for k, v in cls.from_config(cfg).state_dict().items()}
        total = sum(int(np.prod(s)) * 2 for s in shapes.values())
        shard, size, idx = {}, 0, 0

        def flush():
            nonlocal shard, size, idx
            if shard:
                name = "diffusion_pytorch_model.safetensors" if total <= shard_bytes else f"diffusion_pytorch_model-{idx:05d}.safetensors"
                save_file(shard, os.path.join(d, name))
                shard, size, idx = {}, 0, idx + 1
        for k, shp in shapes.items():
            t = synth_tensor(k, shp, seed, style).to(torch.bfloat16)
            shard[k] = t
            size += t.numel() * 2
            if total > shard_bytes and size >= shard_bytes:
                flush()
        flush()
    d = os.path.join(model_dir, "scheduler")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "scheduler_config.json"), "w") as f:
        json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "num_train_timesteps": 1000, "shift": 1.0,
                   "use_dynamic_shifting": False}, f, indent=1)
    return model_dir


def synthetic_embeddings(text_embed_dim: int, tokens: int = 256, seed: int = 1):
    """[negative, positive] prompt embeddings, N(0,1) from generator seed 1, bf16-representable."""
    e = torch.randn(2, tokens, text_embed_dim, generator=torch.Generator().manual_seed(seed)).bfloat16().float()
    return e[1:2], e[0:1]   # (prompt_embeds, negative_prompt_embeds)


def build_pipeline(config_path: str, model_name: str, device: str = "cuda", weight_dtype=torch.bfloat16,
                   fp8_storage: bool = False, teacache_threshold: float | None = None, num_inference_steps: int = 50):
    """predict_t2v.py:91-278."""
    from easyanimate_amd import (EasyAnimateInpaintPipeline, EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler,
                                 get_teacache_coefficients, name_to_autoencoder_magvit, name_to_transformer3d)
    from easyanimate_amd.config import load_yaml
    config = load_yaml(config_path)
    Choosen_Transformer3DModel = name_to_transformer3d[
        config["transformer_additional_kwargs"].get("transformer_type", "Transformer3DModel")]
    transformer = Choosen_Transformer3DModel.from_pretrained_2d(
        model_name, subfolder="transformer", transformer_additional_kwargs=dict(config["transformer_additional_kwargs"]),
        torch_dtype=torch.float8_e4m3fn if fp8_storage else weight_dtype, low_cpu_mem_usage=True)
    Choosen_AutoencoderKL = name_to_autoencoder_magvit[config["vae_kwargs"].get("vae_type", "AutoencoderKL")]
    vae = Choosen_AutoencoderKL.from_pretrained(model_name, subfolder="vae",
                                                vae_additional_kwargs=dict(config["vae_kwargs"])).to(weight_dtype)
    scheduler = FlowMatchEulerDiscreteScheduler.from_pretrained(model_name, subfolder="scheduler")
    slots = dict(text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, vae=vae, transformer=transformer,
                 scheduler=scheduler)
    if transformer.config.in_channels != vae.config.latent_channels:
        pipeline = EasyAnimateInpaintPipeline(**slots)
    else:
        pipeline = EasyAnimatePipeline(**slots)
    # predict_t2v.py:256-273 -- every GPU_memory_mode branch ends in one of these; here they make the models resident
    # (288 GB of HBM: nothing is offloaded, no hooks)
    pipeline.enable_model_cpu_offload(device=device)
    coefficients = get_teacache_coefficients(model_name)
    if coefficients is not None and teacache_threshold is not None:
        print(f"Enable TeaCache with threshold: {teacache_threshold}.")
        pipeline.transformer.enable_teacache(num_inference_steps, teacache_threshold, coefficients=coefficients)
    return pipeline


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default=None, help="v5.1 YAML (written from the built-in key set when omitted)")
    ap.add_argument("--model-dir", required=True)
    ap.add_argument("--make-synthetic", choices=sorted(SYNTHETIC), default=None)
    ap.add_argument("--in-channels", type=int, default=None, help="33 makes the synthetic checkpoint an InP model")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=672)
    ap.add_argument("--video-length", type=int, default=49)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--guidance-scale", type=float, default=6.0)
    ap.add_argument("--seed", type=int, default=43)
    ap.add_argument("--teacache", type=float, default=None)
    ap.add_argument("--fp8-storage", action="store_true")
    ap.add_argument("--embeds", default=None, help=".pt file with prompt_embeds / negative_prompt_embeds")
    ap.add_argument("--save-path", default="samples/easyanimate-videos")
    a = ap.parse_args(argv)
    if a.make_synthetic and not os.path.isdir(os.path.join(a.model_dir, "transformer")):
        make_synthetic_checkpoint(a.model_dir, a.make_synthetic, in_channels=a.in_channels)
    config_path = a.config or write_yaml(os.path.join(a.model_dir, "easyanimate_video_v5.1_magvit_qwen.yaml"))
    pipeline = build_pipeline(config_path, a.model_dir, teacache_threshold=a.teacache, num_inference_steps=a.steps,
                              fp8_storage=a.fp8_storage)
    if a.embeds:
        e = torch.load(a.embeds)
        pos, neg = e["prompt_embeds"], e["negative_prompt_embeds"]
    else:
        pos, neg = synthetic_embeddings(pipeline.transformer.config.text_embed_dim)
    generator = torch.Generator(device="cpu").manual_seed(a.seed)
    with torch.no_grad():
        sample = pipeline(None, video_length=a.video_length, height=a.height, width=a.width, generator=generator,
                          guidance_scale=a.guidance_scale, num_inference_steps=a.steps, prompt_embeds=pos,
                          negative_prompt_embeds=neg, output_type="np").frames
    os.makedirs(a.save_path, exist_ok=True)
    prefix = str(len(os.listdir(a.save_path)) + 1).zfill(8)
    sample = np.asarray(sample)
    np.save(os.path.join(a.save_path, prefix + ".npy"), sample)
    if a.video_length == 1:
        try:
            from PIL import Image
            Image.fromarray((sample[0, :, 0].transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(a.save_path, prefix + ".png"))
        except ImportError:
            pass
    print(f"frames {sample.shape} in [{sample.min():.3f}, {sample.max():.3f}] -> {os.path.join(a.save_path, prefix)}.*")
    return sample


if __name__ == "__main__":
    main()
