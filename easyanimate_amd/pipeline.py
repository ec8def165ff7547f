"""EasyAnimatePipeline / EasyAnimateInpaintPipeline: the denoise loop and latent plumbing of
/root/reference/easyanimate/pipeline/pipeline_easyanimate.py:175-1149 and pipeline_easyanimate_inpaint.py
(constructor slots, __call__ keywords and `.frames` output kept; text encoding is out of scope -- callers pass
prompt_embeds / negative_prompt_embeds, SURVEY.md section 7 "hard parts")."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from . import ops
from . import _params
from ._params import invalidate_weight_cache
from .embeddings import get_3d_rotary_pos_embed, get_resize_crop_region_for_grid
from .scheduler import FlowMatchEulerDiscreteScheduler


@dataclass
class EasyAnimatePipelineOutput:
    frames: Union[torch.Tensor, np.ndarray]


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers randn_tensor: sample on the generator's device, then move (CPU generator + GPU target => CPU sample)."""
    gen_device = generator.device if generator is not None else torch.device(device or "cpu")
    return torch.randn(shape, generator=generator, device=gen_device, dtype=dtype).to(device or gen_device)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    if timesteps is not None or sigmas is not None:
        raise NotImplementedError("custom timesteps/sigmas")
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps


def get_image_to_video_latent(image: torch.Tensor, video_length: int):
    """reference: easyanimate/utils/utils.py:128-157 (single start image, already a [3,H,W] tensor in [0,1]):
    the image tiled over all frames, mask 0 on frame 0 and 255 elsewhere."""
    input_video = torch.tile(image[None, :, None], [1, 1, video_length, 1, 1]).to(torch.float32)
    mask = torch.zeros_like(input_video[:, :1])
    mask[:, :, 1:] = 255
    return input_video, mask


def resize_mask(mask, latent, process_first_frame_only=True):
    """reference: pipeline_easyanimate_inpaint.py:116-149 (trilinear resize, first frame separately)."""
    import torch.nn.functional as F
    latent_size = latent.size()
    if process_first_frame_only:
        target_size = list(latent_size[2:])
        target_size[0] = 1
        first = F.interpolate(mask[:, :, 0:1, :, :], size=target_size, mode="trilinear", align_corners=False)
        target_size = list(latent_size[2:])
        target_size[0] = target_size[0] - 1
        if target_size[0] != 0:
            rest = F.interpolate(mask[:, :, 1:, :, :], size=target_size, mode="trilinear", align_corners=False)
            return torch.cat([first, rest], dim=2)
        return first
    return F.interpolate(mask, size=list(latent_size[2:]), mode="trilinear", align_corners=False)


class EasyAnimatePipeline:
    """Text-to-video sampling loop (reference: pipeline_easyanimate.py:175, __call__ :769-1149).

    `latents_fp32` (default True): the loop keeps an fp32 master copy of the latents (0.87 MB at 49 x 1024^2) and hands
    the transformer fp32 latents and the un-rounded fp32 timestep; the bf16 model weights / activations are unchanged.
    The reference's bf16 path stores the latents in bf16 after every Euler update and rounds the timestep to bf16
    (pipeline_easyanimate.py:1079-1081,1111): over a 50-step CFG-6 schedule that storage rounding alone puts the
    reference's OWN bf16 run 6e-4 (MSE) away from its fp32 run (tests/golden/denoise_loop_50*.pt), six times the 1e-4
    parity bar, while a bf16 model stepping fp32 latents stays at 3e-5.  `latents_fp32 = False` reproduces the
    reference's bf16 bookkeeping exactly."""
    latents_fp32 = True

    def __init__(self, vae=None, text_encoder=None, tokenizer=None, text_encoder_2=None, tokenizer_2=None,
                 transformer=None, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.transformer, self.scheduler = transformer, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self._interrupt = False
        self._guidance_scale = 1.0
        self._guidance_rescale = 0.0
        self._num_timesteps = 0

    # -- properties mirrored from the reference ------------------------------------------------
    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def guidance_rescale(self):
        return self._guidance_rescale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def device(self):
        return self.transformer.device

    def to(self, device):
        self.transformer.to(device)
        if self.vae is not None:
            self.vae.to(device)
        return self

    @property
    def compute_dtype(self):
        """dtype of latents / prompt embeddings / conditioning.  The reference takes it from the text encoder (bf16,
        pipeline_easyanimate.py:938-944); with the transformer stored as float8_e4m3fn (predict_t2v.py's default
        GPU_memory_mode) `transformer.dtype` is a STORAGE type nothing can be sampled or embedded in."""
        for m in (self.text_encoder, self.text_encoder_2):
            dt = getattr(m, "dtype", None)
            if isinstance(dt, torch.dtype) and dt.is_floating_point and dt.itemsize >= 2:
                return dt
        dt = self.transformer.dtype
        if dt.is_floating_point and dt.itemsize >= 2:
            return dt
        vdt = getattr(self.vae, "dtype", None)
        return vdt if isinstance(vdt, torch.dtype) and vdt.itemsize >= 2 else torch.bfloat16

    # -- diffusers DiffusionPipeline offload entry points (every GPU_memory_mode branch of predict_t2v.py:256-273 calls one) --
    def _resident(self, gpu_id=None, device=None):
        if isinstance(device, str) or device is None:
            device = torch.device(device or "cuda")
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", gpu_id if gpu_id is not None else torch.cuda.current_device())
        for name in ("transformer", "vae", "text_encoder", "text_encoder_2", "clip_image_encoder"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    def enable_model_cpu_offload(self, gpu_id=None, device=None):
        """diffusers moves one whole model at a time between host and GPU to fit 24-80 GB cards.  An MI355X holds all of
        them at once (23.6 GB of DiT weights, 0.5 GB VAE, the text encoder, against 288 GB), so NOTHING IS OFFLOADED: every
        model is made resident on the device and stays there -- no per-call PCIe traffic, no hooks."""
        return self._resident(gpu_id, device)

    def enable_sequential_cpu_offload(self, gpu_id=None, device=None):
        """See enable_model_cpu_offload: nothing is offloaded on a 288 GB device (diffusers would page every sub-module
        in and out per call)."""
        return self._resident(gpu_id, device)

    def maybe_free_model_hooks(self):
        """diffusers re-arms its offload hooks here at the end of __call__; there are none."""
        return None

    def enable_vae_slicing(self):
        return None

    def enable_vae_tiling(self):
        """The reference's tiled VAE (autoencoder_magvit.py:339-448) blends overlapping tiles, so the result differs from the
        whole-clip one; never needed for memory here (49 x 1024^2 fits in 83 GB), available for checkpoints / callers that ask."""
        self.vae.use_tiling = True

    def disable_vae_tiling(self):
        self.vae.use_tiling = self.vae.use_tiling_encoder = self.vae.use_tiling_decoder = False

    # -- helpers --------------------------------------------------------------------------------
    def latent_shape(self, batch_size, num_channels_latents, video_length, height, width):
        """reference: prepare_latents :677-690 (3-D VAE with cache_mag_vae)."""
        mbe = getattr(self.vae, "mini_batch_encoder", 4) if self.vae is not None else 4
        mbd = getattr(self.vae, "mini_batch_decoder", 1) if self.vae is not None else 1
        cache = getattr(self.vae, "cache_mag_vae", True) if self.vae is not None else True
        if video_length == 1:
            f = 1
        elif cache:
            f = int((video_length - 1) // mbe * mbd + 1)
        else:
            f = int(video_length // mbe * mbd)
        return (batch_size, num_channels_latents, f, height // self.vae_scale_factor, width // self.vae_scale_factor)

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        shape = self.latent_shape(batch_size, num_channels_latents, video_length, height, width)
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents

    def rotary_embedding(self, height, width, latent_frames):
        """reference: :999-1011"""
        p = self.transformer.config.patch_size
        gh, gw = height // 8 // p, width // 8 // p
        base_w, base_h = 720 // 8 // p, 480 // 8 // p
        cc = get_resize_crop_region_for_grid((gh, gw), base_w, base_h)
        cos, sin = get_3d_rotary_pos_embed(self.transformer.config.attention_head_dim, cc, grid_size=(gh, gw),
                                           temporal_size=latent_frames, use_real=True)
        dev = self.transformer.device
        if dev.type == "cuda":
            # resident for the whole call (the reference re-uploads them in every attention call); a sequence-parallel
            # rank slices its shard out of the device tables (a view)
            cos, sin = cos.to(dev), sin.to(dev)
        return cos, sin

    def decode_latents(self, latents):
        """reference: :722-742.  The 1/scaling_factor multiply touches the 0.9 MB latent only; the clamp / rescale
        / clamp of the 150 M-pixel video is fused into the VAE's final layout kernel (postprocess=True)."""
        latents = (1 / self.vae.config.scaling_factor) * latents
        video = self.vae.decode(latents, postprocess=True)[0]
        return video.cpu().float().numpy()

    def encode_prompt(self, prompt, device, dtype, num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                      negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None, prompt_attention_mask=None,
                      negative_prompt_attention_mask=None, max_sequence_length: Optional[int] = None,
                      text_encoder_index: int = 0, actual_max_sequence_length: int = 256):
        """reference: pipeline_easyanimate.py:306-580.  The glue around the text encoder -- tokenisation (BERT / T5 tokenizers
        directly, anything else through its chat template), the encoder call (last hidden state for BERT / T5, the
        PENULTIMATE hidden state of the LLM, :438-447), repetition per image, the same for the negative prompt.  The encoder
        itself is whatever `transformers` model sits in the pipeline's text_encoder slot (Qwen2-VL-7B for V5.1): it runs
        once per call and is not part of this build's kernels.
        -> (prompt_embeds, negative_prompt_embeds, prompt_attention_mask, negative_prompt_attention_mask)"""
        tokenizer = [self.tokenizer, self.tokenizer_2][text_encoder_index]
        text_encoder = [self.text_encoder, self.text_encoder_2][text_encoder_index]
        if max_sequence_length is None:
            max_length = min(tokenizer.model_max_length, actual_max_sequence_length)
        else:
            max_length = max_sequence_length
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        use_mask = bool(self.transformer.config.get("enable_text_attention_mask", True))

        def classic(tok_cls_names=("BertTokenizer", "T5Tokenizer")):
            return type(tokenizer).__name__ in tok_cls_names

        def run_classic(texts, length, with_retry_mask):
            ti = tokenizer(texts, padding="max_length", max_length=length, truncation=True, return_attention_mask=True,
                           return_tensors="pt")
            ids = ti.input_ids
            if ids.shape[-1] > actual_max_sequence_length:
                re = tokenizer.batch_decode(ids[:, :actual_max_sequence_length], skip_special_tokens=True)
                ti = tokenizer(re, padding="max_length", max_length=length, truncation=True, return_attention_mask=True,
                               return_tensors="pt")
                ids = ti.input_ids
            mask = ti.attention_mask.to(device)
            out = text_encoder(ids.to(device), attention_mask=mask) if use_mask else text_encoder(ids.to(device))
            return out[0], mask

        def run_llm(p):
            if p is not None and isinstance(p, str):
                messages = [{"role": "user", "content": [{"type": "text", "text": p}]}]
            else:
                messages = [{"role": "user", "content": [{"type": "text", "text": _p}]} for _p in p]
            text = tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
            ti = tokenizer(text=[text], padding="max_length", max_length=max_length, truncation=True, return_attention_mask=True,
                           padding_side="right", return_tensors="pt")
            ti = ti.to(text_encoder.device)
            if not use_mask:
                raise ValueError("LLM needs attention_mask")
            hs = text_encoder(input_ids=ti.input_ids, attention_mask=ti.attention_mask, output_hidden_states=True).hidden_states[-2]
            return hs, ti.attention_mask

        if prompt_embeds is None:
            prompt_embeds, prompt_attention_mask = run_classic(prompt, max_length, True) if classic() else run_llm(prompt)
            prompt_attention_mask = prompt_attention_mask.repeat(num_images_per_prompt, 1)
        prompt_embeds = prompt_embeds.to(dtype=dtype, device=device)
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt, seq_len, -1)
        prompt_attention_mask = prompt_attention_mask.to(device=device)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if classic():
                if negative_prompt is None:
                    uncond = [""] * batch_size
                elif prompt is not None and type(prompt) is not type(negative_prompt):
                    raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
                elif isinstance(negative_prompt, str):
                    uncond = [negative_prompt]
                elif batch_size != len(negative_prompt):
                    raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size {batch_size}.")
                else:
                    uncond = negative_prompt
                negative_prompt_embeds, negative_prompt_attention_mask = run_classic(uncond, prompt_embeds.shape[1], False)
            else:
                negative_prompt_embeds, negative_prompt_attention_mask = run_llm(negative_prompt)
            negative_prompt_attention_mask = negative_prompt_attention_mask.repeat(num_images_per_prompt, 1)
        if do_classifier_free_guidance:
            seq_len = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.to(dtype=dtype, device=device)
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(batch_size * num_images_per_prompt, seq_len, -1)
            negative_prompt_attention_mask = negative_prompt_attention_mask.to(device=device)
        return prompt_embeds, negative_prompt_embeds, prompt_attention_mask, negative_prompt_attention_mask

    def _embeds(self, prompt_embeds, negative_prompt_embeds, device, dtype, prompt=None, negative_prompt=None, index: int = 0):
        """[negative | positive] embeddings of encoder `index` (:920-1056): from the given embeddings, or -- when the
        pipeline holds that tokenizer / text encoder -- from the prompts through encode_prompt."""
        tok = [self.tokenizer, self.tokenizer_2][index]
        enc = [self.text_encoder, self.text_encoder_2][index]
        if prompt_embeds is None and prompt is not None and tok is not None and enc is not None:
            prompt_embeds, negative_prompt_embeds, _, _ = self.encode_prompt(
                prompt, device, dtype, 1, self.do_classifier_free_guidance, negative_prompt, text_encoder_index=index)
        if prompt_embeds is None:
            raise NotImplementedError(
                "no prompt embeddings and no text encoder in the pipeline: pass prompt_embeds / negative_prompt_embeds, or put a "
                "transformers text encoder + tokenizer into the text_encoder / tokenizer slots (SURVEY.md section 2 row 8)")
        pe = prompt_embeds.to(device=device, dtype=dtype)
        if self.do_classifier_free_guidance:
            if negative_prompt_embeds is None:
                raise ValueError("negative_prompt_embeds is required when guidance_scale > 1")
            pe = torch.cat([negative_prompt_embeds.to(device=device, dtype=dtype), pe])
        return pe

    def denoise(self, latents, prompt_embeds, image_rotary_emb, timesteps, guidance_scale, inpaint_latents=None,
                prompt_embeds_2=None, callback_on_step_end=None, guidance_rescale: float = 0.0):
        """The hot loop (reference :1069-1134): CFG duplicate, bf16 timestep, transformer, CFG combine + Euler
        update fused in one kernel.  No host synchronisation inside the loop."""
        do_cfg = guidance_scale > 1
        out_dtype = latents.dtype
        if self.latents_fp32 and latents.dtype != torch.float32:
            latents = latents.float()
        # no parameter is written inside the loop: the derived K-blocked ff.net.2 weights are re-derived once here (a `.data` LoRA
        # merge since the last call is invisible to the version counter), not once per forward (_params.weights_frozen)
        _params.drop_tag("kblock")
        with _params.weights_frozen():
            for i, t in enumerate(timesteps):
                if self._interrupt:
                    continue
                latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
                t_expand = t.reshape(1).expand(latent_model_input.shape[0]).to(dtype=latent_model_input.dtype)
                noise_pred = self.transformer(
                    latent_model_input, t_expand, encoder_hidden_states=prompt_embeds,
                    encoder_hidden_states_t5=prompt_embeds_2, image_rotary_emb=image_rotary_emb,
                    inpaint_latents=inpaint_latents, return_dict=False)[0]
                if noise_pred.size(1) != latents.size(1):
                    noise_pred, _ = noise_pred.chunk(2, dim=1)
                    noise_pred = noise_pred.contiguous()
                latents = self.scheduler.step(noise_pred, t, latents, return_dict=False,
                                              guidance_scale=guidance_scale if do_cfg else None,
                                              guidance_rescale=guidance_rescale if do_cfg else 0.0)[0]
                if callback_on_step_end is not None:
                    out = callback_on_step_end(self, i, t, {"latents": latents})
                    latents = out.pop("latents", latents) if out else latents
        return latents.to(out_dtype)

    @torch.no_grad()
    def __call__(self, prompt=None, video_length: Optional[int] = None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, prompt_embeds=None, prompt_embeds_2=None,
                 negative_prompt_embeds=None, negative_prompt_embeds_2=None, prompt_attention_mask=None,
                 prompt_attention_mask_2=None, negative_prompt_attention_mask=None,
                 negative_prompt_attention_mask_2=None, output_type: str = "latent", return_dict: bool = True,
                 callback_on_step_end: Optional[Callable] = None, callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 guidance_rescale: float = 0.0, original_size=None, target_size=None, crops_coords_top_left=(0, 0),
                 clip_image=None, clip_apply_ratio=0.40, comfyui_progressbar=False, timesteps=None):
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt > 1 (one video per call)")
        height = int(height // 16 * 16)
        width = int(width // 16 * 16)
        self._guidance_scale = guidance_scale
        self._guidance_rescale = guidance_rescale
        self._interrupt = False
        # derived weight copies (fp32 biases, packed conv / patch-embedding weights) are rebuilt once per call, so writes the
        # version counter cannot see -- merge_lora / unmerge_lora do `weight.data += ...` between calls
        # (predict_t2v.py:282,319; utils/lora_utils.py:369-433) -- are always observed; costs a few ms per 50-step call
        invalidate_weight_cache()
        device = self.transformer.device
        dtype = self.compute_dtype
        pe = self._embeds(prompt_embeds, negative_prompt_embeds, device, dtype, prompt, negative_prompt, 0)
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, mu=1)
        self._num_timesteps = len(timesteps)
        nc = self.transformer.config.in_channels
        latents = self.prepare_latents(1 * num_images_per_prompt, nc, video_length, height, width, dtype, device,
                                       generator, latents)
        rope = self.rotary_embedding(height, width, latents.size(2))
        pe2 = None
        if prompt_embeds_2 is not None or (prompt is not None and self.tokenizer_2 is not None):   # V5 two-encoder checkpoints
            pe2 = self._embeds(prompt_embeds_2, negative_prompt_embeds_2, device, dtype, prompt, negative_prompt, 1)
        latents = self.denoise(latents, pe, rope, timesteps, guidance_scale, prompt_embeds_2=pe2,
                               callback_on_step_end=callback_on_step_end, guidance_rescale=guidance_rescale)
        return self._output(latents, output_type, return_dict)

    def _output(self, latents, output_type, return_dict):
        """reference :1136-1149 (both pipelines): always decode; output_type "latent" means "a torch tensor" (of the decoded
        video), anything else the numpy array.  Without a VAE (benchmarks / tests of the loop alone) the latents come back."""
        if self.vae is None:
            video = latents
        else:
            video = self.decode_latents(latents)
            if output_type == "latent":
                video = torch.from_numpy(video)
        if not return_dict:
            return video
        return EasyAnimatePipelineOutput(frames=video)


def add_noise_to_reference_video(image, ratio=None, generator=None):
    """reference: pipeline_easyanimate_inpaint.py:153-167 (noise augmentation of the conditioning video, pixels that are
    exactly -1 -- the masked ones -- stay untouched).  Input preparation on the conditioning video, like randn_tensor."""
    if ratio is None:
        sigma = torch.exp(torch.normal(mean=-3.0, std=0.5, size=(image.shape[0],)).to(image.device)).to(image.dtype)
    else:
        sigma = torch.ones((image.shape[0],)).to(image.device, image.dtype) * ratio
    if generator is not None:
        noise = torch.randn(image.size(), generator=generator, dtype=image.dtype, device=generator.device).to(image.device)
    else:
        noise = torch.randn_like(image)
    noise = noise * sigma[:, None, None, None, None]
    noise = torch.where(image == -1, torch.zeros_like(image), noise)
    return image + noise


class EasyAnimateInpaintPipeline(EasyAnimatePipeline):
    """I2V / inpaint / V2V variant (reference: pipeline_easyanimate_inpaint.py:978-1606): VAE-encode the masked video,
    bring the mask to latent resolution, feed cat[mask, masked_latents] as `inpaint_latents` (1 + 16 channels with
    resize_inpaint_mask_directly, 16 + 16 with the VAE-encoded mask)."""

    def __init__(self, vae=None, text_encoder=None, tokenizer=None, text_encoder_2=None, tokenizer_2=None, transformer=None,
                 scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None, clip_image_encoder=None, clip_image_processor=None):
        super().__init__(vae, text_encoder, tokenizer, text_encoder_2, tokenizer_2, transformer, scheduler)
        self.clip_image_encoder, self.clip_image_processor = clip_image_encoder, clip_image_processor

    def _encode(self, x, dtype, device):
        """.mode() * scaling_factor of the VAE posterior (:777-789, :797-810, :868-878)"""
        lat = self.vae.encode(x.to(device=device, dtype=self.vae.dtype))[0].mode()
        return lat.to(dtype) * self.vae.config.scaling_factor

    def prepare_mask_latents(self, mask, masked_image, dtype, device, generator=None, noise_aug_strength=None):
        """reference: :769-826 -> (mask latents or None, masked-video latents or None)"""
        mask_lat = self._encode(mask, dtype, device) if mask is not None else None
        masked_lat = None
        if masked_image is not None:
            masked_image = masked_image.to(device=device, dtype=dtype)
            if self.transformer.config.get("add_noise_in_inpaint_model", False):
                masked_image = add_noise_to_reference_video(masked_image, ratio=noise_aug_strength, generator=generator)
            masked_lat = self._encode(masked_image, dtype, device)
        return mask_lat, masked_lat

    @staticmethod
    def preprocess_video(video: torch.Tensor, height=None, width=None, normalize: bool = True) -> torch.Tensor:
        """diffusers VaeImageProcessor.preprocess on a [B,C,F,H,W] tensor (called per frame at :1231-1233, :1340): nearest
        resize to (height, width) when the size differs, then [0,1] -> [-1,1] -- skipped, like diffusers does, when the
        input already has negative values."""
        video = video.to(torch.float32)
        if height is not None and tuple(video.shape[-2:]) != (height, width):
            import torch.nn.functional as F
            b, c, f = video.shape[:3]
            frames = video.permute(0, 2, 1, 3, 4).reshape(b * f, c, *video.shape[-2:])
            frames = F.interpolate(frames, size=(height, width))
            video = frames.reshape(b, f, c, height, width).permute(0, 2, 1, 3, 4)
        if normalize and video.numel() and float(video.min()) >= 0:
            video = video * 2.0 - 1.0
        return video

    @classmethod
    def masked_video_and_mask(cls, video: torch.Tensor, mask_video: torch.Tensor, height=None, width=None):
        """Host-side preprocessing exactly as the reference does it on CPU tensors (:1337-1346):
        video in [0,1] -> init_video in [-1,1] (VaeImageProcessor normalize); mask in [0,255] binarised at 0.5;
        masked_video = init_video * (mask < 0.5) - (mask > 0.5).  Both are first brought to (height, width)."""
        init_video = cls.preprocess_video(video, height, width)
        mask_condition = (cls.preprocess_video(mask_video, height, width, normalize=False) >= 0.5).to(torch.float32)
        tile = torch.tile(mask_condition, [1, 3, 1, 1, 1])
        masked_video = init_video * (tile < 0.5) + torch.ones_like(init_video) * (tile > 0.5) * -1
        return masked_video, mask_condition

    def inpaint_conditioning(self, video, mask_video, dtype, device, do_cfg=True, latent_shape=None, generator=None,
                             noise_aug_strength=0.0563, masked_video_latents=None, height=None, width=None):
        """reference: pipeline_easyanimate_inpaint.py:1321-1383."""
        direct = self.transformer.resize_inpaint_mask_directly
        if mask_video is None or (self.transformer.config.get("enable_zero_in_inpaint", True) and bool((mask_video == 255).all())):
            # nothing is kept from the input video (plain T2V through an InP checkpoint): zero conditioning (:1322-1336)
            shp = tuple(latent_shape)
            mask_latents = torch.zeros((shp[0], 1 if direct else shp[1]) + shp[2:], dtype=dtype, device=device)
            masked_latents = torch.zeros(shp, dtype=dtype, device=device)
        else:
            masked_video, mask_condition = self.masked_video_and_mask(video.cpu(), mask_video.cpu(), height, width)
            if masked_video_latents is not None:
                masked_video = masked_video_latents
            if direct:
                _, masked_latents = self.prepare_mask_latents(None, masked_video, dtype, device, generator, noise_aug_strength)
                mask_latents = resize_mask(1 - mask_condition, masked_latents, getattr(self.vae, "cache_mag_vae", True))
                mask_latents = mask_latents.to(device, dtype) * self.vae.config.scaling_factor
            else:
                mask_latents, masked_latents = self.prepare_mask_latents(torch.tile(mask_condition, [1, 3, 1, 1, 1]), masked_video,
                                                                         dtype, device, generator, noise_aug_strength)
        inpaint = torch.cat([mask_latents, masked_latents], dim=1).to(dtype)
        return torch.cat([inpaint] * 2) if do_cfg else inpaint

    def get_timesteps(self, num_inference_steps, strength, device):
        """reference: :760-767"""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    @torch.no_grad()
    def __call__(self, prompt=None, video_length=None, video=None, mask_video=None, masked_video_latents=None,
                 height=None, width=None, num_inference_steps: int = 50, guidance_scale: float = 5.0, negative_prompt=None,
                 num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
                 prompt_embeds_2=None, negative_prompt_embeds=None, negative_prompt_embeds_2=None, prompt_attention_mask=None,
                 prompt_attention_mask_2=None, negative_prompt_attention_mask=None, negative_prompt_attention_mask_2=None,
                 output_type: str = "latent", return_dict: bool = True, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], guidance_rescale: float = 0.0,
                 original_size=(1024, 1024), target_size=None, crops_coords_top_left=(0, 0), clip_image=None,
                 clip_apply_ratio=0.40, strength: float = 1.0, noise_aug_strength: float = 0.0563, comfyui_progressbar=False,
                 timesteps=None):
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt > 1 (one video per call)")
        nc_ = self.vae.config.latent_channels if self.vae is not None else 16
        if self.transformer.config.get("enable_clip_in_inpaint", True) and self.transformer.config.in_channels != nc_:
            # pipeline_easyanimate_inpaint.py:1270-1311: an inpaint checkpoint with enable_clip_in_inpaint feeds CLIP tokens of
            # clip_image -- or ZERO tokens when there is none -- through clip_proj into every forward.  The V5 / V5.1 YAMLs set the
            # flag false; with it true the reference's own V5.1 transformer cannot run that branch either (transformer3d.py:1558-1561
            # concatenates the CLIP tokens with ref_latents, which this pipeline never passes).  Skipping the tokens silently
            # would be a different model: refuse.
            raise NotImplementedError("enable_clip_in_inpaint=True on an inpaint transformer: the CLIP-token conditioning of "
                                      "pipeline_easyanimate_inpaint.py:1270-1311 (clip_image, or zero tokens without one) is not part of "
                                      "the V5 / V5.1 path (their YAMLs set enable_clip_in_inpaint: false); set it false in the transformer "
                                      "config to run this checkpoint without CLIP conditioning")
        height = int(height // 16 * 16)
        width = int(width // 16 * 16)
        self._guidance_scale = guidance_scale
        self._guidance_rescale = guidance_rescale
        self._interrupt = False
        # derived weight copies (fp32 biases, packed conv / patch-embedding weights) are rebuilt once per call, so writes the
        # version counter cannot see -- merge_lora / unmerge_lora do `weight.data += ...` between calls
        # (predict_t2v.py:282,319; utils/lora_utils.py:369-433) -- are always observed; costs a few ms per 50-step call
        invalidate_weight_cache()
        device = self.transformer.device
        dtype = self.compute_dtype
        pe = self._embeds(prompt_embeds, negative_prompt_embeds, device, dtype, prompt, negative_prompt, 0)
        pe2 = None
        if prompt_embeds_2 is not None or (prompt is not None and self.tokenizer_2 is not None):
            pe2 = self._embeds(prompt_embeds_2, negative_prompt_embeds_2, device, dtype, prompt, negative_prompt, 1)
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, mu=1)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, device)
        self._num_timesteps = len(timesteps)
        if video is not None:
            video_length = video.shape[2]
        nc = self.vae.config.latent_channels if self.vae is not None else 16
        noise_or_latents = self.prepare_latents(1, nc, video_length, height, width, dtype, device, generator, latents)
        if strength < 1.0 and latents is None:
            # V2V re-noising (:862-893): start from the encoded input video at the first kept timestep
            if video is None:
                raise ValueError("strength < 1 needs `video`")
            video_latents = self._encode(self.preprocess_video(video.cpu(), height, width), dtype, device)
            latents = self.scheduler.scale_noise(video_latents, timesteps[:1], noise_or_latents)
        else:
            latents = noise_or_latents
        rope = self.rotary_embedding(height, width, latents.size(2))
        inpaint = None
        if self.transformer.config.in_channels != nc:
            inpaint = self.inpaint_conditioning(video, mask_video, dtype, device, self.do_classifier_free_guidance,
                                                latent_shape=latents.shape, generator=generator,
                                                noise_aug_strength=noise_aug_strength, masked_video_latents=masked_video_latents,
                                                height=height, width=width)
            if inpaint.shape[1] + nc != self.transformer.config.in_channels:
                raise ValueError(f"the transformer expects {self.transformer.config.in_channels} input channels, latents + "
                                 f"inpaint conditioning give {nc + inpaint.shape[1]} (resize_inpaint_mask_directly mismatch?)")
        elif video is not None and mask_video is not None:
            raise NotImplementedError("latent-space blending for a T2V checkpoint inside the inpaint pipeline (the reference's own "
                                      "branch, :1562-1575, does not run: scale_noise is called with a mis-placed parenthesis)")
        latents = self.denoise(latents, pe, rope, timesteps, guidance_scale, inpaint_latents=inpaint, prompt_embeds_2=pe2,
                               callback_on_step_end=callback_on_step_end, guidance_rescale=guidance_rescale)
        return self._output(latents, output_type, return_dict)
