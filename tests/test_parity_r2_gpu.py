"""Parity at the STATED bar on STATED configurations (VERDICT r1 "next round" item 1), all through the C ABI:

  (a) the 50-step CFG-6 sampling loop: latent MSE < 1e-4 against the reference's fp32 run, no floor term;
  (b) the full-width VAE (128/256/512/512, mid-block head_dim 512) at 9 x 256^2 against the reference's chunked /
      cached run, with the dispatch counters proving that the row-slab and 256x256 ping-pong kernels served it;
  (c) full-width (d = 3072) two-layer transformer forwards, 16- and 33-channel;
  (d) the V5 two-encoder text path (text_proj_t5) and control_latents.

Every golden file was produced by oracle/gen_golden.py executing the unchanged reference modules; large inputs are
regenerated from seeds with the generator functions of that script (checksums stored in the fixture)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
BAR = 1e-4   # BASELINE.json north_star: "latent MSE vs reference < 1e-4"


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def _mse(a, b):
    return ((a.double().cpu() - b.double().cpu()) ** 2).mean().item()


def _model(cfg, shapes, seed, style):
    from easyanimate_amd import EasyAnimateTransformer3DModel
    from easyanimate_amd.synthetic import synth_state_dict
    m = EasyAnimateTransformer3DModel.from_config(cfg)
    m.load_state_dict(synth_state_dict(shapes, seed, style), strict=True)
    return m.to(torch.bfloat16).to(DEV).eval()


# ---------------------------------------------------------------------------------------------------------
# (a) 50 Flow steps, CFG 6 (pipeline_easyanimate.py:1069-1111) through EasyAnimatePipeline.denoise
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["denoise_loop_50", "denoise_loop_50_default", "denoise_loop_50_bf16in",
                                  "denoise_loop_50_default_bf16in"])
def test_denoise_loop_50_steps_meets_the_bar(name):
    from easyanimate_amd import EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler
    g = _load(name + ".pt")
    m = _model(g["cfg"], g["shapes"], g["seed"], g["style"])
    ref, refb = g["trace"], g["trace_bf16"]
    enc = g["enc"].to(DEV).bfloat16()
    res = {}
    for fp32_latents in (True, False):
        sched = FlowMatchEulerDiscreteScheduler(shift=1.0)
        sched.set_timesteps(g["steps"], device=DEV, mu=1)
        pipe = EasyAnimatePipeline(vae=None, transformer=m, scheduler=sched)
        pipe.latents_fp32 = fp32_latents
        kept = {}

        def keep(p, i, t, kw, _k=kept):
            if (i + 1) in ref:
                _k[i + 1] = kw["latents"].float().cpu().clone()
            return {}
        with torch.no_grad():
            x = pipe.denoise(g["latents"].to(DEV).bfloat16(), enc, (g["cos"], g["sin"]), sched.timesteps, g["guidance"],
                             callback_on_step_end=keep)
        assert x.dtype == torch.bfloat16 and sched.step_index == g["steps"]
        res[fp32_latents] = {k: _mse(v, ref[k]) for k, v in kept.items()}
        floor = {k: _mse(refb[k], ref[k]) for k in ref}
        vs_b = {k: _mse(v, refb[k]) for k, v in kept.items()}
        mode = "fp32 master latents (default)" if fp32_latents else "bf16 latents (reference bf16 bookkeeping)"
        print(f"[parity] {name} [{mode}] latent MSE by step: new vs ref-fp32 "
              + ", ".join(f"{k}: {v:.3e}" for k, v in res[fp32_latents].items())
              + " | ref-bf16 vs ref-fp32 (floor) " + ", ".join(f"{k}: {v:.3e}" for k, v in floor.items())
              + " | new vs ref-bf16 " + ", ".join(f"{k}: {v:.3e}" for k, v in vs_b.items()))
    # the bar, with NO floor term, on the default path
    assert res[True][g["steps"]] < BAR, res[True]
    assert all(v < BAR for v in res[True].values())
    # the reference-bookkeeping mode is held to the reference's own bf16 distance (it cannot do better than that)
    assert res[False][g["steps"]] <= 1.5 * _mse(refb[g["steps"]], ref[g["steps"]])


# ---------------------------------------------------------------------------------------------------------
# (d) text_proj_t5 and control_latents (transformer3d.py:1523-1526, 1533-1536)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["transformer_t5", "transformer_t5_norm", "transformer_control", "transformer_inp_control"])
def test_transformer_branches_vs_golden(name):
    g = _load(name + ".pt")
    m = _model(g["cfg"], g["shapes"], g["seed"], g["style"])
    dev = lambda x: None if x is None else x.to(DEV).bfloat16()
    with torch.no_grad():
        out = m(dev(g["latents"]), dev(g["t"]), encoder_hidden_states=dev(g["enc"]), encoder_hidden_states_t5=dev(g["enc_t5"]),
                image_rotary_emb=(g["cos"], g["sin"]), inpaint_latents=dev(g["inpaint"]), control_latents=dev(g["control"]),
                return_dict=False)[0]
    mse, floor = _mse(out.float(), g["out"]), _mse(g["out_bf16"], g["out"])
    print(f"[parity] {name}: new-bf16 vs ref-fp32 MSE={mse:.3e} | ref-bf16 floor {floor:.3e} | new vs ref-bf16 "
          f"{_mse(out.float(), g['out_bf16']):.3e} | ref std {g['out'].std().item():.3f}")
    assert out.shape == g["out"].shape
    assert mse < BAR


@pytest.mark.parametrize("name", ["transformer_after_norm", "transformer_ref_clip", "transformer_ref"])
def test_transformer_after_norm_ref_clip_vs_golden(name):
    """after_norm blocks (FFN -> FP32LayerNorm -> gated residual) and the ref-latent / CLIP conditioning
    that replaces the text stream (transformer3d.py:1538-1561), against the unchanged reference."""
    g = _load(name + ".pt")
    m = _model(g["cfg"], g["shapes"], g["seed"], g["style"])
    dev = lambda x: None if x is None else x.to(DEV).bfloat16()
    with torch.no_grad():
        out = m(dev(g["latents"]), dev(g["t"]), encoder_hidden_states=dev(g["enc"]), image_rotary_emb=(g["cos"], g["sin"]),
                control_latents=dev(g["control"]), ref_latents=dev(g["ref"]), clip_encoder_hidden_states=dev(g["clip"]),
                return_dict=False)[0]
    mse, floor = _mse(out.float(), g["out"]), _mse(g["out_bf16"], g["out"])
    print(f"[parity] {name}: new-bf16 vs ref-fp32 MSE={mse:.3e} | ref-bf16 floor {floor:.3e} | new vs ref-bf16 "
          f"{_mse(out.float(), g['out_bf16']):.3e} | ref std {g['out'].std().item():.3f}")
    assert out.shape == g["out"].shape and mse < BAR


def test_fp8_weight_storage_vs_reference_wrappers():
    """SURVEY 8f rank 2, as a PARITY test: the reference's default memory mode stores every parameter as
    float8_e4m3fn and up-casts per call (predict_t2v.py:37,104,266; utils/fp8_optimization.py:17-35).  The golden holds the
    reference's own wrappers run on the CPU (bf16 compute) and fp32 compute on the same fp8-representable values; the product
    model is put into the same storage mode (bf16 checkpoint values -> .to(float8_e4m3fn), as from_pretrained_2d does)."""
    g = _load("transformer_fp8_storage.pt")
    m = _model(g["cfg"], g["shapes"], g["seed"], g["style"])
    for p_ in m.parameters():
        p_.data = p_.data.to(torch.float8_e4m3fn)              # convert_model_weight_to_float8
    assert m.proj_out.weight.dtype == torch.float8_e4m3fn
    with torch.no_grad():
        out = m(g["latents"].to(DEV).bfloat16(), g["t"].to(DEV).bfloat16(), encoder_hidden_states=g["enc"].to(DEV).bfloat16(),
                image_rotary_emb=(g["cos"], g["sin"]), return_dict=False)[0]
    mse, floor = _mse(out.float(), g["out"]), _mse(g["out_bf16"], g["out"])
    print(f"[parity] fp8 weight storage: new (fp8-stored, bf16 compute) vs reference fp32-on-fp8-values MSE={mse:.3e} | reference bf16 "
          f"through its fp8 wrappers (floor) {floor:.3e} | new vs that {_mse(out.float(), g['out_bf16']):.3e} | the fp8 rounding itself "
          f"moves the output by {_mse(g['out'], g['out_unquantised']):.3e}")
    assert mse < BAR and mse < 0.1 * _mse(g["out"], g["out_unquantised"])


@pytest.mark.parametrize("name", ["transformer_swa", "transformer_swa_mixed", "transformer_swa_long"])
def test_transformer_swa_vs_golden(name):
    """sliding-window attention blocks (swa_layers; processor.py:320-459) -- strided cross keys (interval 2
    here), six scan orders over the head groups, band attention of +-(h*w) positions -- against the reference's own
    processor (flash_attn_func restated, oracle/flash_attn_shim.py)."""
    from easyanimate_amd import _lib
    from easyanimate_amd.embeddings import get_3d_rotary_pos_embed
    from oracle.gen_golden import swa_inputs
    g = _load(name + ".pt")
    B, Fr, H, W, T = g["dims"]
    lat, enc = swa_inputs(g["cfg"], g["input_seed"], *g["dims"])
    assert abs(lat.double().sum().item() - g["lat_sum"]) < 1e-6
    rope = get_3d_rotary_pos_embed(64, g["crops"], grid_size=(H // 2, W // 2), temporal_size=Fr, use_real=True)
    m = _model(g["cfg"], g["shapes"], g["seed"], g["style"])
    _lib.reset_counters()
    with torch.no_grad():
        out = m(lat.to(DEV).bfloat16(), g["t"].to(DEV).bfloat16(), encoder_hidden_states=enc.to(DEV).bfloat16(),
                image_rotary_emb=rope, return_dict=False)[0]
    torch.cuda.synchronize()
    cnt = _lib.counters()
    mse = _mse(out.float(), g["out"].float())
    print(f"[parity] {name}: new-bf16 vs ref-fp32 MSE={mse:.3e} | ref-bf16 floor {g['floor_mse']:.3e} | ref std {g['out_std']:.3f} | kernels {cnt}")
    assert mse < BAR and cnt.get("attention_window_mapped", 0) == len(g["cfg"]["swa_layers"]) == cnt.get("permute_cols", 0)


# ---------------------------------------------------------------------------------------------------------
# (c) full-width two-layer forwards
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["transformer_full_t2v", "transformer_full_inp", "transformer_full_ragged"])
def test_transformer_full_width_vs_golden(name):
    from easyanimate_amd import _lib
    from easyanimate_amd.embeddings import get_3d_rotary_pos_embed
    from oracle.gen_golden import dit_full_inputs
    g = _load(name + ".pt")
    cfg = g["cfg"]
    assert cfg["num_attention_heads"] * cfg["attention_head_dim"] == 3072 and cfg["num_layers"] == 2
    B, Fr, H, W, T = g["dims"]
    lat, extra, enc = dit_full_inputs(cfg, g["input_seed"], *g["dims"])
    assert abs(lat.double().sum().item() - g["lat_sum"]) < 1e-6 and abs(enc.double().sum().item() - g["enc_sum"]) < 1e-5
    rope = get_3d_rotary_pos_embed(64, g["crops"], grid_size=(H // 2, W // 2), temporal_size=Fr, use_real=True)
    m = _model(cfg, g["shapes"], g["seed"], g["style"])
    _lib.reset_counters()
    with torch.no_grad():
        out = m(lat.to(DEV).bfloat16(), g["t"].to(DEV).bfloat16(), encoder_hidden_states=enc.to(DEV).bfloat16(),
                image_rotary_emb=rope, inpaint_latents=None if extra is None else extra.to(DEV).bfloat16(),
                return_dict=False)[0]
    torch.cuda.synchronize()
    cnt = _lib.counters()
    mse = _mse(out.float(), g["out"])
    rel = ((out.float().cpu().double() - g["out"].double()).norm() / g["out"].double().norm()).item()
    print(f"[parity] {name} (d=3072, 2 layers, N={Fr * (H // 2) * (W // 2)}, T={T}): new-bf16 vs ref-fp32 MSE={mse:.3e} "
          f"rel_l2={rel:.3e} ref std {g['out'].std().item():.3f}"
          + (f" | ref-bf16 floor {g['floor_mse']:.3e}" if "floor_mse" in g else "") + f" | kernels {cnt}")
    assert mse < BAR
    assert cnt.get("attention_v3", 0) == 2                             # the product attention kernel, once per block
    if name.endswith("t2v"):
        assert cnt.get("gemm_256_w4a", 0) > 0                           # M = 2 x 5120 rows: the large-tile GEMM path


def test_transformer_full_length_s53504_vs_golden():
    """Config 3's REAL sequence length against the unchanged reference (round 4): full width (d = 3072, 48 heads), 2 layers, latents
    [1,16,13,128,128] = 53 248 video + 256 text tokens (S = 53 504 = 49 f x 1024^2), one batch element; golden = the reference in fp32
    on the host cores (132 s), stored at every second latent pixel.  The one place where the attention kernel's 836-tile online
    softmax, the 208-row-tile GEMM grids and the RoPE tables of the benchmark shape meet a reference OUTPUT (the kernel-level test
    at this length compares with an fp64 softmax of synthetic q / k / v)."""
    from easyanimate_amd import _lib
    from easyanimate_amd.embeddings import get_3d_rotary_pos_embed
    from oracle.gen_golden import dit_full_inputs
    g = _load("transformer_full_length_s53504.pt")
    cfg = g["cfg"]
    B, Fr, H, W, T = g["dims"]
    assert (B, Fr * (H // 2) * (W // 2) + T) == (1, 53504) and cfg["num_attention_heads"] * 64 == 3072
    lat, extra, enc = dit_full_inputs(cfg, g["input_seed"], *g["dims"])
    assert abs(lat.double().sum().item() - g["lat_sum"]) < 1e-5 and abs(enc.double().sum().item() - g["enc_sum"]) < 1e-4
    rope = get_3d_rotary_pos_embed(64, g["crops"], grid_size=(H // 2, W // 2), temporal_size=Fr, use_real=True)
    m = _model(cfg, g["shapes"], g["seed"], g["style"])
    _lib.reset_counters()
    with torch.no_grad():
        out = m(lat.to(DEV).bfloat16(), g["t"].to(DEV).bfloat16(), encoder_hidden_states=enc.to(DEV).bfloat16(), image_rotary_emb=rope,
                return_dict=False)[0]
    torch.cuda.synchronize()
    cnt = _lib.counters()
    assert tuple(out.shape) == tuple(g["out_shape"])
    ref = g["out_sub_f16"].double()
    d = out[..., ::2, ::2].float().cpu().double() - ref
    mse, rel = (d ** 2).mean().item(), (d.norm() / ref.norm()).item()
    print(f"[parity] full-width 2-layer transformer at S = 53504 (49f x 1024^2): new-bf16 vs ref-fp32 MSE={mse:.3e} rel_l2={rel:.3e} "
          f"ref std {g['out_std']:.3f} | kernels {cnt}")
    assert mse < BAR and torch.isfinite(out.float()).all()
    assert cnt.get("attention_v3", 0) == 2 and cnt.get("gemm_qkv_fused", 0) == 4 and cnt.get("gemm_256_w4a", 0) >= 6, cnt


def test_vae_full_width_ragged_shape_vs_golden():
    """Full-width VAE at 5 x 96 x 168 = a quarter of the reference's published 384 x 672 shape: latents 12 x 21 (odd width), rows
    of 21 / 42 / 84 / 168 voxels (no multiple of any tile: every convolution runs on the general implicit-GEMM kernels with
    ragged last tiles), 252 mid-block keys (padded to 256 for the head_dim-512 kernel)."""
    from easyanimate_amd import AutoencoderKLMagvit, _lib
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle.gen_golden import vae_ragged_inputs
    g = _load("vae_full_ragged_5x96x168.pt")
    video, z = vae_ragged_inputs(g["input_seed"], g["frames"], g["height"], g["width"])
    assert abs(video.double().sum().item() - g["video_sum"]) < 1e-3 and abs(z.double().sum().item() - g["z_sum"]) < 1e-4
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    _lib.reset_counters()
    with torch.no_grad():
        mom = vae.encode(video.to(DEV).bfloat16())[0].parameters
        dec = vae.decode(z.to(DEV).bfloat16())[0]
    torch.cuda.synchronize()
    cnt = _lib.counters()
    assert mom.shape == g["moments"].shape and dec.shape == g["dec_f16"].shape == (1, 3, 5, 96, 168)
    mse_e, mse_d = _mse(mom.float(), g["moments"]), _mse(dec.float(), g["dec_f16"].float())
    print(f"[parity] full-width VAE 5x96x168 (ragged): encode moments MSE={mse_e:.3e} (ref-bf16 floor {g['moments_floor_mse']:.3e}); "
          f"decode MSE={mse_d:.3e} (ref-bf16 floor {g['dec_floor_mse']:.3e}) | kernels {cnt}")
    assert mse_e < BAR and mse_d < BAR
    assert cnt.get("attention_d512", 0) == 2       # the mid-block flash kernel served both passes with padded keys


# ---------------------------------------------------------------------------------------------------------
# (b) full-width VAE at 9 x 256^2
# ---------------------------------------------------------------------------------------------------------
def test_vae_full_width_vs_golden():
    from easyanimate_amd import AutoencoderKLMagvit, _lib
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle.gen_golden import vae_full_inputs
    g = _load("vae_full_9x256.pt")
    assert g["cfg"]["block_out_channels"] == [128, 256, 512, 512]
    video, z = vae_full_inputs(g["input_seed"], g["frames"], g["size"])
    assert abs(video.double().sum().item() - g["video_sum"]) < 1e-3 and abs(z.double().sum().item() - g["z_sum"]) < 1e-4
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    _lib.reset_counters()
    with torch.no_grad():
        mom = vae.encode(video.to(DEV).bfloat16())[0].parameters
    torch.cuda.synchronize()
    c_enc = _lib.counters()
    _lib.reset_counters()
    with torch.no_grad():
        dec = vae.decode(z.to(DEV).bfloat16())[0]
    torch.cuda.synchronize()
    c_dec = _lib.counters()
    assert mom.shape == g["moments"].shape and dec.shape == g["dec_f16"].shape == (1, 3, 9, 256, 256)
    mse_e, mse_d = _mse(mom.float(), g["moments"]), _mse(dec.float(), g["dec_f16"].float())
    print(f"[parity] full-width VAE 9x256^2: encode moments MSE={mse_e:.3e} (ref-bf16 floor {g.get('moments_floor_mse', float('nan')):.3e}, "
          f"ref std {g['moments_std']:.3f}); decode MSE={mse_d:.3e} (ref-bf16 floor {g.get('dec_floor_mse', float('nan')):.3e}, "
          f"ref std {g['dec_std']:.3f})")
    print(f"[parity] full-width VAE kernels: encode {c_enc}; decode {c_dec}")
    assert mse_e < BAR and mse_d < BAR
    # the kernels that carry the 49 x 1024^2 decode must be the ones that ran here (automatic dispatch, no options set)
    assert c_dec.get("conv_row16_128", 0) >= 6          # C_out = 128 full-resolution layers (row-slab, 16x16x32)
    assert c_dec.get("conv_row16_256_ups", 0) >= 1      # the 256-channel up-sampler with folded x2 addressing
    assert c_dec.get("conv_pp_256x256", 0) >= 1         # 256 x 256 ping-pong implicit GEMM
    assert c_enc.get("conv_row16_128", 0) >= 4 and c_enc.get("conv_pp_256x256", 0) >= 1
    assert c_dec.get("attention_d512", 0) == 1 and c_enc.get("attention_d512", 0) == 1     # mid block: head_dim-512 flash kernel


def test_vae_decoder_512_rows_vs_golden():
    """Full-width decoder at 5 x 512^2 against the reference (fp32, chunked mode; every second pixel stored): output rows of
    512 voxels send the C_out = 128 layers to the 512-voxel row-slab kernel -- the kernel that carries a 49 x 1024^2 decode
    -- inside the real network, automatic dispatch."""
    from easyanimate_amd import AutoencoderKLMagvit, _lib
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle.gen_golden import vae_full_inputs
    g = _load("vae_dec_5x512.pt")
    _, z = vae_full_inputs(g["input_seed"], g["frames"], g["size"])
    assert abs(z.double().sum().item() - g["z_sum"]) < 1e-4
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    _lib.reset_counters()
    with torch.no_grad():
        dec = vae.decode(z.to(DEV).bfloat16())[0]
    torch.cuda.synchronize()
    c = _lib.counters()
    assert tuple(dec.shape) == tuple(g["dec_shape"]) == (1, 3, 5, 512, 512)
    mse = _mse(dec[..., ::2, ::2].float(), g["dec_sub_f16"].float())
    print(f"[parity] full-width decoder 5x512^2: MSE={mse:.3e} (ref std {g['dec_std']:.3f}); kernels {c}")
    assert mse < BAR
    assert c.get("conv_row16_m512", 0) >= 6 and c.get("conv_row16_256_ups", 0) >= 1 and (c.get("conv_row16_256", 0) + c.get("conv_row16_256_k32", 0)) >= 1
    # the two algebraic shortcuts are in play in this decode: the 256-voxel source rows of the last up-sampler take the sub-pixel
    # kernel, and the first convolution behind each virtual temporal x2 runs 18 merged taps
    assert c.get("conv_row16_256_subpixel", 0) >= 1 and c.get("conv_tmerge_18_taps", 0) >= 1, c


def test_vae_decoder_two_tile_rows_vs_golden():
    """Full-width decoder at 5 x 256 x 1024 against the reference (fp32, chunked mode; every second pixel stored): output rows of
    1024 voxels = TWO 512-voxel tiles per row (the geometry of every full-resolution layer of a 49 x 1024^2 decode), both large
    up-samplers on the sub-pixel kernel (source rows of 256 and 512 voxels), merged temporal taps behind both virtual x2 -- all
    of the round-3 shortcuts active together at depth, against the unchanged reference (vaemodules/upsamplers.py:123-153,
    omnigen_enc_dec.py:555-677)."""
    from easyanimate_amd import AutoencoderKLMagvit, _lib
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle.gen_golden import vae_ragged_inputs
    g = _load("vae_dec_5x256x1024.pt")
    _, z = vae_ragged_inputs(g["input_seed"], g["frames"], g["height"], g["width"])
    assert abs(z.double().sum().item() - g["z_sum"]) < 1e-4
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    _lib.reset_counters()
    with torch.no_grad():
        dec = vae.decode(z.to(DEV).bfloat16())[0]
    torch.cuda.synchronize()
    c = _lib.counters()
    assert tuple(dec.shape) == tuple(g["dec_shape"]) == (1, 3, 5, 256, 1024)
    mse = _mse(dec[..., ::2, ::2].float(), g["dec_sub_f16"].float())
    print(f"[parity] full-width decoder 5x256x1024 (two 512-voxel tiles per row): MSE={mse:.3e} (ref std {g['dec_std']:.3f}; the reference's "
          f"own bf16 decode: {g.get('dec_floor_mse', float('nan')):.3e}); kernels {c}")
    assert mse < BAR
    assert c.get("conv_row16_m512", 0) >= 6 and c.get("conv_row16_256_subpixel", 0) >= 2 and c.get("conv_tmerge_18_taps", 0) >= 2, c


# ---------------------------------------------------------------------------------------------------------
# (e) round 3: a multi-step loop at FULL WIDTH (VERDICT r2 next #6b)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["loop_full_width", "loop_full_width_default"])
def test_denoise_loop_full_width_meets_the_bar(name):
    """d = 3072 (48 heads x 64, ff 12288, text 3584), 2 layers, 1680 video + 256 text tokens, 10 Flow steps, CFG 6, through
    EasyAnimatePipeline.denoise with the production kernels (attention v3, 256^2 GEMMs, fused QKV asserted), against the
    unchanged reference's fp32 loop (tests/golden/loop_full_width*.pt).  Bar 1e-4 on the latents, NO floor term."""
    from easyanimate_amd import EasyAnimatePipeline, FlowMatchEulerDiscreteScheduler, _lib
    from easyanimate_amd.embeddings import get_3d_rotary_pos_embed
    from oracle.gen_golden import loop_full_width_inputs
    g = _load(name + ".pt")
    Fr, H, W, T = g["dims"]
    latents, enc = loop_full_width_inputs(g["cfg"], g["input_seed"])
    assert torch.equal(latents, g["latents"]) and abs(enc.double().sum().item() - g["enc_sum"]) < 1e-4
    rope = get_3d_rotary_pos_embed(64, g["crops"], grid_size=(H // 2, W // 2), temporal_size=Fr, use_real=True)
    m = _model(g["cfg"], g["shapes"], g["seed"], g["style"])
    sched = FlowMatchEulerDiscreteScheduler(shift=1.0)
    sched.set_timesteps(g["steps"], device=DEV, mu=1)
    pipe = EasyAnimatePipeline(vae=None, transformer=m, scheduler=sched)
    ref, refb = g["trace"], g["trace_bf16"]
    kept = {}

    def keep(p, i, t, kw):
        if (i + 1) in ref:
            kept[i + 1] = kw["latents"].float().cpu().clone()
        return {}
    _lib.reset_counters()
    with torch.no_grad():
        pipe.denoise(latents.to(DEV).bfloat16(), enc.to(DEV).bfloat16(), rope, sched.timesteps, g["guidance"], callback_on_step_end=keep)
    cnt = _lib.counters()
    res = {k: _mse(v, ref[k]) for k, v in kept.items()}
    floor = {k: _mse(refb[k], ref[k]) for k in ref}
    print(f"[parity] {name} (full width d=3072, 2 layers, {Fr * (H // 2) * (W // 2)} video + {T} text tokens, {g['steps']} steps, CFG "
          f"{g['guidance']:g}): latent MSE by step new vs ref-fp32 " + ", ".join(f"{k}: {v:.3e}" for k, v in res.items())
          + " | ref-bf16 vs ref-fp32 (floor) " + ", ".join(f"{k}: {v:.3e}" for k, v in floor.items())
          + f" | final latent std {ref[g['steps']].std().item():.3f} | kernels {cnt}")
    assert cnt.get("attention_v3", 0) >= 2 * g["steps"] and cnt.get("gemm_256_w4a", 0) > 0 and cnt.get("gemm_qkv_fused", 0) > 0
    assert all(v < BAR for v in res.values()), res
