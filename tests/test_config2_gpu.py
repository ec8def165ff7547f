"""BASELINE.json configs[1] at FULL DEPTH x FULL LENGTH: one forward of the declared 7B-class DiT (L = 28, d = 3072) at
49 frames x 512 x 512 -- latents [.,16,13,64,64], 13 312 video + 256 text tokens = S 13 568 -- for the CFG pair, against
tests/golden/config2_7b_49x512_v0.pt, which oracle/gen_golden.py (section config2_forward) produced by running the UNCHANGED
reference transformer in fp32 on the host cores (~3e14 FLOP, two B = 1 passes of 6-8 minutes), together with the reference's
own bf16 forward from the same inputs (its per-forward noise floor).  Reference: easyanimate/models/transformer3d.py:1496-1689.

The deepest full-width check before this was L = 28 at 256 video tokens (config 1), the longest S = 5 376 at L = 2."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DEV = "cuda"


def test_config2_full_depth_forward_vs_reference_golden():
    from easyanimate_amd import EasyAnimateTransformer3DModel, _lib
    from easyanimate_amd.embeddings import get_3d_rotary_pos_embed
    from easyanimate_amd.synthetic import fill_module_
    from oracle.gen_golden import config2_inputs
    g = torch.load(os.path.join(GOLD, "config2_7b_49x512_v0.pt"), weights_only=False)
    assert g["cfg"]["num_layers"] == 28 and g["cfg"]["num_attention_heads"] * 64 == 3072
    latents, enc = config2_inputs()
    assert abs(latents.double().sum().item() - g["latents_sum"]) < 1e-6 and abs(enc.double().sum().item() - g["enc_sum"]) < 1e-3
    with torch.device("meta"):
        m = EasyAnimateTransformer3DModel.from_config(g["cfg"])
    m = m.to(torch.bfloat16).to_empty(device=DEV).eval()
    fill_module_(m, g["seed"], g["style"])          # the values the reference run used (bf16-representable), streamed per tensor
    rope = get_3d_rotary_pos_embed(64, g["crops"], grid_size=(32, 32), temporal_size=13, use_real=True)
    t = torch.tensor([g["timestep"]] * 2, device=DEV).bfloat16()
    _lib.reset_counters()
    with torch.no_grad():
        v = m(torch.cat([latents] * 2).to(DEV).bfloat16(), t, encoder_hidden_states=enc.to(DEV).bfloat16(), image_rotary_emb=rope,
              return_dict=False)[0]
    torch.cuda.synchronize()
    cnt = _lib.counters()
    ref = g["v"].double()
    d = v.float().cpu().double() - ref
    mse, rel = (d ** 2).mean().item(), (d.norm() / ref.norm()).item()
    print(f"[parity] config 2 (7B-class L=28 d=3072, 49f x 512^2, S=13568, CFG pair) one forward vs the reference's fp32 forward: "
          f"velocity MSE {mse:.3e}, rel-L2 {rel:.3e}; the reference's own bf16 forward: {g['floor_mse']:.3e} / {g['rel_l2_floor']:.3e}; "
          f"fp16 storage of the golden: {g['fp16_storage_mse']:.1e}; kernels {dict((k, n) for k, n in cnt.items() if k.startswith(('attention', 'gemm_qkv', 'gemm_256')))}")
    assert v.shape == (2, 16, 13, 64, 64) and torch.isfinite(v.float()).all()
    assert mse < 1e-4
    # one contiguous attention launch per block over both batch elements; text and video streams both on the fused QKV launch
    assert cnt.get("attention_v3", 0) == 28 and cnt.get("gemm_qkv_fused", 0) == 56 and cnt.get("gemm_256_w4a", 0) >= 28 * 3 and cnt.get("gemm_qkv_fused_w4a", 0) == 56, cnt
