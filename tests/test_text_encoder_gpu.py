"""The text-encoder step on the device library (easyanimate_amd/text_encoder.py, ea_text.hip) against the class the reference
loads into the slot -- transformers' Qwen2VLForConditionalGeneration (pipeline_easyanimate.py:438-447 takes
`.hidden_states[-2]` of 256 right-padded prompt tokens) -- random-init (no checkpoint is reachable), run in fp32 on the host as
the oracle and in bf16 on the GPU as the floor.  Kernel-level cases against plain fp64 torch."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _qwen(hidden, heads, kv_heads, inter, layers, vocab=512, seed=0):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    hd = hidden // heads
    sec = [hd // 2 - 2 * (hd * 3 // 16), hd * 3 // 16, hd * 3 // 16]          # three sections summing to head_dim / 2 (7B: [16, 24, 24])
    cfg = Qwen2VLConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                        num_key_value_heads=kv_heads, max_position_embeddings=1024, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                        rope_scaling={"type": "mrope", "mrope_section": sec}, rope_theta=1000000.0,
                        vision_config=dict(depth=1, embed_dim=32, hidden_size=hidden, num_heads=2, mlp_ratio=2, in_channels=3,
                                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2))
    torch.manual_seed(seed)
    m = Qwen2VLForConditionalGeneration(cfg).eval()
    with torch.no_grad():                      # bf16-representable weights: the fp32 oracle and the bf16 product share them exactly;
        for n, p in m.named_parameters():      # norm gains / biases perturbed so that they matter
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape))
            p.copy_(p.bfloat16().float())
    return m


def _prompt_batch(B, S, vocab, lens, seed=3):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (B, S), generator=g)
    mask = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    return ids, mask


@pytest.mark.parametrize("D,Hq,Hkv,S,causal,lens", [(128, 4, 2, 80, True, (80, 33)), (64, 6, 2, 45, True, (45, 7)), (128, 28, 4, 256, True, (256, 61)),
                                                    (64, 2, 2, 96, False, (96, 50))])
def test_attention_causal_gqa_kernel(D, Hq, Hkv, S, causal, lens):
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(D + S)
    B = len(lens)
    q = torch.randn(B, Hq, S, D, generator=g).bfloat16()
    k = torch.randn(B, Hkv, S, D, generator=g).bfloat16()
    v = torch.randn(B, Hkv, S, D, generator=g).bfloat16()
    sp = ops.round_up(S, 32)
    vt = torch.zeros(B, Hkv, D, sp, dtype=torch.bfloat16)
    vt[..., :S] = v.transpose(2, 3)
    valid = torch.tensor(lens, dtype=torch.int32)
    out = ops.attention_causal_gqa(q.to(DEV), k.to(DEV), vt.to(DEV), S, D ** -0.5, valid.to(DEV), causal=causal).float().cpu()
    kk, vv = k.double().repeat_interleave(Hq // Hkv, 1), v.double().repeat_interleave(Hq // Hkv, 1)
    sc = q.double() @ kk.transpose(2, 3) * D ** -0.5
    keys = torch.arange(S)
    bad = keys[None, None, None, :] >= valid[:, None, None, None]
    if causal:
        bad = bad | (keys[None, None, None, :] > keys[None, None, :, None])
    ref = (sc.masked_fill(bad, -1e300).softmax(-1) @ vv).transpose(1, 2).reshape(B, S, Hq * D)
    rel = ((out.double() - ref).norm() / ref.norm()).item()
    print(f"[parity] attention_causal_gqa D={D} Hq={Hq} Hkv={Hkv} S={S} causal={causal}: rel-L2 {rel:.3e}, max {(out.double() - ref).abs().max().item():.3e}")
    assert rel < 6e-3


def test_rope_half_scatter_and_silu_mul_kernels():
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(4)
    B, S, H, D = 2, 37, 3, 128
    src = torch.randn(B * S, H * D + 64, generator=g).bfloat16()               # a wider projection output: row stride != H * D
    ang = torch.rand(B * S, D // 2, generator=g) * 6.28
    ang = torch.cat([ang, ang], -1)
    cos, sin = ang.cos(), ang.sin()
    got = ops.rope_half_scatter(src.to(DEV)[:, :H * D], H, D, B, S, cos.to(DEV), sin.to(DEV)).float().cpu()
    x = src[:, :H * D].double().view(B, S, H, D)
    rot = torch.cat([-x[..., D // 2:], x[..., :D // 2]], -1)
    ref = (x * cos.double().view(B, S, 1, D) + rot * sin.double().view(B, S, 1, D)).permute(0, 2, 1, 3)
    assert (got.double() - ref).abs().max().item() < 2e-2 and ((got.double() - ref).norm() / ref.norm()).item() < 3e-3
    plain = ops.rope_half_scatter(src.to(DEV)[:, :H * D], H, D, B, S).cpu()
    assert torch.equal(plain, src[:, :H * D].view(B, S, H, D).permute(0, 2, 1, 3))
    gu = torch.randn(50, 2 * 264, generator=g).bfloat16()
    got = ops.silu_mul(gu.to(DEV)[:, :264], gu.to(DEV)[:, 264:]).float().cpu()
    ref = torch.nn.functional.silu(gu[:, :264].double()) * gu[:, 264:].double()
    assert ((got.double() - ref).norm() / ref.norm()).item() < 3e-3


CASES = {
    "hd64": dict(hidden=256, heads=4, kv_heads=2, inter=512, layers=3, B=2, S=48, lens=(48, 19)),
    "hd128": dict(hidden=512, heads=4, kv_heads=2, inter=1024, layers=4, B=2, S=64, lens=(41, 64)),
    # Qwen2-VL-7B's widths (3584 = 28 x 128, 4 kv heads, MLP 18 944), 2 layers, the pipelines' 256 padded tokens
    "7b_width": dict(hidden=3584, heads=28, kv_heads=4, inter=18944, layers=2, B=1, S=256, lens=(37,)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_text_encoder_vs_transformers(case):
    from easyanimate_amd import _lib
    from easyanimate_amd.text_encoder import Qwen2VLTextEncoderHIP
    c = CASES[case]
    hf32 = _qwen(c["hidden"], c["heads"], c["kv_heads"], c["inter"], c["layers"])
    ids, mask = _prompt_batch(c["B"], c["S"], 512, c["lens"])
    with torch.no_grad():
        ref = hf32(input_ids=ids, attention_mask=mask, output_hidden_states=True).hidden_states
        hfb = hf32.to(torch.bfloat16).to(DEV)
        floor = hfb(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), output_hidden_states=True).hidden_states
        enc = Qwen2VLTextEncoderHIP(hfb)
        assert enc.device.type == "cuda" and enc.dtype == torch.bfloat16
        _lib.reset_counters()
        got = enc(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), output_hidden_states=True).hidden_states
    cnt = _lib.counters()
    assert len(got) == len(ref) == c["layers"] + 1
    rels = []
    for i, (a, b, f) in enumerate(zip(got, ref, floor)):
        r = ((a.float().cpu().double() - b.double()).norm() / b.double().norm()).item()
        rf = ((f.float().cpu().double() - b.double()).norm() / b.double().norm()).item()
        rels.append((r, rf))
    pen, pen_floor = rels[-2]
    # the rows the DiT really consumes include the PADDED ones (transformer3d.py:1502 drops the mask): all rows are compared
    print(f"[parity] text encoder {case} (hidden {c['hidden']}, {c['heads']}/{c['kv_heads']} heads, {c['layers']} layers, S={c['S']}, valid {c['lens']}) vs "
          f"transformers fp32: rel-L2 by hidden state {[f'{r:.2e}' for r, _ in rels]}; transformers' own bf16 run {[f'{r:.2e}' for _, r in rels]}; "
          f"hidden_states[-2]: {pen:.3e} (floor {pen_floor:.3e}); kernels {cnt}")
    assert torch.equal(got[0].cpu(), floor[0].cpu())                       # the embedding lookup
    assert pen < max(1e-2, 2 * pen_floor) and all(r < max(1.5e-2, 2.5 * rf) for r, rf in rels[1:])
    assert cnt.get("attention_causal_gqa", 0) == c["layers"]


def test_encode_prompt_through_the_hip_text_encoder():
    """The pipeline's encode_prompt (pipeline_easyanimate.py:421-460) with the device-library encoder in the text_encoder slot
    (text_encoder.use_hip_text_encoder): same call, same outputs as with the transformers model it wraps."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import types
    from test_pipeline_gpu import _DeviceTokenizer
    from easyanimate_amd.pipeline import EasyAnimatePipeline
    from easyanimate_amd.text_encoder import Qwen2VLTextEncoderHIP, use_hip_text_encoder
    hf = _qwen(256, 4, 2, 512, 3, vocab=128).to(torch.bfloat16).to(DEV)
    cfg = types.SimpleNamespace(enable_text_attention_mask=True, get=lambda k, d=None: {"enable_text_attention_mask": True}.get(k, d))
    tr = types.SimpleNamespace(config=cfg)
    tok = _DeviceTokenizer()
    a = EasyAnimatePipeline(vae=None, text_encoder=hf, tokenizer=tok, transformer=tr, scheduler=None)
    b = EasyAnimatePipeline(vae=None, text_encoder=hf, tokenizer=tok, transformer=tr, scheduler=None)
    assert isinstance(use_hip_text_encoder(b), Qwen2VLTextEncoderHIP) and b.text_encoder.hf is hf
    with torch.no_grad():
        ra = a.encode_prompt("a dog shakes its head", DEV, torch.bfloat16, 1, True, "blurry, static")
        rb = b.encode_prompt("a dog shakes its head", DEV, torch.bfloat16, 1, True, "blurry, static")
    for x, y in zip(ra[:2], rb[:2]):
        assert x.shape == y.shape == (1, 256, 256)
        rel = ((x.float() - y.float()).norm() / x.float().norm()).item()
        assert rel < 2e-2, rel
    assert torch.equal(ra[2], rb[2]) and torch.equal(ra[3], rb[3])
