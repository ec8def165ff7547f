""""predict_t2v.py drops in unchanged" as a test: the reference's own entry scripts are parsed with
`ast`, every attribute they read on `pipeline`, `transformer`, `vae`, `scheduler`, every keyword they pass to
`pipeline(...)` and to the loaders is collected, and the product classes must provide all of them.  The scripts are read
from /root/reference when it is present (this container); the collected surface is ALSO committed as
tests/golden/script_surface.json so the check runs on the GPU box, where the reference tree does not exist."""
import ast
import inspect
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SURFACE = os.path.join(ROOT, "tests", "golden", "script_surface.json")
SCRIPTS = ("predict_t2v.py", "predict_i2v.py", "predict_v2v.py")
ROOTS = ("pipeline", "transformer", "vae", "scheduler")
LOADERS = {"Choosen_Transformer3DModel": "transformer_cls", "Choosen_AutoencoderKL": "vae_cls", "Choosen_Scheduler": "scheduler_cls"}


def _chain(node):
    """a.b.c -> ["a", "b", "c"] for pure Name/Attribute chains, else None"""
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return parts[::-1]
    return None


def collect(path):
    tree = ast.parse(open(path).read())
    reads, call_kw, loader_kw = set(), set(), {}
    stores = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Assign, ast.AugAssign)):
            for t in (node.targets if isinstance(node, ast.Assign) else [node.target]):
                c = _chain(t)
                if c and c[0] in ROOTS and len(c) > 1:
                    stores.add(".".join(c))
        if isinstance(node, ast.Attribute) and isinstance(node.ctx, ast.Load):
            c = _chain(node)
            if c and c[0] in ROOTS:
                reads.add(".".join(c))
        if isinstance(node, ast.Call):
            if isinstance(node.func, ast.Name) and node.func.id == "pipeline":
                call_kw |= {k.arg for k in node.keywords if k.arg}
                call_kw.add(f"<positional:{len(node.args)}>")
            c = _chain(node.func)
            if c and c[0] in LOADERS and len(c) == 2:
                loader_kw.setdefault(f"{LOADERS[c[0]]}.{c[1]}", set()).update(k.arg for k in node.keywords if k.arg)
    # a read of a.b.c implies reads of a.b; keep maximal chains only for the report, resolve all prefixes in the check
    return {"reads": sorted(reads), "stores": sorted(stores), "pipeline_call_keywords": sorted(call_kw),
            "loader_keywords": {k: sorted(v) for k, v in sorted(loader_kw.items())}}


def _surface():
    if os.path.isdir(REF):
        out = {s: collect(os.path.join(REF, s)) for s in SCRIPTS}
        if not os.path.exists(SURFACE) or json.load(open(SURFACE)) != out:
            os.makedirs(os.path.dirname(SURFACE), exist_ok=True)
            json.dump(out, open(SURFACE, "w"), indent=1, sort_keys=True)
        return out
    return json.load(open(SURFACE))


def _tiny_objects():
    from easyanimate_amd import (AutoencoderKLMagvit, EasyAnimateInpaintPipeline, EasyAnimatePipeline,
                                 EasyAnimateTransformer3DModel, FlowMatchEulerDiscreteScheduler)
    tcfg = dict(num_attention_heads=2, attention_head_dim=64, in_channels=16, out_channels=16, patch_size=2, num_layers=1,
                time_embed_dim=64, add_norm_text_encoder=True, text_embed_dim=32, text_embed_dim_t5=None, norm_eps=1e-5,
                time_position_encoding_type="3d_rope", enable_text_attention_mask=True)
    with torch.device("meta"):
        tr = EasyAnimateTransformer3DModel(**tcfg)
        tr_inp = EasyAnimateTransformer3DModel(**{**tcfg, "in_channels": 33})
        vae = AutoencoderKLMagvit(
            in_channels=3, out_channels=3, block_out_channels=[64, 64, 128, 128], norm_num_groups=16, latent_channels=16,
            down_block_types=("SpatialDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D"),
            up_block_types=("SpatialUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D"),
            mid_block_attention_type="spatial", spatial_group_norm=True, cache_mag_vae=True, slice_mag_vae=False,
            mini_batch_encoder=4, mini_batch_decoder=1)
    sch = FlowMatchEulerDiscreteScheduler(shift=1.0)
    objs = []
    for cls, t in ((EasyAnimatePipeline, tr), (EasyAnimateInpaintPipeline, tr_inp)):
        objs.append({"pipeline": cls(vae=vae, transformer=t, scheduler=sch), "transformer": t, "vae": vae, "scheduler": sch})
    return objs


def test_reference_scripts_find_every_attribute_they_touch():
    surface = _surface()
    objs = _tiny_objects()
    missing = []
    for script, s in surface.items():
        for chain in s["reads"]:
            parts = chain.split(".")
            for o in objs:
                cur = o[parts[0]]
                for i, a in enumerate(parts[1:], 1):
                    if cur is None:       # e.g. pipeline.text_encoder is None in this fixture: its own attributes are transformers'
                        break
                    if not hasattr(cur, a):
                        missing.append(f"{script}: {'.'.join(parts[:i + 1])} (on {type(cur).__name__})")
                        break
                    cur = getattr(cur, a)
    assert not missing, "the reference scripts read attributes the product does not provide:\n  " + "\n  ".join(sorted(set(missing)))


def test_reference_scripts_pipeline_call_keywords_are_accepted():
    from easyanimate_amd import EasyAnimateInpaintPipeline, EasyAnimatePipeline
    surface = _surface()
    t2v = set(inspect.signature(EasyAnimatePipeline.__call__).parameters)
    inp = set(inspect.signature(EasyAnimateInpaintPipeline.__call__).parameters)
    inp_only = {"video", "mask_video", "clip_image", "strength", "masked_video_latents", "noise_aug_strength"}
    for script, s in surface.items():
        kws = {k for k in s["pipeline_call_keywords"] if not k.startswith("<")}
        assert kws <= inp, f"{script}: EasyAnimateInpaintPipeline.__call__ lacks {sorted(kws - inp)}"
        # a script's plain-T2V branch passes only the non-conditioning keywords
        assert (kws - inp_only) <= t2v, f"{script}: EasyAnimatePipeline.__call__ lacks {sorted(kws - inp_only - t2v)}"
        # `prompt` is passed positionally as the first argument
        assert list(inspect.signature(EasyAnimatePipeline.__call__).parameters)[1] == "prompt"
        assert list(inspect.signature(EasyAnimateInpaintPipeline.__call__).parameters)[1] == "prompt"


def test_reference_scripts_loader_keywords_are_accepted():
    from easyanimate_amd import AutoencoderKLMagvit, EasyAnimateTransformer3DModel, FlowMatchEulerDiscreteScheduler
    cls = {"transformer_cls": EasyAnimateTransformer3DModel, "vae_cls": AutoencoderKLMagvit, "scheduler_cls": FlowMatchEulerDiscreteScheduler}
    for script, s in _surface().items():
        for name, kws in s["loader_keywords"].items():
            c, meth = name.split(".")
            assert hasattr(cls[c], meth), f"{script}: {cls[c].__name__}.{meth} missing"
            params = inspect.signature(getattr(cls[c], meth)).parameters
            has_var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
            lacking = [k for k in kws if k not in params and not has_var_kw]
            assert not lacking, f"{script}: {cls[c].__name__}.{meth} does not accept {lacking}"


def test_offload_entry_points_keep_everything_resident():
    """predict_t2v.py:256-273: every GPU_memory_mode branch calls one of these; on a 288 GB device they make the models
    resident instead of installing offload hooks.  (CPU here: the models simply stay where they are.)"""
    o = _tiny_objects()[0]
    p = o["pipeline"]
    p._manual_cpu_offload_in_sequential_cpu_offload = []      # the script sets this attribute before the call
    assert p.enable_model_cpu_offload(device="meta") is p
    assert p.enable_sequential_cpu_offload(device="meta") is p
    assert p.maybe_free_model_hooks() is None


def test_compute_dtype_is_never_a_float8_storage_type():
    """with the transformer stored as float8_e4m3fn the pipeline must embed / sample in bf16."""
    o = _tiny_objects()[0]
    p = o["pipeline"]

    class _T:
        dtype = torch.float8_e4m3fn
        device = torch.device("cpu")
    p.transformer = _T()
    assert p.compute_dtype == p.vae.dtype          # the script casts the VAE to weight_dtype (predict_t2v.py:142)
    p.vae = None
    assert p.compute_dtype == torch.bfloat16
    lat = p.prepare_latents(1, 16, 1, 64, 64, p.compute_dtype, "cpu", torch.Generator().manual_seed(0))
    assert lat.dtype == torch.bfloat16 and lat.shape == (1, 16, 1, 8, 8)


def test_inpaint_preprocessing_resizes_and_guards_normalisation():
    """video / mask are brought to (height, width) like VaeImageProcessor.preprocess(height=, width=) does
    (pipeline_easyanimate_inpaint.py:1231-1233,1340), and an input that is already in [-1, 1] is not normalised twice."""
    from easyanimate_amd import EasyAnimateInpaintPipeline as P
    v = torch.rand(1, 3, 5, 40, 56)
    m = torch.zeros(1, 1, 5, 40, 56)
    m[:, :, 1:] = 255
    mv, mc = P.masked_video_and_mask(v, m, 32, 48)
    assert mv.shape == (1, 3, 5, 32, 48) and mc.shape == (1, 1, 5, 32, 48)
    assert torch.equal(mc[:, :, 0], torch.zeros(1, 1, 32, 48)) and bool((mc[:, :, 1:] == 1).all())
    assert bool((mv[:, :, 1:] == -1).all()) and float(mv[:, :, 0].min()) >= -1 and float(mv[:, :, 0].max()) <= 1
    same, _ = P.masked_video_and_mask(v, m)          # no size given: untouched geometry
    assert same.shape == v.shape
    neg = v * 2 - 1
    assert torch.equal(P.preprocess_video(neg), neg)  # min() < 0: diffusers skips the normalisation
    assert torch.allclose(P.preprocess_video(v), v * 2 - 1)
