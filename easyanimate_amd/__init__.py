"""easyanimate_amd -- MI355X-native implementation of EasyAnimate's diffusion sampling hot path.

Host code mirrors the reference's module / processor / pipeline interface (same class names, constructor
and forward signatures, config attributes and state-dict keys); all arithmetic on the path runs in the
hand-written gfx950 kernels of libea_mi355x.so (include/ea_mi355x.h) -- there is no eager fallback.
"""
__version__ = "0.1.0"

_LAZY = {
    "EasyAnimateTransformer3DModel": "transformer3d",
    "EasyAnimateDiTBlock": "attention",
    "EasyAnimateAttnProcessor2_0": "processor",
    "AutoencoderKLMagvit": "autoencoder_magvit",
    "FlowMatchEulerDiscreteScheduler": "scheduler",
    "EasyAnimatePipeline": "pipeline",
    "EasyAnimateInpaintPipeline": "pipeline",
    "TeaCache": "teacache",
    "get_teacache_coefficients": "teacache",
    "name_to_transformer3d": "registry",
    "name_to_autoencoder_magvit": "registry",
    "Qwen2VLTextEncoderHIP": "text_encoder",
    "use_hip_text_encoder": "text_encoder",
}


def invalidate_weight_cache() -> None:
    from ._params import invalidate_weight_cache as _inv
    _inv()


def observes_weight_writes(fn):
    """Wrap a function that edits parameters through `.data` (the reference's utils/lora_utils.py merge_lora /
    unmerge_lora): the derived-weight cache is dropped after it returns.  The pipelines also do this at the start of every
    __call__, so `pipeline = merge_lora(pipeline, ...); pipeline(...)` needs nothing; use this when driving
    transformer.forward / vae.decode directly.  (bf16 Linear weights are read in place and need nothing; their one derived copy,
    the K-blocked ff.net.2 weight, is refreshed by FeedForward on every call outside _params.weights_frozen().)"""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        try:
            return fn(*a, **kw)
        finally:
            invalidate_weight_cache()
    return wrapped


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    raise AttributeError(name)
