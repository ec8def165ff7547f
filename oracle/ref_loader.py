"""TEST INFRASTRUCTURE ONLY.  Imports the *unchanged* reference from /root/reference through the
diffusers shim.  Works only in the build container (the reference tree does not travel to the GPU
box); used by oracle/gen_golden.py and by the `not gpu` tests that pin oracle/restatement.py.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("EA_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "easyanimate", "models"))


def load():
    """Returns a namespace with the reference classes on the hot path."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    from . import diffusers_shim
    diffusers_shim.install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import types
    ns = types.SimpleNamespace()
    from easyanimate.models import transformer3d, attention, processor, norm, autoencoder_magvit
    from easyanimate.vae.ldm.models import omnigen_enc_dec
    from easyanimate.vae.ldm.modules.vaemodules import common as vae_common
    from easyanimate.pipeline import pipeline_easyanimate  # noqa: F401  (only for helper functions)
    ns.transformer3d = transformer3d
    ns.attention = attention
    ns.processor = processor
    ns.norm = norm
    ns.autoencoder_magvit = autoencoder_magvit
    ns.omnigen_enc_dec = omnigen_enc_dec
    ns.vae_common = vae_common
    ns.pipeline_easyanimate = pipeline_easyanimate
    ns.shim = diffusers_shim
    return ns
