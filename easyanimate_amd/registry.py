"""Model registries with the reference's keys (easyanimate/models/__init__.py:6-15), selected by the YAML
keys transformer_additional_kwargs.transformer_type / vae_kwargs.vae_type (predict_t2v.py:94-96,135-137)."""
from .transformer3d import EasyAnimateTransformer3DModel

name_to_transformer3d = {
    "EasyAnimateTransformer3DModel": EasyAnimateTransformer3DModel,
}


def _vae():
    from .autoencoder_magvit import AutoencoderKLMagvit
    return AutoencoderKLMagvit


class _LazyVaeRegistry(dict):
    def __missing__(self, key):
        if key == "AutoencoderKLMagvit":
            self[key] = _vae()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key == "AutoencoderKLMagvit" or super().__contains__(key)


name_to_autoencoder_magvit = _LazyVaeRegistry()
