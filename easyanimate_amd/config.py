"""Config recording with the surface the reference gets from diffusers' ConfigMixin / register_to_config
(SURVEY.md 5.6): `model.config.x`, `model.config.get("x", default)`, `Model.from_config(dict, **overrides)`.
YAML files (config/easyanimate_video_v5.1_magvit_qwen.yaml) are read with PyYAML (omegaconf is not required)."""
from __future__ import annotations

import functools
import inspect
import json
import os
from typing import Any, Dict


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        raise AttributeError("config is read-only; use register_to_config")


def register_to_config(init):
    """Decorator for __init__: records the bound arguments (defaults applied) into self.config."""
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self" and bound.signature.parameters[k].kind
               not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL)}
        object.__setattr__(self, "_internal_dict", FrozenDict(cfg))
        init(self, *args, **kwargs)

    return wrapped


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self) -> FrozenDict:
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        object.__setattr__(self, "_internal_dict", FrozenDict(d))

    @classmethod
    def from_config(cls, config: Dict[str, Any], **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        names = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in cfg.items() if k in names})

    @classmethod
    def load_config(cls, path: str, subfolder: str | None = None) -> Dict[str, Any]:
        if subfolder:
            path = os.path.join(path, subfolder)
        if os.path.isdir(path):
            path = os.path.join(path, cls.config_name)
        with open(path) as f:
            cfg = json.load(f)
        return {k: v for k, v in cfg.items() if not k.startswith("_")}


def load_yaml(path: str) -> Dict[str, Any]:
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)
