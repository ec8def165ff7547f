"""TEST INFRASTRUCTURE ONLY -- never import from product code (easyanimate_amd/).

A minimal stand-in for ``diffusers`` (pinned by the reference at >=0.30.1,<=0.31.0,
/root/reference/requirements.txt:25) so that the *unchanged* reference modules under
/root/reference/easyanimate can be imported and executed on CPU in this container, where the
real package is not installed.  Only the handful of symbols whose arithmetic is on the hot path
are real restatements; everything else the reference merely imports is an
inert placeholder produced on demand.

Parity status: the restated functions below follow the published diffusers 0.30/0.31 behaviour
but could not be cross-checked against the real package here ("parity unpinned" at this
boundary -- regenerate tests/golden with real diffusers where it is available).
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import inspect
import math
import sys
import types
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

_PREFIX = "diffusers"


# ----------------------------------------------------------------------------------------------
# config plumbing (ConfigMixin / register_to_config / ModelMixin)
# ----------------------------------------------------------------------------------------------
class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k not in ("self", "kwargs")}
        self._internal_dict = FrozenDict(cfg)
        init(self, *args, **kwargs)

    wrapped.__wrapped__ = init
    return wrapped


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        self._internal_dict = FrozenDict(d)

    @classmethod
    def from_config(cls, config, **kw):
        cfg = dict(config)
        cfg.update(kw)
        names = set(inspect.signature(cls.__init__).parameters)
        return cls(**{k: v for k, v in cfg.items() if k in names})


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


# ----------------------------------------------------------------------------------------------
# embeddings
# ----------------------------------------------------------------------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale: int = 1):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift
        self.scale = scale

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift, self.scale)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None, sample_proj_bias=True):
        super().__init__()
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None
        self.linear_1 = nn.Linear(in_channels, time_embed_dim, sample_proj_bias)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim, sample_proj_bias)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


def apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1):
    assert use_real and use_real_unbind_dim == -1
    cos, sin = freqs_cis
    cos = cos[None, None].to(x.device)
    sin = sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)


def get_1d_rotary_pos_embed(dim, pos, theta=10000.0, use_real=False, linear_factor=1.0, ntk_factor=1.0,
                            repeat_interleave_real=True, freqs_dtype=torch.float32):
    assert dim % 2 == 0
    if isinstance(pos, int):
        pos = torch.arange(pos)
    if isinstance(pos, np.ndarray):
        pos = torch.from_numpy(pos)
    theta = theta * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=freqs_dtype, device=pos.device)[: (dim // 2)] / dim)) / linear_factor
    freqs = torch.outer(pos, freqs)
    assert use_real and repeat_interleave_real
    return (freqs.cos().repeat_interleave(2, dim=1).float(), freqs.sin().repeat_interleave(2, dim=1).float())


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta: int = 10000,
                            use_real: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    assert use_real
    start, stop = crops_coords
    grid_size_h, grid_size_w = grid_size
    grid_h = np.linspace(start[0], stop[0], grid_size_h, endpoint=False, dtype=np.float32)
    grid_w = np.linspace(start[1], stop[1], grid_size_w, endpoint=False, dtype=np.float32)
    grid_t = np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32)
    dim_t = embed_dim // 4
    dim_h = embed_dim // 8 * 3
    dim_w = embed_dim // 8 * 3
    freqs_t = get_1d_rotary_pos_embed(dim_t, grid_t, use_real=True)
    freqs_h = get_1d_rotary_pos_embed(dim_h, grid_h, use_real=True)
    freqs_w = get_1d_rotary_pos_embed(dim_w, grid_w, use_real=True)

    def combine(ft, fh, fw):
        ft = ft[:, None, None, :].expand(-1, grid_size_h, grid_size_w, -1)
        fh = fh[None, :, None, :].expand(temporal_size, -1, grid_size_w, -1)
        fw = fw[None, None, :, :].expand(temporal_size, grid_size_h, -1, -1)
        return torch.cat([ft, fh, fw], dim=-1).reshape(temporal_size * grid_size_h * grid_size_w, -1)

    return combine(freqs_t[0], freqs_h[0], freqs_w[0]), combine(freqs_t[1], freqs_h[1], freqs_w[1])


# ----------------------------------------------------------------------------------------------
# attention / feed-forward / norms
# ----------------------------------------------------------------------------------------------

def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0, interpolation_scale=1.0, base_size=16):
    """diffusers.models.embeddings.get_2d_sincos_pos_embed (0.30/0.31) [restated]: called by the reference at
    models/transformer3d.py:1424 (ref_pos_embedding of the ref-latent control branch).  float64 numpy, [H*W, embed_dim]."""
    import numpy as np
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    grid_h = np.arange(grid_size[0], dtype=np.float32) / (grid_size[0] / base_size) / interpolation_scale
    grid_w = np.arange(grid_size[1], dtype=np.float32) / (grid_size[1] / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0)          # w goes first
    grid = grid.reshape([2, 1, grid_size[1], grid_size[0]])

    def one(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)
    if cls_token and extra_tokens > 0:
        emb = np.concatenate([np.zeros([extra_tokens, embed_dim]), emb], axis=0)
    return emb


class Attention(nn.Module):
    """Holder of to_q/k/v, norm_q/k, to_out; forwards to the processor with signature-filtered kwargs."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 qk_norm=None, eps=1e-5, processor=None, out_bias=True, **unused):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.is_cross_attention = cross_attention_dim is not None
        self.scale = dim_head ** -0.5
        if qk_norm is None:
            self.norm_q = self.norm_k = None
        elif qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps)
        else:  # pragma: no cover
            raise ValueError(qk_norm)
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        names = set(inspect.signature(self.processor.__call__).parameters)
        kw = {k: v for k, v in kw.items() if k in names}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        inner_dim = inner_dim or int(dim * mult)
        dim_out = dim_out or dim
        if activation_fn == "gelu-approximate":
            act = GELU(dim, inner_dim, approximate="tanh", bias=bias)
        elif activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        else:  # pragma: no cover
            raise ValueError(activation_fn)
        layers = [act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)]
        if final_dropout:
            layers.append(nn.Dropout(dropout))
        self.net = nn.ModuleList(layers)

    def forward(self, x, *a, **k):
        for m in self.net:
            x = m(x)
        return x


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, output_dim=None, norm_elementwise_affine=False,
                 norm_eps=1e-5, chunk_dim=0):
        super().__init__()
        assert num_embeddings is None
        self.chunk_dim = chunk_dim
        output_dim = output_dim or embedding_dim * 2
        self.emb = None
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)

    def forward(self, x, timestep=None, temb=None):
        temb = self.linear(self.silu(temb))
        if self.chunk_dim == 1:
            shift, scale = temb.chunk(2, dim=1)
            shift = shift[:, None, :]
            scale = scale[:, None, :]
        else:
            scale, shift = temb.chunk(2, dim=0)
        return self.norm(x) * (1 + scale) + shift


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * randn_tensor(self.mean.shape, generator=generator,
                                                   device=self.parameters.device, dtype=self.parameters.dtype)

    def mode(self):
        return self.mean


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    gen_device = generator.device if generator is not None else (device or "cpu")
    return torch.randn(shape, generator=generator, device=gen_device, dtype=dtype).to(device or gen_device)


# ----------------------------------------------------------------------------------------------
# scheduler
# ----------------------------------------------------------------------------------------------
class FlowMatchEulerDiscreteScheduler(ConfigMixin):
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, base_shift=0.5,
                 max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096):
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, shift=shift,
                                         use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift,
                                         max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                                         max_image_seq_len=max_image_seq_len)
        timesteps = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sigmas = torch.from_numpy(timesteps).to(dtype=torch.float32) / num_train_timesteps
        if not use_dynamic_shifting:
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = sigmas * num_train_timesteps
        self._step_index = None
        self._begin_index = None
        self.sigmas = sigmas.to("cpu")
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    def time_shift(self, mu, sigma, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        if self.config.use_dynamic_shifting and mu is None:
            raise ValueError("you have to pass a value for `mu` when `use_dynamic_shifting` is True")
        if sigmas is None:
            self.num_inference_steps = num_inference_steps
            timesteps = np.linspace(self.sigma_max * self.config.num_train_timesteps,
                                    self.sigma_min * self.config.num_train_timesteps, num_inference_steps)
            sigmas = timesteps / self.config.num_train_timesteps
        if self.config.use_dynamic_shifting:
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            sigmas = self.config.shift * sigmas / (1 + (self.config.shift - 1) * sigmas)
        sigmas = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32, device=device)
        timesteps = sigmas * self.config.num_train_timesteps
        self.timesteps = timesteps.to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        schedule_timesteps = self.timesteps if schedule_timesteps is None else schedule_timesteps
        indices = (schedule_timesteps == timestep).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()

    def scale_noise(self, sample, timestep, noise):
        idx = [self.index_for_timestep(t) for t in timestep]
        sigma = self.sigmas[idx].flatten().to(sample.dtype)
        while sigma.ndim < sample.ndim:
            sigma = sigma.unsqueeze(-1)
        return sigma * noise + (1.0 - sigma) * sample

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, **unused):
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        sigma_next = self.sigmas[self._step_index + 1]
        prev_sample = sample + (sigma_next - sigma) * model_output
        prev_sample = prev_sample.to(model_output.dtype)
        self._step_index += 1
        return (prev_sample,)


# ----------------------------------------------------------------------------------------------
# module registry + auto-stubs for everything the reference imports but never executes on the path
# ----------------------------------------------------------------------------------------------
class _Placeholder(nn.Module):
    """Stands in for any diffusers class the hot path never instantiates."""

    def __init__(self, *a, **k):
        super().__init__()

    def __class_getitem__(cls, item):
        return cls


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def _identity_decorator(fn):
    return fn


def _is_torch_version(op, ver):
    from packaging import version
    import operator
    ops = {">": operator.gt, ">=": operator.ge, "<": operator.lt, "<=": operator.le, "==": operator.eq}
    return ops[op](version.parse(torch.__version__.split("+")[0]), version.parse(ver))


class BaseOutput(dict):
    def __init__(self, **kw):
        super().__init__(**kw)
        self.__dict__.update(kw)

    def __getitem__(self, k):
        if isinstance(k, int):
            return list(self.values())[k]
        return super().__getitem__(k)


class Transformer2DModelOutput(BaseOutput):
    pass


class AutoencoderKLOutput(BaseOutput):
    pass


class DecoderOutput(BaseOutput):
    pass


_REAL = {
    "diffusers": dict(__version__="0.30.1", FlowMatchEulerDiscreteScheduler=FlowMatchEulerDiscreteScheduler),
    "diffusers.configuration_utils": dict(ConfigMixin=ConfigMixin, register_to_config=register_to_config,
                                          FrozenDict=FrozenDict),
    "diffusers.models.modeling_utils": dict(ModelMixin=ModelMixin),
    "diffusers.models.attention": dict(Attention=Attention, FeedForward=FeedForward, GELU=GELU),
    "diffusers.models.attention_processor": dict(Attention=Attention, ADDED_KV_ATTENTION_PROCESSORS=(),
                                                 CROSS_ATTENTION_PROCESSORS=()),
    "diffusers.models.embeddings": dict(Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding,
                                        get_timestep_embedding=get_timestep_embedding,
                                        apply_rotary_emb=apply_rotary_emb,
                                        get_1d_rotary_pos_embed=get_1d_rotary_pos_embed,
                                        get_3d_rotary_pos_embed=get_3d_rotary_pos_embed,
                                        get_2d_sincos_pos_embed=get_2d_sincos_pos_embed),
    "diffusers.models.normalization": dict(AdaLayerNorm=AdaLayerNorm),
    "diffusers.models.autoencoders.vae": dict(DiagonalGaussianDistribution=DiagonalGaussianDistribution,
                                              DecoderOutput=DecoderOutput),
    "diffusers.models.modeling_outputs": dict(Transformer2DModelOutput=Transformer2DModelOutput,
                                              AutoencoderKLOutput=AutoencoderKLOutput),
    "diffusers.utils": dict(USE_PEFT_BACKEND=False, is_torch_version=_is_torch_version, logging=_logging,
                            BaseOutput=BaseOutput, deprecate=lambda *a, **k: None,
                            is_torch_xla_available=lambda: False, is_bs4_available=lambda: False,
                            is_ftfy_available=lambda: False, BACKENDS_MAPPING={},
                            replace_example_docstring=lambda doc: _identity_decorator),
    "diffusers.utils.import_utils": dict(is_xformers_available=lambda: False),
    "diffusers.utils.torch_utils": dict(maybe_allow_in_graph=_identity_decorator, randn_tensor=randn_tensor),
    "diffusers.utils.accelerate_utils": dict(apply_forward_hook=_identity_decorator),
    "diffusers.schedulers": dict(FlowMatchEulerDiscreteScheduler=FlowMatchEulerDiscreteScheduler),
}


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Placeholder,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname == _PREFIX or fullname.startswith(_PREFIX + "."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        for k, v in _REAL.get(spec.name, {}).items():
            setattr(m, k, v)
        return m

    def exec_module(self, module):
        pass


_INSTALLED = False


def install():
    """Make ``import diffusers...`` resolve to this shim.  Refuses to shadow a real install."""
    global _INSTALLED
    if _INSTALLED:
        return
    if "diffusers" in sys.modules and not isinstance(sys.modules["diffusers"], _StubModule):
        raise RuntimeError("a real diffusers is already imported; the shim must not shadow it")
    sys.meta_path.insert(0, _Finder())
    _INSTALLED = True
