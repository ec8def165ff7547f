"""FlowMatchEulerDiscreteScheduler with the diffusers 0.30/0.31 interface the pipelines use
(set_timesteps / timesteps / sigmas / step / scale_noise).  The sigma schedule is
host arithmetic (51 floats); the state update runs in ea_cfg_euler_step."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from .config import ConfigMixin, FrozenDict


class FlowMatchEulerDiscreteScheduler(ConfigMixin):
    order = 1
    config_name = "scheduler_config.json"

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False,
                 base_shift: float = 0.5, max_shift: float = 1.15, base_image_seq_len: int = 256,
                 max_image_seq_len: int = 4096):
        object.__setattr__(self, "_internal_dict", FrozenDict(
            num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting,
            base_shift=base_shift, max_shift=max_shift, base_image_seq_len=base_image_seq_len,
            max_image_seq_len=max_image_seq_len))
        timesteps = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sigmas = torch.from_numpy(timesteps).to(torch.float32) / num_train_timesteps
        if not use_dynamic_shifting:
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = sigmas * num_train_timesteps
        self.sigmas = sigmas
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self._step_index: Optional[int] = None
        self._begin_index: Optional[int] = None
        self._sigmas_host = self.sigmas.tolist()
        self.num_inference_steps = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        cfg = cls.load_config(path, subfolder)
        cfg.update(kw)
        return cls.from_config(cfg)

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def time_shift(self, mu: float, sigma: float, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas=None, mu: Optional[float] = None):
        if self.config.use_dynamic_shifting and mu is None:
            raise ValueError("you have to pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            self.num_inference_steps = num_inference_steps
            timesteps = np.linspace(self.sigma_max * self.config.num_train_timesteps,
                                    self.sigma_min * self.config.num_train_timesteps, num_inference_steps)
            sigmas = timesteps / self.config.num_train_timesteps
        else:
            sigmas = np.asarray(sigmas, dtype=np.float64)
            self.num_inference_steps = len(sigmas)
        if self.config.use_dynamic_shifting:
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            sigmas = self.config.shift * sigmas / (1 + (self.config.shift - 1) * sigmas)
        sigmas_t = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32)
        timesteps = sigmas_t * self.config.num_train_timesteps
        full = torch.cat([sigmas_t, torch.zeros(1)])
        self._sigmas_host = full.tolist()  # fp32 values, read on the host: no device sync inside the loop
        self._timesteps_host = timesteps.tolist()
        self.timesteps = timesteps.to(device=device)
        self.sigmas = full.to(device=device)
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        ts = self._timesteps_host if schedule_timesteps is None else [float(x) for x in schedule_timesteps]
        t = float(timestep)
        idx = [i for i, v in enumerate(ts) if v == t]
        if not idx:
            raise ValueError(f"timestep {t} is not in the schedule")
        return idx[1] if len(idx) > 1 else idx[0]

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self._begin_index is None else self._begin_index

    def dsigma(self) -> float:
        """sigma_{i+1} - sigma_i for the current step, in fp32 arithmetic like the reference tensors."""
        i = self._step_index
        return float(np.float32(self._sigmas_host[i + 1]) - np.float32(self._sigmas_host[i]))

    def scale_noise(self, sample: torch.Tensor, timestep, noise: torch.Tensor) -> torch.Tensor:
        ts = timestep.tolist() if isinstance(timestep, torch.Tensor) else [timestep]
        if not isinstance(ts, list):
            ts = [ts]
        sig = torch.tensor([self._sigmas_host[self.index_for_timestep(t)] for t in ts], dtype=sample.dtype,
                           device=sample.device)
        while sig.dim() < sample.dim():
            sig = sig.unsqueeze(-1)
        return sig * noise + (1.0 - sig) * sample

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None, return_dict: bool = True,
             guidance_scale: Optional[float] = None, guidance_rescale: float = 0.0, **unused):
        """x_prev = float(x) + (sigma_next - sigma) * v, cast to v.dtype.  If `guidance_scale` is given,
        model_output holds the [uncond, text] pair and the CFG combine (and, with guidance_rescale > 0, rescale_noise_cfg)
        is fused into the same launch sequence."""
        from . import ops
        if self._step_index is None:
            self._init_step_index(timestep)
        ds = self.dsigma()
        v = model_output.contiguous()
        prev = sample.to(v.dtype).contiguous().clone()
        if guidance_scale is not None and guidance_rescale > 0.0:
            ops.cfg_rescale_euler_step(v, prev, guidance_scale, ds, guidance_rescale)
        else:
            ops.cfg_euler_step(v, prev, guidance_scale if guidance_scale is not None else 0.0, ds,
                               guidance_scale is not None)
        self._step_index += 1
        return (prev,) if not return_dict else FrozenDict(prev_sample=prev)
