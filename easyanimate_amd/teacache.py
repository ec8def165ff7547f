"""TeaCache, device-resident (reference: easyanimate/models/transformer3d.py:90-137 `TeaCache`,
`get_teacache_coefficients`; used at :1564-1590 and :1635).

Same attributes and decision logic as the reference class; what changes is where the data lives: the previous
modulated input and the cached residual stay in HBM (the reference copies two [B,N,d] tensors to the host per step,
~1.3 GB at 49 x 1024^2), the rel-L1 numerator / denominator are reduced by `ea_teacache_rel_l1_bf16`, and 16 bytes
cross PCIe per step.  The skip decision still has to reach the host (it decides which kernels are launched), exactly
like the reference's `.item()`.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import ops


def _bf16_round(x: float) -> float:
    return torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).item()


class TeaCache:
    def __init__(self, coefficients: List[float], num_steps: int, rel_l1_thresh: float = 0.0):
        if num_steps < 1:
            raise ValueError(f"`num_steps` must be greater than 0 but is {num_steps}.")
        if rel_l1_thresh < 0:
            raise ValueError(f"`rel_l1_thresh` must be greater than or equal to 0 but is {rel_l1_thresh}.")
        self.coefficients = coefficients
        self.cnt = 0
        self.num_steps = num_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input: Optional[torch.Tensor] = None
        self.previous_residual: Optional[torch.Tensor] = None
        self.rescale_func = np.poly1d(self.coefficients)
        self.last_rel_l1_distance: Optional[float] = None   # (diagnostics / tests)
        self.last_should_calc: Optional[bool] = None

    @staticmethod
    def compute_rel_l1_distance(prev: torch.Tensor, cur: torch.Tensor, sp=None) -> float:
        """(|cur - prev|.mean() / |prev|.mean()).item() with the roundings torch applies to bf16 tensors (the
        difference per element, each mean, the quotient).  Under multi-GPU sampling the sums run over every rank's
        batch slice / token shard (the reference reduces the whole CFG batch)."""
        sums, n = ops.teacache_rel_l1_sums(cur, prev)
        if sp is not None:
            sums, n = sp.all_reduce_sums(sums, n)
        s = sums.cpu()  # 16 bytes; the only host synchronisation of a TeaCache step
        a, b = _bf16_round(s[0].item() / n), _bf16_round(s[1].item() / n)
        return _bf16_round(a / b)

    def reset(self):
        self.cnt = 0
        self.previous_modulated_input = None
        self.previous_residual = None

    def should_calc(self, modulated_inp: torch.Tensor, sp=None) -> bool:
        """transformer3d.py:1569-1584, verbatim control flow."""
        if self.cnt == 0 or self.cnt == self.num_steps - 1:
            should_calc = True
            self.accumulated_rel_l1_distance = 0
            self.last_rel_l1_distance = None
        else:
            rel = self.compute_rel_l1_distance(self.previous_modulated_input, modulated_inp, sp)
            self.last_rel_l1_distance = rel
            self.accumulated_rel_l1_distance += self.rescale_func(rel)
            if self.accumulated_rel_l1_distance < self.rel_l1_thresh:
                should_calc = False
            else:
                should_calc = True
                self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = modulated_inp
        self.cnt += 1
        if self.cnt == self.num_steps:
            self.reset()
        self.last_should_calc = should_calc
        return should_calc


def get_teacache_coefficients(model_name: str):
    """reference: transformer3d.py:124-137 (coefficients fitted by the EasyAnimate authors)."""
    if "v5.1-7b" in model_name.lower():
        return [1.07862322, -4.19362456, 3.06725828, 0.33161686, 0.02374758]
    elif "v5.1-12b" in model_name.lower():
        return [-10.47857366, 8.33844143, -0.78477557, 0.68798618, 0.0136149]
    print(f"The model {model_name} is not supported by TeaCache.")
    return None
