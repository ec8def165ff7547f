"""Exact multi-GPU evaluation of the causal 3-D VAE by a TEMPORAL split (new capability: SURVEY.md 8(f) rank 4; the
reference decodes on one GPU, and its own spatial tiling -- autoencoder_magvit.py:339-448 -- is not exact).

Why time and not space: under the V5 / V5.1 settings GroupNorm is per frame and the mid-block attention is per frame
(common.py:301-305, vaemodules/attention.py:391-423), so the ONLY operators that look across frames are the causal
3x3x3 convolutions -- and they only look BACKWARDS, by two frames (one for the stride-2 down-samplers).  A rank that owns a
contiguous range of frames therefore needs, per convolution, the last two (one) input frames of its left neighbour and
nothing else: no all-reduce of statistics, no K/V exchange, no halo in space.

  * partition: the latent frames [0, T_lat) in contiguous ranges of at least two frames (P' = min(world, T_lat // 2) ranks
    are active, the rest only join the final gather).  Through a temporal x2 up-sampler a range [a, b) becomes
    [2a-1, 2b-1) (frame 0 is never duplicated, upsamplers.py:146-152), through a stride-2 down-sampler the inverse -- so the
    ranges stay contiguous at every resolution of the encoder and the decoder;
  * per causal convolution: receive the neighbour's last frames (one point-to-point message), prepend them, convolve, drop
    the outputs that belong to the halo (2; 3 under the temporal duplication; 1 for stride 2).  The first rank keeps the
    kernel's replicate padding.  Every retained output sees exactly the inputs it sees in the whole-clip evaluation.

Communication and indexing only (no arithmetic): works on any device; covered by gloo tests on CPU tensors
(tests/test_vae_parallel_cpu.py) and by ranks sharing the one GPU of the test box (tests/test_vae_parallel_gpu.py).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

_ACTIVE: Optional["TemporalParallel"] = None


def current() -> Optional["TemporalParallel"]:
    return _ACTIVE


class TemporalParallel:
    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.active_ranks = self.world
        self.messages = 0            # halo messages received (tests / diagnostics)

    # ---- partition ---------------------------------------------------------------------------------------------
    def plan(self, latent_frames: int) -> List[Tuple[int, int]]:
        """Latent-frame range [a, b) of every rank; inactive ranks get (T, T)."""
        p = max(1, min(self.world, latent_frames // 2))
        self.active_ranks = p
        base, extra = divmod(latent_frames, p)
        out, a = [], 0
        for r in range(self.world):
            n = (base + (1 if r < extra else 0)) if r < p else 0
            out.append((a, a + n))
            a += n
        return out

    @staticmethod
    def finer(rng: Tuple[int, int]) -> Tuple[int, int]:
        """The frame range one temporal level up (x2 in time, first frame single): [a, b) -> [2a-1, 2b-1), 0 stays 0."""
        a, b = rng
        return (2 * a - 1 if a > 0 else 0, 2 * b - 1 if b > 0 else 0)

    @property
    def is_active(self) -> bool:
        return self.rank < self.active_ranks

    # ---- halo exchange -----------------------------------------------------------------------------------------
    def exchange(self, x: torch.Tensor, n: int) -> Optional[torch.Tensor]:
        """Send this rank's last n frames of x [T, ...] to the right neighbour, receive the left neighbour's.
        Returns the received halo [n, ...], or None on the first rank."""
        send_to = self.rank + 1 if self.rank + 1 < self.active_ranks else None
        recv_from = self.rank - 1 if self.rank > 0 else None
        if x.shape[0] < n and send_to is not None:
            raise ValueError(f"temporal parallel VAE: rank {self.rank} owns {x.shape[0]} frames, fewer than the halo of {n}")
        staged = x.is_cuda and dist.get_backend(self.group) == "gloo"      # gloo moves host tensors (tests on a shared GPU)
        ops_, recv = [], None
        if send_to is not None:
            tail = x[-n:].contiguous()
            ops_.append(dist.P2POp(dist.isend, tail.cpu() if staged else tail, self._global(send_to), self.group))
        if recv_from is not None:
            recv = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device="cpu" if staged else x.device)
            ops_.append(dist.P2POp(dist.irecv, recv, self._global(recv_from), self.group))
        if ops_:
            for w in dist.batch_isend_irecv(ops_):
                w.wait()
        if recv is not None:
            self.messages += 1
            return recv.to(x.device) if staged else recv
        return None

    def _global(self, r: int) -> int:
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def halo_frames(self, temporal_stride: int) -> int:
        return 2 if temporal_stride == 1 else 1

    @staticmethod
    def dropped_outputs(temporal_stride: int, halo: int, tdup: bool) -> int:
        """Outputs computed from the halo's positions: `halo` for stride 1 (2*halo - 1 when every frame but the tensor's
        first is stored twice), 1 for stride 2 with its one-frame halo."""
        if temporal_stride == 2:
            return 1
        return 2 * halo - 1 if tdup else halo

    # ---- gather ------------------------------------------------------------------------------------------------
    def gather_frames(self, x: Optional[torch.Tensor], ranges: List[Tuple[int, int]], frame_dim: int, like: torch.Tensor) -> torch.Tensor:
        """Concatenate the ranks' frame ranges along frame_dim on every rank.  x: this rank's frames (None if inactive);
        `like` gives shape (with any frame count) / dtype / device."""
        total = ranges[-1][1] if ranges else 0
        total = max(b for _, b in ranges)
        mx = max(b - a for a, b in ranges)
        shape = list(like.shape)
        shape[frame_dim] = mx
        buf = torch.zeros(shape, dtype=like.dtype, device=like.device)
        if x is not None and x.shape[frame_dim] > 0:
            buf.narrow(frame_dim, 0, x.shape[frame_dim]).copy_(x)
        staged = buf.is_cuda and dist.get_backend(self.group) == "gloo"
        src = buf.cpu() if staged else buf
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src.contiguous(), group=self.group)
        out = torch.cat([p.narrow(frame_dim, 0, b - a) for p, (a, b) in zip(parts, ranges) if b > a], dim=frame_dim)
        assert out.shape[frame_dim] == total
        return out.to(like.device) if staged else out


class activate:
    """Context manager: the convolutions of vae_modules consult `current()` while a split encode / decode runs."""

    def __init__(self, tp: Optional[TemporalParallel]):
        self.tp = tp

    def __enter__(self):
        global _ACTIVE
        self.prev, _ACTIVE = _ACTIVE, self.tp
        return self.tp

    def __exit__(self, *a):
        global _ACTIVE
        _ACTIVE = self.prev
