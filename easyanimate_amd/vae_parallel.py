"""Exact multi-GPU evaluation of the causal 3-D VAE by a TEMPORAL split, optionally composed with a SPATIAL split (new
capability: SURVEY.md 8(f) rank 4; the reference decodes on one GPU, and its own spatial tiling -- autoencoder_magvit.py:
339-448 -- is not exact).

Time first: under the V5 / V5.1 settings GroupNorm is per frame and the mid-block attention is per frame
(common.py:301-305, vaemodules/attention.py:391-423), so the ONLY operators that look across frames are the causal
3x3x3 convolutions -- and they only look BACKWARDS, by two frames (one for the stride-2 down-samplers).  A rank that owns a
contiguous range of frames therefore needs, per convolution, the last two (one) input frames of its left neighbour and
nothing else: no all-reduce of statistics, no K/V exchange, no halo in space.

  * partition: the latent frames [0, T_lat) in contiguous ranges of at least two frames (P_t = min(world / P_s, T_lat // 2)
    temporal ranks are active, the rest only join the final gather).  Through a temporal x2 up-sampler a range [a, b) becomes
    [2a-1, 2b-1) (frame 0 is never duplicated, upsamplers.py:146-152), through a stride-2 down-sampler the inverse -- so the
    ranges stay contiguous at every resolution of the encoder and the decoder;
  * per causal convolution: receive the neighbour's last frames (one point-to-point message), prepend them, convolve, drop
    the outputs that belong to the halo (2; 3 under the temporal duplication; 1 for stride 2).  The first rank keeps the
    kernel's replicate padding.  Every retained output sees exactly the inputs it sees in the whole-clip evaluation.

Space second (`spatial` = P_s > 1; world = P_t x P_s, the P_s ranks of a temporal range are consecutive): 13 latent frames
keep only 6 temporal ranks busy, so an 8-GPU node runs 4 (time) x 2 (rows).  A rank owns H / P_s consecutive rows of its
frames at every resolution.  Then
  * every 3x3x3 convolution also exchanges ONE input row with the row neighbours above / below (stride-2 layers: only from
    below, downsamplers.py:44-46 pad on the high side), convolves the extended slab with the kernel's own zero padding at
    the outer edges, and keeps the rows it owns (x2 under the folded nearest up-sampling);
  * GroupNorm (per frame, over the WHOLE frame) all-reduces its additive statistics -- (sum, sum of squares) per frame and
    group, in fp64 -- over the P_s ranks of the frame, then normalises locally;
  * the mid-block attention (per frame, one head over all H x W tokens) all-gathers the normalised tokens of the frame over
    the P_s ranks (at latent resolution: 16 MB per frame at 1024^2) and attends its own rows' queries over all keys.
Every retained value is computed from exactly the inputs of the whole-clip evaluation; only the fp64 order of the GroupNorm
sums differs.

Communication and indexing only (no arithmetic): works on any device; covered by gloo tests on CPU tensors
(tests/test_vae_parallel_cpu.py) and by ranks sharing the one GPU of the test box (tests/test_vae_parallel_gpu.py).
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

_ACTIVE: Optional["TemporalParallel"] = None
_SUBGROUPS: dict = {}
_WORLD_REF = None      # weakref to the default process group _SUBGROUPS was filled under


def _subgroups(rank_lists: List[List[int]], parent: Optional[dist.ProcessGroup]) -> dist.ProcessGroup:
    """Partition `parent` (None = the world) into the sub-groups `rank_lists` (global rank numbers) and return the one this rank
    belongs to.  Groups are created ONCE per process and reused by every later enable_* call (no communicator leak).

    * parent spans the world: every rank creates every sub-group, in the same order -- `dist.new_group` is collective over the
      default group, and with an eagerly initialised RCCL world (`init_process_group(..., device_id=...)`, what bench.py does) it is
      an ncclCommSplit in which the NON-members take part too.  Member-local creation would hang there.
    * parent is a strict sub-group (one CFG half, one replica of a serving node): the ranks outside it never get here, so a
      world-collective call cannot be made; the groups are created member-locally (`use_local_synchronization=True`) -- possible
      only when the default group has no eagerly bound communicator to split from; otherwise the caller has to create the groups
      up front, world-collectively."""
    # The cache belongs to ONE world: the default-group OBJECT it was filled under, held by weak reference and compared by identity.
    # (Not its name -- c10d restarts its group counter when the world is destroyed, so every default group is named "0" -- and not a
    # bare id(), which a later object may reuse)  A world that is gone takes its sub-groups with it; cached groups are
    # also checked against c10d's own registry before they are handed out again.
    global _WORLD_REF
    default = dist.distributed_c10d._get_default_group()
    if _WORLD_REF is None or _WORLD_REF() is not default:
        _SUBGROUPS.clear()
        _WORLD_REF = weakref.ref(default)
    key = tuple(tuple(r) for r in rank_lists)
    pg_map = dist.distributed_c10d._world.pg_map
    if key in _SUBGROUPS and any(isinstance(g, dist.ProcessGroup) and g not in pg_map for g in _SUBGROUPS[key].values()):
        del _SUBGROUPS[key]
    me = dist.get_rank()
    if key not in _SUBGROUPS:
        world = dist.get_world_size()
        spans_world = parent is None or dist.get_world_size(parent) == world
        made = {}
        if spans_world:
            for ranks in rank_lists:
                made[tuple(ranks)] = dist.new_group(list(ranks))
        else:
            if getattr(dist.distributed_c10d._get_default_group(), "bound_device_id", None) is not None:
                raise NotImplementedError("sub-groups of a strict sub-group cannot be created member-locally when the default process group was "
                                          "initialised with device_id (communicator split is world-collective): create them up front")
            for ranks in rank_lists:
                if me in ranks:
                    made[tuple(ranks)] = dist.new_group(list(ranks), use_local_synchronization=True)
        _SUBGROUPS[key] = made
    for ranks, g in _SUBGROUPS[key].items():
        if me in ranks:
            return g
    raise RuntimeError(f"rank {me} is in none of the sub-groups {rank_lists}")


def current() -> Optional["TemporalParallel"]:
    return _ACTIVE


class TemporalParallel:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, spatial: int = 1):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if spatial < 1 or self.world % spatial:
            raise ValueError(f"the spatial degree ({spatial}) must divide the world size ({self.world})")
        self.ps = spatial                      # ranks along the rows of a frame
        self.pt = self.world // spatial        # temporal ranks
        self.rank_t, self.rank_s = self.rank // spatial, self.rank % spatial
        self.active_ranks = self.pt            # ACTIVE temporal ranks (set by plan())
        self.messages = 0            # temporal halo messages received (tests / diagnostics)
        self.row_messages = 0        # row halo messages received
        self.row_group = None
        if spatial > 1:
            base = dist.get_process_group_ranks(group) if group is not None else list(range(self.world))
            self.row_group = _subgroups([[base[t * spatial + i] for i in range(spatial)] for t in range(self.pt)], group)

    # ---- partition ---------------------------------------------------------------------------------------------
    def plan(self, latent_frames: int) -> List[Tuple[int, int]]:
        """Latent-frame range [a, b) of every WORLD rank (the P_s ranks of a temporal rank share it); inactive ranks get (T, T)."""
        p = max(1, min(self.pt, latent_frames // 2))
        self.active_ranks = p
        base, extra = divmod(latent_frames, p)
        out, a = [], 0
        for r in range(self.pt):
            n = (base + (1 if r < extra else 0)) if r < p else 0
            out.extend([(a, a + n)] * self.ps)
            a += n
        return out

    def rows(self, H: int, rank_s: Optional[int] = None) -> Tuple[int, int]:
        """Row range of a spatial rank in a frame of H rows."""
        if H % self.ps:
            raise ValueError(f"spatially split VAE: {H} rows do not divide over {self.ps} ranks")
        r = self.rank_s if rank_s is None else rank_s
        return r * (H // self.ps), (r + 1) * (H // self.ps)

    @staticmethod
    def finer(rng: Tuple[int, int]) -> Tuple[int, int]:
        """The frame range one temporal level up (x2 in time, first frame single): [a, b) -> [2a-1, 2b-1), 0 stays 0."""
        a, b = rng
        return (2 * a - 1 if a > 0 else 0, 2 * b - 1 if b > 0 else 0)

    @property
    def is_active(self) -> bool:
        return self.rank_t < self.active_ranks

    # ---- halo exchange -----------------------------------------------------------------------------------------
    def exchange(self, x: torch.Tensor, n: int) -> Optional[torch.Tensor]:
        """Send this rank's last n frames of x [T, ...] to the right neighbour, receive the left neighbour's.
        Returns the received halo [n, ...], or None on the first rank."""
        send_to = self.rank + self.ps if self.rank_t + 1 < self.active_ranks else None      # same rows, next frame range
        recv_from = self.rank - self.ps if self.rank_t > 0 else None
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

    def exchange_rows(self, x: torch.Tensor, n_above: int, n_below: int):
        """Row halo of x [T, H_loc, W, C]: returns (the upper neighbour's last n_above rows or None, the lower neighbour's
        first n_below rows or None); sends this rank's last n_above rows down and its first n_below rows up."""
        if self.ps == 1:
            return None, None
        up = self.rank - 1 if self.rank_s > 0 else None
        down = self.rank + 1 if self.rank_s + 1 < self.ps else None
        staged = x.is_cuda and dist.get_backend(self.group) == "gloo"
        dev = "cpu" if staged else x.device
        ops_, above, below = [], None, None
        mk = lambda t: t.contiguous().cpu() if staged else t.contiguous()
        if up is not None and n_below > 0:
            ops_.append(dist.P2POp(dist.isend, mk(x[:, :n_below]), self._global(up), self.group))
        if down is not None and n_above > 0:
            ops_.append(dist.P2POp(dist.isend, mk(x[:, -n_above:]), self._global(down), self.group))
        if up is not None and n_above > 0:
            above = torch.empty((x.shape[0], n_above) + tuple(x.shape[2:]), dtype=x.dtype, device=dev)
            ops_.append(dist.P2POp(dist.irecv, above, self._global(up), self.group))
        if down is not None and n_below > 0:
            below = torch.empty((x.shape[0], n_below) + tuple(x.shape[2:]), dtype=x.dtype, device=dev)
            ops_.append(dist.P2POp(dist.irecv, below, self._global(down), self.group))
        if ops_:
            for w in dist.batch_isend_irecv(ops_):
                w.wait()
        self.row_messages += (above is not None) + (below is not None)
        if staged:
            above = None if above is None else above.to(x.device)
            below = None if below is None else below.to(x.device)
        return above, below

    def all_reduce_rows(self, t: torch.Tensor) -> torch.Tensor:
        """Sum over the P_s ranks that share this rank's frames (GroupNorm statistics)."""
        if self.ps == 1:
            return t
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, group=self.row_group)
            return h.to(t.device)
        dist.all_reduce(t, group=self.row_group)
        return t

    def all_gather_rows(self, x: torch.Tensor, dim: int) -> torch.Tensor:
        """Concatenate the P_s ranks' equally sized slabs along `dim` (tokens of a frame for the mid-block attention)."""
        if self.ps == 1:
            return x
        staged = x.is_cuda and dist.get_backend(self.group) == "gloo"
        src = x.contiguous().cpu() if staged else x.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.ps)]
        dist.all_gather(parts, src, group=self.row_group)
        out = torch.cat(parts, dim=dim)
        return out.to(x.device) if staged else out

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
    def gather_frames(self, x: Optional[torch.Tensor], ranges: List[Tuple[int, int]], frame_dim: int, like: torch.Tensor,
                      row_dim: Optional[int] = None) -> torch.Tensor:
        """Concatenate the ranks' frame ranges along frame_dim (and, under a spatial split, the row slabs along row_dim) on
        every rank.  x: this rank's block (None if inactive); `like` gives the FULL-frame shape (with any frame count) /
        dtype / device."""
        total = max(b for _, b in ranges)
        mx = max(b - a for a, b in ranges)
        shape = list(like.shape)
        shape[frame_dim] = mx
        if self.ps > 1:
            assert row_dim is not None
            lo, hi = self.rows(shape[row_dim])
            shape[row_dim] = hi - lo
        buf = torch.zeros(shape, dtype=like.dtype, device=like.device)
        if x is not None and x.shape[frame_dim] > 0:
            buf.narrow(frame_dim, 0, x.shape[frame_dim]).copy_(x)
        staged = buf.is_cuda and dist.get_backend(self.group) == "gloo"
        src = buf.cpu() if staged else buf
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src.contiguous(), group=self.group)
        if self.ps > 1:   # the P_s consecutive ranks of a temporal range: their row slabs side by side
            parts = [torch.cat(parts[i:i + self.ps], dim=row_dim) for i in range(0, self.world, self.ps)]
            ranges = ranges[::self.ps]
        out = torch.cat([p.narrow(frame_dim, 0, b - a) for p, (a, b) in zip(parts, ranges) if b > a], dim=frame_dim)
        assert out.shape[frame_dim] == total
        return out.to(like.device) if staged else out


class SelfLoopTemporal(TemporalParallel):
    """Hardware bring-up on a ONE-GPU box: a world of one rank plays `virtual` temporal ranks one after the other, and every
    frame halo travels through the process group's point-to-point path as a send + receive TO ITSELF (one batched
    isend / irecv pair per causal convolution -- with backend "nccl" that is RCCL's ncclSend / ncclRecv on device tensors,
    no host staging), exactly the call sequence a real neighbour pair issues.  The convolutions of virtual rank v run after
    those of rank v - 1, which left the tails they would have sent in a queue (same convolution order in both).
    tests/test_vae_parallel_gpu.py::test_rccl_point_to_point_world_of_one."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, virtual: int = 2):
        super().__init__(group, spatial=1)
        if self.world != 1:
            raise ValueError("SelfLoopTemporal is a world-of-one bring-up mode")
        self.pt = virtual
        self.virtual_rank = 0
        self._tails = []        # tails left by the previous virtual rank, in convolution order
        self._next = []         # tails this virtual rank leaves for the next one
        self._cursor = 0

    def plan(self, latent_frames: int) -> List[Tuple[int, int]]:
        return [r for r in super().plan(latent_frames)]

    def enter(self, v: int) -> None:
        """Start the pass of virtual rank v (0, 1, ... in order)."""
        self.virtual_rank = self.rank_t = v
        self._tails, self._next, self._cursor = self._next, [], 0

    def exchange(self, x: torch.Tensor, n: int) -> Optional[torch.Tensor]:
        if self.virtual_rank + 1 < self.active_ranks:
            if x.shape[0] < n:
                raise ValueError(f"temporal parallel VAE: virtual rank {self.virtual_rank} owns {x.shape[0]} frames, fewer than the halo of {n}")
            self._next.append(x[-n:].contiguous().clone())
        if self.virtual_rank == 0:
            return None
        tail = self._tails[self._cursor]
        self._cursor += 1
        assert tail.shape[0] == n and tuple(tail.shape[1:]) == tuple(x.shape[1:]), "convolution order differs between the virtual ranks"
        self.messages += 1
        if dist.get_backend(self.group) == "gloo":
            return tail.clone()       # gloo has no send-to-self: the CPU test covers the queue bookkeeping only
        recv = torch.empty_like(tail)
        me = self._global(0)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, tail, me, self.group), dist.P2POp(dist.irecv, recv, me, self.group)]):
            w.wait()
        return recv


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
