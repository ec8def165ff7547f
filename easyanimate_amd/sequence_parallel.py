"""Multi-GPU sampling: CFG parallelism x sequence parallelism over the video-token axis (new capability; the
reference has no multi-GPU inference).  One process per GPU, `torch.distributed` (backend "nccl"
== RCCL over xGMI).  All weights and the T text tokens are replicated.

Two orthogonal axes, chosen per forward call:

* **CFG axis** (only when the call carries the CFG pair, batch == 2, and the world size is even): ranks
  [0, P/2) run the first batch element, ranks [P/2, P) the second.  Nothing is exchanged between the halves until the
  final velocity prediction -- at P = 2 a denoise step has no per-block communication at all, and at P = 8 every
  per-block exchange moves 2.3x fewer bytes than an 8-way sequence split of both batch elements would.
* **Sequence axis** (the remaining P' = P or P/2 ranks of a half): rank r owns a contiguous shard of the N video
  tokens (shard stride n_loc, a multiple of 64; only the last rank may be short).  All per-token work (norms,
  QKV / out / FFN GEMMs, RoPE, residuals) is local.  The one exchange per block is an all-gather of the K rows /
  V^T columns of the shards.  It is *asynchronous*: while it is in flight the rank already attends its queries over
  its LOCAL keys (text + own shard) with `ea_attention_fwd_range_bf16(..., flags=store state)`; when the remote
  shards have landed a second launch resumes from that state over the REMOTE keys, which it reads straight out of the
  all-gather's output buffer (segment addressing, ea_attention_fwd_segments_bf16: no unpack copies between the two
  passes).  Softmax does not care about the key order, so each rank keeps a private row layout

        [ text 0..T | (gap to a multiple of 64) | own shard | remote shards in rank order | pad ]

  in which the query range and the remote key range are contiguous (the local keys are one range, or two when T is not
  a multiple of 64); padding inside a short last shard is never part of a key range (a range may end anywhere, it only
  has to start on a multiple of 64).

One more all-gather (whole world) returns the velocity prediction of both batch elements / all shards to every rank,
so that all ranks run the identical scheduler step.

The text stream is replicated, not synchronised: its attention rows are accumulated in a rank-dependent key order, so
the replicas agree to fp32-accumulation-order noise, not bit for bit (the values every rank feeds back are rounded to
bf16 each block; measured drift is at the bf16 noise floor of the model).

This module contains communication and indexing only (no arithmetic); it works on any device, so the exchange is
covered by gloo/CPU tests (tests/test_sequence_parallel_cpu.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Layout:
    """Row layout of this rank's attention buffers for one block (see module docstring)."""
    T: int             # text rows [0, T)
    v_off: int         # first row of the own shard = round_up(T, 64): key ranges must start on a multiple of 64
    n_own: int         # valid rows of the own shard
    n_loc: int         # shard stride
    s_pad: int         # rows of the q / k / v^T workspaces
    q_end: int         # queries are rows [0, q_end) (text, [alignment gap,] own shard)
    local_ranges: List[Tuple[int, int]]  # local keys: [0, T + n_own) or, if T % 64 != 0, [0, T) and [v_off, v_off + n_own)
    remote_begin: int  # remote keys [remote_begin, remote_end)   (empty when the sequence axis has one rank)
    remote_end: int


class _Axis:
    """One partition of the world: `groups` sequence-parallel groups of `size` consecutive ranks each."""

    def __init__(self, world: int, rank: int, cfg_degree: int, group):
        self.cfg_degree = cfg_degree
        self.size = world // cfg_degree
        self.cfg_rank = rank // self.size
        self.rank = rank % self.size
        self.group = group


class SequenceParallel:
    # force_exchange (hardware bring-up on a 1-GPU box, tests/test_sequence_parallel_gpu.py): a world of ONE rank still runs
    # the whole multi-rank choreography -- local-key attention pass with stored state, asynchronous all_gather_into_tensor on
    # the process group's stream, work.wait(), second attention pass resuming the state -- by splitting its own keys in two.
    force_exchange = False

    def __init__(self, group: Optional[dist.ProcessGroup] = None, cfg_parallel: bool = True):
        self.world_group = group
        self.world = dist.get_world_size(group)
        self.world_rank = dist.get_rank(group)
        # flat axis: every rank is a sequence shard of every batch element
        self._flat = _Axis(self.world, self.world_rank, 1, group)
        self._cfg = None
        if cfg_parallel and self.world % 2 == 0:
            half = self.world // 2
            base = dist.get_process_group_ranks(group) if group is not None else list(range(self.world))
            mine = None
            for c in range(2):  # every rank creates every sub-group (collective call)
                g = dist.new_group([base[c * half + i] for i in range(half)])
                if self.world_rank // half == c:
                    mine = g
            self._cfg = _Axis(self.world, self.world_rank, 2, mine)
        self.axis = self._flat
        self.n_total = 0
        self.n_loc = 0

    # ---- per-call mode ------------------------------------------------------------------------
    def begin(self, batch: int) -> Tuple[int, int]:
        """Pick the partition for a forward call with `batch` elements; returns this rank's batch slice."""
        if self._cfg is not None and batch == 2:
            self.axis = self._cfg
            return self.axis.cfg_rank, self.axis.cfg_rank + 1
        self.axis = self._flat
        return 0, batch

    @property
    def rank(self) -> int:
        return self.axis.rank

    @property
    def size(self) -> int:
        return self.axis.size

    # ---- token partition ----------------------------------------------------------------------
    def plan(self, n_total: int) -> None:
        self.n_total = n_total
        self.n_loc = _round_up((n_total + self.size - 1) // self.size, 64)

    def shard_range(self, rank: Optional[int] = None) -> Tuple[int, int]:
        r = self.rank if rank is None else rank
        lo = min(r * self.n_loc, self.n_total)
        hi = min(lo + self.n_loc, self.n_total)
        return lo, hi

    def shard_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """[B, N, C] -> this rank's [B, n_own, C] (contiguous)."""
        self.plan(x.shape[1])
        if self.shard_range(self.size - 1)[1] - self.shard_range(self.size - 1)[0] <= 0:
            raise ValueError(f"sequence parallel: the last rank would own no tokens (N={x.shape[1]}, P={self.size})")
        lo, hi = self.shard_range()
        return x[:, lo:hi].contiguous()

    def shard_rope(self, rope, device):
        cos, sin = rope
        lo, hi = self.shard_range()
        return (cos[lo:hi].to(device=device, dtype=torch.float32).contiguous(),
                sin[lo:hi].to(device=device, dtype=torch.float32).contiguous())

    # ---- attention layout ---------------------------------------------------------------------
    def layout(self, T: int, n_own: int) -> Layout:
        lo, hi = self.shard_range()
        assert hi - lo == n_own, "hidden-state shard does not match the plan"
        nl, P = self.n_loc, self.size
        v_off = _round_up(T, 64)
        s_pad = _round_up(v_off + P * nl, 256)
        last = self.shard_range(P - 1)
        n_last = last[1] - last[0]
        remote_valid = 0 if P == 1 else ((P - 1) * nl if self.rank == P - 1 else (P - 2) * nl + n_last)
        local = [(0, T + n_own)] if v_off == T else [(0, T), (v_off, v_off + n_own)]
        if P == 1 and self.force_exchange and v_off == T and T + n_own >= 128:
            # bring-up mode: the second half of the rank's own keys plays the part of the remote shards
            split = (T + n_own) // 2 // 64 * 64
            return Layout(T=T, v_off=v_off, n_own=n_own, n_loc=nl, s_pad=s_pad, q_end=v_off + n_own, local_ranges=[(0, split)],
                          remote_begin=split, remote_end=T + n_own)
        return Layout(T=T, v_off=v_off, n_own=n_own, n_loc=nl, s_pad=s_pad, q_end=v_off + n_own, local_ranges=local,
                      remote_begin=v_off + nl, remote_end=v_off + nl + remote_valid)

    def slot(self, src_rank: int) -> int:
        """Row-slot (in units of n_loc after the text rows) of rank src_rank's shard in THIS rank's buffers."""
        if src_rank == self.rank:
            return 0
        return 1 + (src_rank if src_rank < self.rank else src_rank - 1)

    def _gloo_device_staging(self, t: torch.Tensor) -> bool:
        return t.is_cuda and dist.get_backend(self.axis.group) == "gloo"

    def exchange_start(self, ws: dict, v_off: int):
        """Start the all-gather of every rank's own K rows / V^T columns (rows [v_off, v_off+n_loc) of its buffers).
        Returns a handle for exchange_finish.  With RCCL the collective runs on the process group's stream, behind
        everything already queued on the current stream, and the caller keeps launching compute."""
        if self.size == 1 and not self.force_exchange:
            return None
        k, vt = ws["k"], ws["vt"]
        B, H = k.shape[0], k.shape[1]
        nl = self.n_loc
        send = torch.empty((2, B, H, nl * 64), dtype=k.dtype, device=k.device)
        send[0] = k[:, :, v_off:v_off + nl].reshape(B, H, nl * 64)
        send[1] = vt[:, :, :, v_off:v_off + nl].reshape(B, H, 64 * nl)
        if self._gloo_device_staging(send):
            # gloo (CPU / shared-GPU tests only) cannot gather device tensors into one buffer: stage through the host
            r = torch.empty((self.size * send.numel(),), dtype=send.dtype, device="cpu")
            dist.all_gather_into_tensor(r, send.view(-1).cpu(), group=self.axis.group)
            return (None, r.to(send.device), send)
        recv = torch.empty((self.size * send.numel(),), dtype=k.dtype, device=k.device)
        work = dist.all_gather_into_tensor(recv, send.view(-1), group=self.axis.group, async_op=True)
        return (work, recv, send)

    def exchange_finish(self, handle, ws: dict, v_off: int, unpack: Optional[bool] = None):
        """Wait for the all-gather (a stream-level wait with RCCL).  Returns the gathered buffer viewed as
        [P, 2, B, H, n_loc*64] (per rank: its K rows, then its V^T columns): on the GPU the attention kernel reads the
        remote shards straight out of it (ea_attention_fwd_segments_bf16 -- no unpack copies on the critical path between the
        local-key and the remote-key pass).  unpack=True (the default on CPU tensors: the gloo tests and the oracle
        arithmetic they run) additionally scatters the remote shards into their slots of the rank-private layout."""
        if handle is None:
            return None
        work, recv, send = handle
        if work is not None:
            work.wait()
        if self.size == 1:      # bring-up mode: the collective ran, its result (== what was sent) is not needed
            return None
        k, vt = ws["k"], ws["vt"]
        B, H = k.shape[0], k.shape[1]
        nl = self.n_loc
        recv = recv.view((self.size,) + tuple(send.shape))
        if unpack is None:
            unpack = not k.is_cuda
        if unpack:
            for r in range(self.size):
                if r == self.rank:
                    continue
                o = v_off + self.slot(r) * nl
                k[:, :, o:o + nl] = recv[r, 0].reshape(B, H, nl, 64)
                vt[:, :, :, o:o + nl] = recv[r, 1].reshape(B, H, 64, nl)
        return recv

    def all_reduce_sums(self, sums: torch.Tensor, n: int):
        """TeaCache: add the rel-L1 partial sums / element counts of every rank (batch slices and token shards)."""
        t = torch.cat([sums.to(torch.float64), torch.tensor([float(n)], dtype=torch.float64, device=sums.device)])
        if t.is_cuda and dist.get_backend(self.world_group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, group=self.world_group)
            t = h.to(sums.device)
        else:
            dist.all_reduce(t, group=self.world_group)
        return t[:2], int(round(t[2].item()))

    # ---- final prediction ---------------------------------------------------------------------
    def gather_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """This rank's [b_loc, n_own, C] -> the full [batch, N, C] on every rank (one world-wide all-gather)."""
        if self.world == 1 and not self.force_exchange:
            return x
        b, n, C = x.shape
        send = torch.zeros((b, self.n_loc, C), dtype=x.dtype, device=x.device)
        send[:, :n] = x
        if send.is_cuda and dist.get_backend(self.world_group) == "gloo":
            r = torch.empty((self.world * send.numel(),), dtype=x.dtype, device="cpu")
            dist.all_gather_into_tensor(r, send.view(-1).cpu(), group=self.world_group)
            recv = r.to(x.device)
        else:
            recv = torch.empty((self.world * send.numel(),), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(recv, send.view(-1), group=self.world_group)
        # world rank = cfg_rank * size + seq_rank
        recv = recv.view(self.axis.cfg_degree, self.size, b, self.n_loc, C)
        out = recv.permute(0, 2, 1, 3, 4).reshape(self.axis.cfg_degree * b, self.size * self.n_loc, C)
        return out[:, :self.n_total].contiguous()


class EmulatedRank(SequenceParallel):
    """One GPU runs exactly the per-step COMPUTE of rank `rank` in a world of `world` ranks -- its batch slice (CFG axis),
    its token shard through every per-token kernel, its queries (text + own shard) over all keys in the two-pass
    local / remote form -- with no communication: the remote K rows / V^T columns are random data written once, the final
    gather returns the rank's own prediction tiled.  bench.py --emulate-rank P,r uses it to MODEL the 1 -> P scaling curve
    (speed-up <= T_1 / T_rank) on a one-GPU box; the collectives, their overlap and the pack / unpack copies are NOT in that
    number."""

    def __init__(self, world: int, rank: int, cfg_parallel: bool = True):
        assert 0 <= rank < world
        self.world_group = None
        self.world = world
        self.world_rank = rank
        self._flat = _Axis(world, rank, 1, None)
        self._cfg = _Axis(world, rank, 2, None) if (cfg_parallel and world % 2 == 0) else None
        self.axis = self._flat
        self.n_total = 0
        self.n_loc = 0
        self._filled = {}

    def exchange_start(self, ws: dict, v_off: int):
        if self.size == 1:
            return None
        k = ws["k"]
        key = (tuple(k.shape), self.size, self.n_loc, str(k.device))
        if key not in self._filled:   # the gathered buffer: N(0,1) keys / values, written once (zeros would run at a higher clock)
            self._filled = {key: torch.randn((self.size, 2, k.shape[0], k.shape[1], self.n_loc * 64), device=k.device).to(k.dtype)}
        return self._filled[key]

    def exchange_finish(self, handle, ws: dict, v_off: int, unpack: Optional[bool] = None):
        return handle

    def all_reduce_sums(self, sums: torch.Tensor, n: int):
        return sums.to(torch.float64) * self.world, n * self.world

    def gather_tokens(self, x: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return x
        b, n, C = x.shape
        out = x.new_zeros((self.axis.cfg_degree * b, self.size * self.n_loc, C))
        for c in range(self.axis.cfg_degree):
            for r in range(self.size):
                out[c * b:(c + 1) * b, r * self.n_loc:r * self.n_loc + n] = x
        return out[:, :self.n_total].contiguous()


def enable(transformer, group: Optional[dist.ProcessGroup] = None, cfg_parallel: bool = True,
           force: bool = False) -> SequenceParallel:
    """Attach multi-GPU sampling to an EasyAnimateTransformer3DModel (all ranks hold identical weights).
    force=True keeps the multi-rank code path on even in a world of one rank (bring-up, see force_exchange)."""
    sp = SequenceParallel(group, cfg_parallel=cfg_parallel)
    sp.force_exchange = bool(force)
    transformer.sequence_parallel = sp if (sp.world > 1 or force) else None
    return sp
