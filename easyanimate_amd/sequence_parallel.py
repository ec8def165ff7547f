"""Sequence parallelism over the video-token axis (new capability; the reference has no multi-GPU inference,
SURVEY.md 5.7 / 8e).  One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

Design A of SURVEY 5.7: every rank owns a contiguous shard of the N video tokens (shard stride n_loc, a multiple
of 64; only the last ranks may be short), the T text tokens and all weights are replicated.  All per-token work
(norms, QKV/out/FFN GEMMs, RoPE, residuals) is local.  The only exchange per block is an all-gather of the
K and V^T shards; each rank then attends its own queries (its video rows + the replicated text rows) over the
full key/value set.  One more all-gather returns the velocity prediction to every rank so that all ranks run the
identical scheduler step.

Global row layout of the attention buffers:  [ text 0..T | rank0 video | rank1 video | ... | pad ]  -- padding
only ever sits at the end (rows >= T+N), where ea_attention_fwd_bf16 masks it.

This module contains communication and indexing only (no arithmetic); it works on any device, so the exchange
is covered by gloo/CPU tests (tests/test_sequence_parallel_cpu.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class SequenceParallel:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, comm_stream=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total = 0
        self.n_loc = 0

    def _all_gather_flat(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        """recv[world*n] <- concat over ranks of send[n].  RCCL ("nccl") gathers device buffers directly over xGMI.
        The gloo backend (used only by the CPU / single-GPU tests) cannot gather device tensors into one buffer, so
        device tensors are staged through host memory there."""
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            r = torch.empty(recv.shape, dtype=recv.dtype, device="cpu")
            dist.all_gather_into_tensor(r, send.cpu(), group=self.group)
            recv.copy_(r)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)

    # ---- token partition ----------------------------------------------------------------------
    def plan(self, n_total: int) -> None:
        self.n_total = n_total
        self.n_loc = _round_up((n_total + self.world - 1) // self.world, 64)

    def shard_range(self, rank: Optional[int] = None) -> Tuple[int, int]:
        r = self.rank if rank is None else rank
        lo = min(r * self.n_loc, self.n_total)
        hi = min(lo + self.n_loc, self.n_total)
        return lo, hi

    def shard_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """[B, N, C] -> this rank's [B, n_valid, C] (contiguous)."""
        self.plan(x.shape[1])
        lo, hi = self.shard_range()
        if hi <= lo:
            raise ValueError(f"sequence parallel: rank {self.rank} would own no tokens (N={x.shape[1]}, P={self.world})")
        return x[:, lo:hi].contiguous()

    def shard_rope(self, rope, device):
        cos, sin = rope
        lo, hi = self.shard_range()
        return (cos[lo:hi].to(device=device, dtype=torch.float32).contiguous(),
                sin[lo:hi].to(device=device, dtype=torch.float32).contiguous())

    # ---- attention layout ---------------------------------------------------------------------
    def layout(self, T: int, n_valid: int):
        """-> (S_global, q_begin, q_end, seq_off_video, s_pad) for this rank's video queries."""
        lo, hi = self.shard_range()
        assert hi - lo == n_valid, "hidden-state shard does not match the plan"
        S = T + self.n_total
        s_pad = _round_up(T + self.world * self.n_loc, 256)
        return S, T + lo, T + hi, T + lo, s_pad

    def exchange_kv(self, ws: dict, T: int, n_valid: int) -> None:
        """All-gather the K rows / V^T columns of every rank's video shard into the full-sequence buffers.
        ws["k"]: [B,H,s_pad,64], ws["vt"]: [B,H,64,s_pad]; this rank has already written its own shard."""
        if self.world == 1:
            return
        k, vt = ws["k"], ws["vt"]
        B, H = k.shape[0], k.shape[1]
        nl = self.n_loc
        off = T + self.rank * nl
        send = torch.empty((2, B, H, nl * 64), dtype=k.dtype, device=k.device)
        send[0] = k[:, :, off:off + nl].reshape(B, H, nl * 64)
        send[1] = vt[:, :, :, off:off + nl].reshape(B, H, 64 * nl)
        recv = torch.empty((self.world * send.numel(),), dtype=k.dtype, device=k.device)
        self._all_gather_flat(recv, send.view(-1))  # flat buffers: accepted by nccl/RCCL and gloo
        recv = recv.view((self.world,) + tuple(send.shape))
        for r in range(self.world):
            if r == self.rank:
                continue
            o = T + r * nl
            k[:, :, o:o + nl] = recv[r, 0].reshape(B, H, nl, 64)
            vt[:, :, :, o:o + nl] = recv[r, 1].reshape(B, H, 64, nl)

    def split_output(self, o: torch.Tensor, T: int, n_valid: int):
        lo, hi = self.shard_range()
        return o[:, :T], o[:, T + lo:T + hi]

    def gather_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """[B, n_valid, C] shards -> [B, N, C] on every rank."""
        if self.world == 1:
            return x
        B, n, C = x.shape
        send = torch.zeros((B, self.n_loc, C), dtype=x.dtype, device=x.device)
        send[:, :n] = x
        recv = torch.empty((self.world * send.numel(),), dtype=x.dtype, device=x.device)
        self._all_gather_flat(recv, send.view(-1))
        recv = recv.view(self.world, B, self.n_loc, C)
        return recv.permute(1, 0, 2, 3).reshape(B, self.world * self.n_loc, C)[:, :self.n_total].contiguous()


def enable(transformer, group: Optional[dist.ProcessGroup] = None) -> SequenceParallel:
    """Attach sequence parallelism to an EasyAnimateTransformer3DModel (all ranks hold identical weights)."""
    sp = SequenceParallel(group)
    transformer.sequence_parallel = sp if sp.world > 1 else None
    return sp
