"""Attention processor of the MMDiT block -- the narrowest drop-in seam.

Same protocol as /root/reference/easyanimate/models/processor.py:218-312:
    processor(attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None, attn2=None)
        -> (hidden_states, encoder_hidden_states)
plus four optional keyword arguments (residual / gate for both streams) that let the caller fuse the gated
residual add of attention.py:1140-1141 into the output-projection GEMM epilogue.  diffusers' Attention.forward
filters kwargs by this signature, so the extra names are protocol-compatible.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from . import ops
from ._params import f32, gemm_weight

_ws: Dict[tuple, dict] = {}


def _workspace(B: int, H: int, s_pad: int, device) -> dict:
    """q/k/v^T staging buffers, zero-initialised once: rows >= seq are never written, so they stay zero
    (ea_attention_fwd_bf16 requires a finite V^T tail)."""
    key = (B, H, s_pad, str(device))
    ws = _ws.get(key)
    if ws is None:
        for k in [k for k in _ws if k[0] not in ("state", "q", "heads", "vtp")]:  # one live shape at a time keeps the footprint bounded
            del _ws[k]
        ws = dict(q=torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=device),
                  k=torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=device),
                  vt=torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=device))
        _ws[key] = ws
    return ws


def _workspace_q(B: int, H: int, q_pad: int, device) -> torch.Tensor:
    """q staging buffer of the sequence-parallel path (K / V^T live in the exchange buffer), zero-initialised once."""
    key = ("q", B, H, q_pad, str(device))
    q = _ws.get(key)
    if q is None:
        for k in [k for k in _ws if k[0] == "q"]:
            del _ws[k]
        q = torch.zeros(B, H, q_pad, 64, dtype=torch.bfloat16, device=device)
        _ws[key] = q
    return q


_ws_live: dict = {}      # workspace key -> live length of the last call (rows / columns behind it are zero)


def _workspace_heads(B: int, Hl: int, s_pad: int, device, dtype=torch.bfloat16, live: int = None):
    """q / k / v^T of ALL tokens for this rank's heads (head-parallel attention under sequence parallelism), zero-initialised
    once and reused by every block.  Rows >= `live` (the caller's T + Nt) are zero: the workspace is keyed by the PADDED length,
    so when a later call has a shorter sequence with the same padding, the rows the longer one wrote are zeroed again."""
    key = ("heads", B, Hl, s_pad, str(device), dtype)
    w = _ws.get(key)
    if w is None:
        for k in [k for k in _ws if k[0] == "heads"]:
            del _ws[k]
            _ws_live.pop(k, None)
        w = (torch.zeros(B, Hl, s_pad, 64, dtype=dtype, device=device), torch.zeros(B, Hl, s_pad, 64, dtype=dtype, device=device),
             torch.zeros(B, Hl, 64, s_pad, dtype=dtype, device=device))
        _ws[key] = w
    elif live is not None and live < _ws_live.get(key, s_pad):
        prev = _ws_live.get(key, s_pad)
        w[0][:, :, live:prev].zero_()
        w[1][:, :, live:prev].zero_()
        w[2][:, :, :, live:prev].zero_()
    if live is not None:
        _ws_live[key] = live
    return w


def _workspace_vt_perm(B: int, Hl: int, n_pad: int, device, live: int = None) -> torch.Tensor:
    """V^T in scan order for the sliding-window pass, zero-initialised once (columns behind the sequence must be finite); columns
    >= `live` (the caller's N) are zeroed again when a shorter sequence reuses the same padded workspace."""
    key = ("vtp", B, Hl, n_pad, str(device))
    w = _ws.get(key)
    if w is None:
        for k in [k for k in _ws if k[0] == "vtp"]:
            del _ws[k]
            _ws_live.pop(k, None)
        w = torch.zeros(B, Hl, 64, n_pad, dtype=torch.bfloat16, device=device)
        _ws[key] = w
    elif live is not None and live < _ws_live.get(key, n_pad):
        w[:, :, :, live:_ws_live.get(key, n_pad)].zero_()
    if live is not None:
        _ws_live[key] = live
    return w


def _attention_state(B: int, H: int, q_end: int, device) -> torch.Tensor:
    """fp32 scratch of the resumable attention, one live shape at a time."""
    key = ("state", B, H, q_end, str(device))
    st = _ws.get(key)
    if st is None:
        for k in [k for k in _ws if k[0] == "state"]:
            del _ws[k]
        st = ops.attention_state(B, H, 0, q_end, device)
        _ws[key] = st
    return st


def rope_to_device(image_rotary_emb, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 contiguous device tables.  The reference moves the CPU cos/sin tables to the device inside every attention
    call (diffusers apply_rotary_emb); here EasyAnimatePipeline.rotary_embedding uploads them once per pipeline call and
    EasyAnimateTransformer3DModel.forward once per forward for callers that pass CPU tables, so this is a no-op on the
    hot path.  Nothing is cached by address: tables of equal size but different content (384x672 vs 672x384) must never
    alias."""
    cos, sin = image_rotary_emb
    if cos.is_cuda and cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_cuda and sin.dtype == torch.float32 \
            and sin.is_contiguous():
        return cos, sin
    return (cos.to(device=device, dtype=torch.float32).contiguous(), sin.to(device=device, dtype=torch.float32).contiguous())


def _bf16c(x):
    x = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
    return x if x.is_contiguous() else x.contiguous()


class EasyAnimateAttnProcessor2_0:
    fuse_qkv = os.environ.get("EA_FUSE_QKV", "1") != "0"   # False / EA_FUSE_QKV=0: always take the three-GEMM + ea_qknorm_rope_bf16 path (same roundings; tests)

    exchange_in_attend = False   # the SWA processor exchanges differently (heads, not keys): it starts nothing here

    def __init__(self):
        pass

    kv_first = os.environ.get("EA_SP_KV_FIRST", "1") != "0"  # sequence parallel: project K | V first, start the exchange, project Q under it

    def _head_parallel(self, ws, B, H, T, S, d, dev, lay, sp, attend_heads):
        """Sequence parallelism by HEAD all-to-all (SURVEY 5.7 design C, "Ulysses"): every rank sends the q / k / v^T rows of its
        token shard for the heads of rank g to rank g, receives the rows of ALL video tokens for its own H / P' heads, runs
        `attend_heads(qf, kf, vf, Hl, head0, Nt) -> [B, T + Nt, Hl * 64]` on contiguous single-GPU operands (rows = [text 0..T |
        all Nt video tokens]), and a second all-to-all returns every rank's query rows for all heads.  The replicated text rows
        are computed by each head's owner and all-gathered, so they are bit-identical on every rank."""
        P = sp.size
        if H % P:
            raise NotImplementedError(f"head-parallel attention under sequence parallelism needs heads ({H}) % sequence ranks ({P}) == 0")
        Hl, nl, tp, Nt = H // P, lay.n_loc, lay.t_pad, sp.n_total
        head0 = sp.rank * Hl
        q, k, vt = ws["q"], ws["k"], ws["vt"]
        n_own = lay.n_own
        dt = q.dtype                     # bf16 in the product; the gloo / CPU tests drive the exchange with fp32 oracle tensors
        send = torch.empty(P, 3, B, Hl, nl * 64, dtype=dt, device=dev)
        sv = send.view(P, 3, B, Hl, nl, 64)
        sv[:, 0, :, :, :n_own] = q[:, :, tp:tp + n_own].reshape(B, P, Hl, n_own, 64).transpose(0, 1)
        sv[:, 1, :, :, :n_own] = k[:, :, tp:tp + n_own].reshape(B, P, Hl, n_own, 64).transpose(0, 1)
        send[:, 2].view(P, B, Hl, 64, nl)[..., :n_own] = vt[:, :, :, tp:tp + n_own].reshape(B, P, Hl, 64, n_own).transpose(0, 1)
        if n_own != nl:       # a short last shard: the tail of its chunks is never read by the receiver, but must not be garbage on the wire
            sv[:, :2, :, :, n_own:].zero_()
            send[:, 2].view(P, B, Hl, 64, nl)[..., n_own:].zero_()
        recv = sp.all_to_all(send)                                   # [source rank, 3, B, Hl, nl * 64]
        s_pad = ops.round_up(T + Nt, 256)
        qf, kf, vf = _workspace_heads(B, Hl, s_pad, dev, dt, live=T + Nt)   # rows >= T + Nt are zero
        hsl = slice(head0, head0 + Hl)
        qf[:, :, :T], kf[:, :, :T], vf[:, :, :, :T] = q[:, hsl, :T], k[:, hsl, :T], vt[:, hsl, :, :T]
        for g in range(P):
            lo, hi = sp.shard_range(g)
            qf[:, :, T + lo:T + hi] = recv[g, 0].view(B, Hl, nl, 64)[:, :, :hi - lo]
            kf[:, :, T + lo:T + hi] = recv[g, 1].view(B, Hl, nl, 64)[:, :, :hi - lo]
            vf[:, :, :, T + lo:T + hi] = recv[g, 2].view(B, Hl, 64, nl)[:, :, :, :hi - lo]
        ol = attend_heads(qf, kf, vf, Hl, head0, Nt)                  # [B, T + Nt, Hl * 64]
        back = torch.empty(P, B, nl, Hl * 64, dtype=dt, device=dev)
        for g in range(P):
            lo, hi = sp.shard_range(g)
            back[g, :, :hi - lo] = ol[:, T + lo:T + hi]
            if hi - lo != nl:
                back[g, :, hi - lo:].zero_()
        mine = sp.all_to_all(back)                                    # [head owner, B, nl, Hl * 64]
        text = sp.all_gather(ol[:, :T].contiguous())                  # [head owner, B, T, Hl * 64]
        o = torch.zeros(B, S, d, dtype=dt, device=dev) if tp != T else torch.empty(B, S, d, dtype=dt, device=dev)   # (rows [T, tp): alignment gap)
        o[:, :T] = text.permute(1, 2, 0, 3).reshape(B, T, d)
        o[:, tp:tp + n_own] = mine.permute(1, 2, 0, 3).reshape(B, nl, d)[:, :n_own]
        return o

    def _heads_mode(self, lay, sp, H) -> bool:
        """EA_SP_MODE=heads (sequence_parallel.SequenceParallel.mode): the full-attention blocks exchange heads, not keys."""
        if lay is None or sp.size <= 1 or getattr(sp, "mode", "keys") != "heads":
            return False
        if H % sp.size:     # the user asked for the head exchange: no silent switch to the key all-gather (the SWA blocks raise too)
            raise NotImplementedError(f"EA_SP_MODE=heads needs heads ({H}) % sequence ranks ({sp.size}) == 0; use EA_SP_MODE=keys")
        return True

    def _attend(self, ws, B, H, T, N, S, v_off, d, dev, lay, sp, grid, pending=None):
        """softmax(QK^T)V over the rows staged in ws -> bf16 [B, S, d]."""
        # ---- joint attention (processor.py:287-291): queries = text rows + this rank's video rows
        if self._heads_mode(lay, sp, H):
            # one contiguous launch over all keys for H / P' heads: no second pass, no state round trip, no replicated text queries
            def full(qf, kf, vf, Hl, head0, Nt):
                return ops.attention(qf, kf, vf, T + Nt, ops.FOLDED_ATTN_SCALE)
            return self._head_parallel(ws, B, H, T, S, d, dev, lay, sp, full)
        o = torch.empty(B, S, d, dtype=torch.bfloat16, device=dev)
        if lay is not None and sp.exchanges(lay) and ws.get("groups", 1) > 1:
            # the exchange pipelined by head groups: own-slot pass of every group (its K / V^T are the group buffer's own slot), then
            # per group: wait for ITS all-gather only, resume the state over the other ranks' slots of that group -- group g + 1 is
            # still on the links while group g's remote pass runs
            G, kvb = ws["groups"], ws["kv"]
            Hg = H // G
            state = _attention_state(B, H, S, dev)
            for g in range(G):
                k_g, vt_g = sp.slot_views(kvb[g])
                for i, (lo, hi) in enumerate(lay.own_ranges):
                    ops.attention_range(ws["q"], k_g, vt_g, ops.FOLDED_ATTN_SCALE, 0, S, lo, hi, state=state, load_state=i > 0,
                                        store_state=True, head0=g * Hg)
            for g in range(G):
                other = sp.exchange_finish(pending[g], kind=f"kv_all_gather_wait_g{g}")
                buf = other if other is not None else kvb[g]
                ops.attention_segments(ws["q"], buf, sp.size, sp.rank, lay.rows, lay.remote_valid, 0, S, state=state, load_state=True,
                                       out=o, first_row=lay.t_pad, used_rows=lay.n_loc, head0=g * Hg, group_heads=Hg)
            return o
        if lay is not None and sp.exchanges(lay):
            # sequence-parallel: attend the OWN slot (text + own shard) while the in-place K / V^T all-gather is in flight,
            # then resume the online-softmax state over the other ranks' slots where the all-gather left them
            buf = ws["kv"]
            state = _attention_state(B, H, S, dev)
            for i, (lo, hi) in enumerate(lay.own_ranges):       # the own slot is a plain K / V^T operand (same rows as q)
                ops.attention_range(ws["q"], ws["k"], ws["vt"], ops.FOLDED_ATTN_SCALE, 0, S, lo, hi, state=state, load_state=i > 0,
                                    store_state=True)
            other = sp.exchange_finish(pending)
            if other is not None:                          # (EA_SP_INPLACE=0: the remote slots arrived in a second buffer)
                buf = other
            if lay.bringup_ranges is not None:
                # bring-up mode (a world of one rank with force_exchange): the "remote" keys are the second half of its own rows
                (lo, hi), = lay.bringup_ranges
                ops.attention_range(ws["q"], ws["k"], ws["vt"], ops.FOLDED_ATTN_SCALE, 0, S, lo, hi, state=state, load_state=True, out=o)
            else:
                ops.attention_segments(ws["q"], buf, sp.size, sp.rank, lay.rows, lay.remote_valid, 0, S, state=state, load_state=True,
                                       out=o, first_row=lay.t_pad, used_rows=lay.n_loc)
        elif lay is not None:
            # one sequence rank (CFG split only): the own slot is all there is
            state = _attention_state(B, H, S, dev) if len(lay.own_ranges) > 1 else None
            for i, (lo, hi) in enumerate(lay.own_ranges):
                last = i == len(lay.own_ranges) - 1
                ops.attention_range(ws["q"], ws["k"], ws["vt"], ops.FOLDED_ATTN_SCALE, 0, S, lo, hi, state=state, load_state=i > 0,
                                    store_state=not last, out=o if last else None)
        elif v_off != T:
            # single GPU with unaligned text: rows [T, v_off) are padding between the two key ranges
            state = _attention_state(B, H, S, dev)
            ops.attention_range(ws["q"], ws["k"], ws["vt"], ops.FOLDED_ATTN_SCALE, 0, S, 0, T, state=state, store_state=True)
            ops.attention_range(ws["q"], ws["k"], ws["vt"], ops.FOLDED_ATTN_SCALE, 0, S, v_off, S, state=state, load_state=True, out=o)
        else:
            ops.attention(ws["q"], ws["k"], ws["vt"], S, ops.FOLDED_ATTN_SCALE, out=o)
        return o

    def __call__(
        self,
        attn,
        hidden_states: torch.Tensor,
        encoder_hidden_states: torch.Tensor,
        attention_mask: Optional[torch.Tensor] = None,
        image_rotary_emb: Optional[torch.Tensor] = None,
        attn2=None,
        residual: Optional[torch.Tensor] = None,
        encoder_residual: Optional[torch.Tensor] = None,
        gate: Optional[torch.Tensor] = None,
        encoder_gate: Optional[torch.Tensor] = None,
        sp=None,
        num_frames: Optional[int] = None,
        height: Optional[int] = None,
        width: Optional[int] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        if attention_mask is not None:
            # the V5.1 path never passes a mask (transformer3d.py:1502 drops text_embedding_mask)
            raise NotImplementedError("EasyAnimateAttnProcessor2_0 (HIP): attention_mask is not supported")
        x = _bf16c(hidden_states)
        e = _bf16c(encoder_hidden_states)
        B, N, d = x.shape
        T = e.shape[1]
        H = attn.heads
        assert d == H * 64, "head_dim must be 64"
        tattn = attn2 if attn2 is not None else attn  # non-MMDiT blocks share the video weights (:241-242)
        dev = x.device

        # ---- row layout of the attention operands: text rows first, then video (torch.cat at :277-279)
        lay = None
        if sp is not None:
            # per-rank rows: [text | gap | own shard]; K / V^T live in the rank's slot of the exchange buffer
            lay = sp.layout(T, N)
            S, v_off = lay.q_end, lay.t_pad
            # head groups of the pipelined K / V^T exchange (sequence_parallel.SequenceParallel.groups): the projections must be able to
            # write K / V^T group-wise, which the fused QKV launch does (bf16 weights, both streams on it); otherwise one group
            G = 1
            if (sp.exchanges(lay) and lay.bringup_ranges is None and self.exchange_in_attend is False and not self._heads_mode(lay, sp, H)
                    and self.fuse_qkv and ops.qkv_fused_ok(T, d, d, 0) and ops.qkv_fused_ok(N, d, d, lay.t_pad)
                    and all(gemm_weight(m.weight).dtype == torch.bfloat16 for a_ in (attn, tattn) for m in (a_.to_q, a_.to_k, a_.to_v))):
                G = sp.head_groups(H)
            kvb = sp.kv_buffer(B, H, lay, dev, groups=G) if G > 1 else sp.kv_buffer(B, H, lay, dev)
            k_own, vt_own = sp.slot_views(kvb[0] if G > 1 else kvb)      # (G > 1: group 0's own slot; group g is kvb.stride(0) * g further on)
            ws = dict(q=_workspace_q(B, H, lay.q_pad, dev), k=k_own, vt=vt_own, kv=kvb, groups=G, gstride=kvb.stride(0) if G > 1 else 0)
        else:
            S = T + N
            s_pad, v_off = ops.round_up(S, 256), T
            ws = _workspace(B, H, s_pad, dev)
        cos = sin = None
        if image_rotary_emb is not None:
            cos, sin = rope_to_device(image_rotary_emb, dev)
        if attn.norm_q is None or attn.norm_k is None:
            raise NotImplementedError("qk_norm=None is not supported by the HIP processor")

        # ---- QKV projections + qk LayerNorm + RoPE + head-major scatter (processor.py:244-285)
        exchange = lay is not None and sp.exchanges(lay)
        pending = None

        def start_exchange_now():
            # one all-gather per head group, posted back to back (one group: the single all-gather of the whole buffer)
            G_ = ws.get("groups", 1)
            return [sp.exchange_start(ws["kv"][g]) for g in range(G_)] if G_ > 1 else sp.exchange_start(ws["kv"])

        def qkv_stream(inp, mod, n_tok, seq_off, c, s_, start_exchange=False):
            """rows [seq_off, seq_off + n_tok) of q (workspace) and of K / V^T (workspace, or the rank's exchange slot: the
            row numbering is the same).  start_exchange: the K / V^T all-gather starts as soon as K | V exist."""
            nonlocal pending
            lq, lk, lv = mod.to_q, mod.to_k, mod.to_v
            nq, nk = mod.norm_q, mod.norm_k
            # (ragged sequence-parallel shards included: the ragged last tile together with the exchange-slot geometry and the
            # K | V-first split is covered by tests/test_kernels_gpu.py::test_qkv_projection_into_an_exchange_slot)
            if self.fuse_qkv and ops.qkv_fused_ok(n_tok, d, inp.shape[2], seq_off):
                # one launch: the [B, n, 3d] QKV buffer never exists (ea_qkv_gemm_norm_rope_bf16) -- or two, K | V first
                args = (inp, gemm_weight(lq.weight), gemm_weight(lk.weight), gemm_weight(lv.weight),
                        f32(lq.bias), f32(lk.bias), f32(lv.bias), ws["q"], ws["k"], ws["vt"],
                        f32(nq.weight), f32(nq.bias), f32(nk.weight), f32(nk.bias), c, s_, seq_off, nq.eps)
                gs = ws.get("gstride", 0)
                if start_exchange and self.kv_first:
                    ops.qkv_gemm_norm_rope(*args, q_scale=ops.FOLDED_Q_SCALE, parts=ops.QKV_KV, kv_group_stride=gs)
                    pending = start_exchange_now()
                    ops.qkv_gemm_norm_rope(*args, q_scale=ops.FOLDED_Q_SCALE, parts=ops.QKV_Q, kv_group_stride=gs)
                    return
                ops.qkv_gemm_norm_rope(*args, q_scale=ops.FOLDED_Q_SCALE, kv_group_stride=gs)
            else:
                # three GEMMs into one [B, n, 3d] buffer, then one normalise / rotate / scatter pass
                qkv = torch.empty(B, n_tok, 3 * d, dtype=torch.bfloat16, device=dev)
                for i, lin in enumerate((lq, lk, lv)):
                    ops.gemm(inp, gemm_weight(lin.weight), f32(lin.bias), ops.EPI_BIAS, out=qkv[:, :, i * d:(i + 1) * d])
                ops.qknorm_rope(qkv, ws["q"], ws["k"], ws["vt"], f32(nq.weight), f32(nq.bias), f32(nk.weight), f32(nk.bias),
                                c, s_, seq_off, nq.eps, q_scale=ops.FOLDED_Q_SCALE)
            if start_exchange:
                pending = start_exchange_now()

        qkv_stream(e, tattn, T, 0, None, None)      # text rows: no RoPE
        qkv_stream(x, attn, N, v_off, cos, sin,
                   start_exchange=exchange and self.exchange_in_attend is False and not self._heads_mode(lay, sp, H))

        o = self._attend(ws, B, H, T, N, S, v_off, d, dev, lay, sp, (num_frames, height, width), pending)
        o_t, o_v = o[:, :T], o[:, v_off:]

        # ---- output projections (:293-311), optionally with the gated residual fused (attention.py:1140-1141)
        lo_v, lo_t = attn.to_out[0], tattn.to_out[0]
        if residual is not None:
            g_v = gate.reshape(B, d)
            g_t = encoder_gate.reshape(B, d)
            h_out = ops.gemm(o_v, gemm_weight(lo_v.weight), f32(lo_v.bias), ops.EPI_BIAS_GATE_RES,
                             res=_bf16c(residual), gate=g_v)
            e_out = ops.gemm(o_t, gemm_weight(lo_t.weight), f32(lo_t.bias), ops.EPI_BIAS_GATE_RES,
                             res=_bf16c(encoder_residual), gate=g_t)
        else:
            h_out = ops.gemm(o_v, gemm_weight(lo_v.weight), f32(lo_v.bias), ops.EPI_BIAS)
            e_out = ops.gemm(o_t, gemm_weight(lo_t.weight), f32(lo_t.bias), ops.EPI_BIAS)
        return h_out, e_out


class EasyAnimateSWAttnProcessor2_0(EasyAnimateAttnProcessor2_0):
    """Sliding-window variant (reference: processor.py:320-459; selected per block by `swa_layers`, attention.py:1065).
    Same projections / qk-norm / RoPE as the full processor; the attention itself is the sum of

      * a *cross* pass: every query (text + video) over the text keys and every `interval`-th video key
        (interval = max(N // (cross_attention_size - T), 1), :393-397), and
      * a *window* pass over the video tokens only: the heads are split into six groups, group g sees the video tokens in
        its own scan order ((f h w), (f w h), (h f w), (h w f), (w f h), (w h f), :400-417) and each query attends the keys
        within +-(height*width) positions of that order (flash_attn_func window_size, :420); the result is brought back to
        (f h w) order (:422-434);

    text rows leave as cross + cross, video rows as window + cross (:435 adds `cross` to a tensor whose text rows already
    are `cross`).  Each pass rounds to bf16 like the two flash_attn_func calls do.  The cross pass is
    ea_attention_fwd_segments_bf16 over one small gathered key segment; the window pass is ea_attention_window_mapped_fwd_bf16:
    q / k are addressed through the per-head scan-order map where the projection wrote them, V^T is re-ordered once
    (ea_permute_cols_bf16), and the kernel's store returns to token order and adds the cross pass -- no index copies."""

    _ORDERS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))

    def __init__(self, cross_attention_size: int = 1024):
        super().__init__()
        self.cross_attention_size = cross_attention_size
        self._maps = None

    exchange_in_attend = True   # under sequence parallelism this processor exchanges HEADS (all-to-all), not keys

    def _swa(self, q, k, vt, B, Hl, head0, H_total, T, N, dev, grid):
        """The two passes on explicit operands: q / k [B, Hl, S_pad, 64], vt [B, Hl, 64, S_pad] hold the heads
        [head0, head0 + Hl) of H_total, rows = [text 0..T | all N video tokens in (f h w) order] -> bf16 [B, T + N, Hl * 64]."""
        F_, Hh, Ww = grid
        if F_ is None or F_ * Hh * Ww != N:
            raise ValueError("EasyAnimateSWAttnProcessor2_0 needs num_frames / height / width of the token grid")
        S = T + N
        # ---- cross pass: the text keys + every interval-th video key, gathered into ONE small segment ([2, B, Hl, rows * 64]:
        # K rows, then V^T columns) that ea_attention_fwd_segments_bf16 reads with its own geometry -- no full-size zero-filled
        # copies of k / v^T (1.3 GB of fills and 0.2 GB of gathers per block at config-3 size in the first version)
        interval = max(N // (self.cross_attention_size - T), 1)
        idx = torch.cat([torch.arange(T, device=dev), T + torch.arange(0, N, interval, device=dev)])
        nc = idx.numel()
        rows_c = ops.round_up(nc, 64)
        seg = torch.zeros(2, B, Hl, rows_c * 64, dtype=torch.bfloat16, device=dev)
        seg[0].view(B, Hl, rows_c, 64)[:, :, :nc] = k[:, :, idx]
        seg[1].view(B, Hl, 64, rows_c)[:, :, :, :nc] = vt[:, :, :, idx]
        cross = torch.empty(B, S, Hl * 64, dtype=torch.bfloat16, device=dev)
        ops.attention_segments(q, seg, 1, -1, rows_c, nc, 0, S, out=cross)
        # ---- window pass over the video tokens in the six scan orders.  q / k stay in token order and are ADDRESSED through the
        # per-head map inside the kernel; V^T (a key is a column of it) is re-ordered once by tiled transposes; the kernel's store
        # brings the result back to token order and adds the cross pass (round 4: no index copies -- the first version gathered
        # q / k / v^T into scan order and scattered the result back with torch indexing: 7 of 24 ms per block at config-3 size)
        hmap, inv, hmap32, order = self._head_maps(F_, Hh, Ww, H_total, head0, Hl, dev)   # scan position -> token, token -> scan position
        o = torch.empty(B, S, Hl * 64, dtype=torch.bfloat16, device=dev)
        if self.index_copies:
            n_pad = ops.round_up(N, 256)
            hh = torch.arange(Hl, device=dev)[:, None]
            qp = torch.empty(B, Hl, n_pad, 64, dtype=torch.bfloat16, device=dev)
            kp = torch.empty_like(qp)
            vtp = torch.empty(B, Hl, 64, n_pad, dtype=torch.bfloat16, device=dev)
            qp[:, :, :N] = q[:, hh, T + hmap]
            kp[:, :, :N] = k[:, hh, T + hmap]
            vtp[:, :, :, :N] = vt[:, :, :, T:T + N].gather(3, hmap[None, :, None, :].expand(B, Hl, 64, N))
            if n_pad != N:      # rows behind the sequence: masked as keys, but V^T must be finite there
                qp[:, :, N:].zero_(); kp[:, :, N:].zero_(); vtp[:, :, :, N:].zero_()
            win = ops.attention_window(qp, kp, vtp, N, Hh * Ww, ops.FOLDED_ATTN_SCALE).view(B, N, Hl, 64)
            back = win.transpose(1, 2)[:, hh, inv].transpose(1, 2)        # [B, N, Hl, 64] in (f h w) order again
            o[:, T:] = ops.bf16_add_(back.reshape(B, N, Hl * 64).contiguous(), cross[:, T:].contiguous())
        else:
            vtp = _workspace_vt_perm(B, Hl, ops.round_up(N, 64), dev, live=N)    # columns >= N are zero
            ops.permute_cols(vt, vtp, order, (F_, Hh, Ww), T)
            ops.attention_window_mapped(q, k, vtp, cross, o, N, T, hmap32, Hh * Ww, ops.FOLDED_ATTN_SCALE)
        # ---- text rows: cross + cross; video rows: window + cross (added in the kernel's store)
        o[:, :T] = ops.bf16_add_(cross[:, :T].contiguous(), cross[:, :T].contiguous())
        return o

    index_copies = os.environ.get("EA_SWA_INDEX_COPIES", "0") == "1"   # the first version's torch index copies (cross-check, tests)

    def _head_maps(self, F_, Hh, Ww, H_total, head0, Hl, dev):
        """Per local head: scan position -> token (long and int32), token -> scan position, and the index of its scan order."""
        key = (F_, Hh, Ww, H_total, head0, Hl, str(dev))
        if self._maps is None or self._maps[0] != key:
            N = F_ * Hh * Ww
            hmap = torch.empty(Hl, N, dtype=torch.long, device=dev)
            order = torch.zeros(Hl, dtype=torch.int32, device=dev)
            base = torch.arange(N, device=dev).view(F_, Hh, Ww)
            groups = torch.tensor_split(torch.arange(H_total, device=dev), 6)          # head groups are defined over ALL heads (:400-417)
            for oi, (hs, o) in enumerate(zip(groups, self._ORDERS)):
                loc = hs[(hs >= head0) & (hs < head0 + Hl)] - head0
                if loc.numel():
                    hmap[loc] = base.permute(*o).reshape(-1)
                    order[loc] = oi
            inv = torch.empty_like(hmap)
            inv.scatter_(1, hmap, torch.arange(N, device=dev)[None].expand(Hl, N))
            self._maps = (key, hmap, inv, hmap.to(torch.int32).contiguous(), order)
        return self._maps[1:]

    def _attend(self, ws, B, H, T, N, S, v_off, d, dev, lay, sp, grid, pending=None):
        if sp is None:
            return self._swa(ws["q"], ws["k"], ws["vt"], B, H, 0, H, T, N, dev, grid)
        # ---- sequence parallelism: the six scan orders scatter a rank's contiguous (f h w) shard over the whole sequence, so
        # the window pass is not a halo exchange.  The block switches to HEAD parallelism instead (_head_parallel): the
        # single-GPU passes run on all tokens of H / P' heads (head groups keep their global scan order).
        return self._head_parallel(ws, B, H, T, S, d, dev, lay, sp,
                                   lambda qf, kf, vf, Hl, head0, Nt: self._swa(qf, kf, vf, B, Hl, head0, H, T, Nt, dev, grid))
