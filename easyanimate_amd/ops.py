"""torch.Tensor-level wrappers over the C ABI (include/ea_mi355x.h).

PyTorch is used here only for device memory and streams; every arithmetic op on the hot path is a
HIP kernel in libea_mi355x.so.  There is no CPU / eager fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

EPI_BIAS = 0
EPI_BIAS_GELU_TANH = 1
EPI_BIAS_GATE_RES = 2
EPI_F32_OUT = 3

_BF16 = torch.bfloat16
_F32 = torch.float32
_FP8 = torch.float8_e4m3fn   # fp8 weight storage (utils/fp8_optimization.py:17-22 of the reference)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("easyanimate_amd.ops: tensors must live on the GPU -- the hot path has no CPU fallback")


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


class KernelTimer:
    """HIP-event timing of selected kernels on the stream they are launched on (torch's current stream).
    Used by bench.py for the roofline of the dominant kernel: `with ops.KernelTimer("attention") as kt: ...`,
    then kt.mean_ms() after a synchronize."""
    active = None

    def __init__(self, name: str):
        self.name = name
        self.events = []
        self.labels = []   # kernel variant that served each timed call (ea_last_dispatch)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *a):
        KernelTimer.active = None

    def wrap(self, name, fn):
        if name != self.name:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.events.append((s, e))
        self.labels.append(_lib.last_dispatch())
        return r

    def durations_ms(self):
        return [s.elapsed_time(e) for s, e in self.events]

    def mean_ms(self):
        d = self.durations_ms()
        return sum(d) / max(1, len(d))


def _timed(name, fn):
    kt = KernelTimer.active
    return fn() if kt is None else kt.wrap(name, fn)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ---------------------------------------------------------------------------------------------------
def layernorm_modulate(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                       scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], eps: float,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 [B,R,D] (last two dims contiguous); scale/shift fp32 [B,D] views with unit inner stride."""
    _dev(x, gamma, beta, scale, shift, out)
    _chk(x, _BF16, "x")
    assert x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == x.shape[2]
    B, R, D = x.shape
    if out is None:
        out = torch.empty((B, R, D), dtype=_BF16, device=x.device)
    assert out.stride(2) == 1 and out.stride(1) == D
    mod_stride = 0
    if scale is not None:
        _chk(scale, _F32, "scale"); _chk(shift, _F32, "shift")
        assert scale.shape == (B, D) and shift.shape == (B, D)
        assert scale.stride(1) == 1 and shift.stride(1) == 1 and scale.stride(0) == shift.stride(0)
        mod_stride = scale.stride(0)
    if gamma is not None:
        _chk(gamma, _F32, "gamma"); _chk(beta, _F32, "beta")
        assert gamma.is_contiguous() and beta.is_contiguous()
    _lib.call("ea_layernorm_modulate_bf16", _p(x), _p(out), _p(gamma), _p(beta), _p(scale), _p(shift),
              mod_stride, B, R, D, x.stride(0), out.stride(0), float(eps), _stream())
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    _dev(x, w)
    _chk(x, _BF16, "x"); _chk(w, _F32, "w")
    assert x.is_contiguous()
    D = x.shape[-1]
    out = torch.empty_like(x)
    _lib.call("ea_rmsnorm_bf16", _p(x), _p(out), _p(w), x.numel() // D, D, float(eps), _stream())
    return out


def linear_small_m(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], act_in: int = 0,
                   act_out: int = 0) -> torch.Tensor:
    """x fp32 [m,k] (m<=8), W bf16 [n,k], bias fp32 [n] -> fp32 [m,n]."""
    _dev(x, W, bias)
    _chk(x, _F32, "x"); _chk(W, _BF16, "W")
    assert x.is_contiguous() and W.is_contiguous()
    m, k = x.shape
    n = W.shape[0]
    assert W.shape[1] == k
    y = torch.empty((m, n), dtype=_F32, device=x.device)
    _lib.call("ea_linear_small_m", _p(x), _p(W), _p(bias), _p(y), m, n, k, act_in, act_out, _stream())
    return y


def timestep_sinusoid(t: torch.Tensor, dim: int, round_bf16: bool = True) -> torch.Tensor:
    _dev(t)
    _chk(t, _F32, "t")
    out = torch.empty((t.numel(), dim), dtype=_F32, device=t.device)
    _lib.call("ea_timestep_sinusoid", _p(t), _p(out), t.numel(), dim, int(round_bf16), _stream())
    return out


def gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int = EPI_BIAS,
         out: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
         gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = epi(A[b] @ W.T + bias).  A bf16 [B,M,K] or [M,K] (row stride free, inner stride 1),
    W bf16 [N,K] contiguous -- or torch.float8_e4m3fn [N,K] (fp8 weight storage: ea_gemm_bf16_w8 widens the weight inside
    the kernel, bit-identical to the GEMM on W.to(bf16)) --, bias fp32 [N]; res bf16 like out (may alias out); gate fp32 [B,N]."""
    _dev(A, W, bias, out, res, gate)
    w8 = W.dtype == _FP8
    _chk(A, _BF16, "A")
    if not w8:
        _chk(W, _BF16, "W")
    assert not (w8 and epilogue == EPI_F32_OUT)
    squeeze = A.dim() == 2
    if squeeze:
        A = A.unsqueeze(0)
        if out is not None:
            out = out.unsqueeze(0)
        if res is not None:
            res = res.unsqueeze(0)
        if gate is not None and gate.dim() == 1:
            gate = gate.unsqueeze(0)
    B, M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.is_contiguous() and A.stride(2) == 1
    if out is None:
        out = torch.empty((B, M, N), dtype=_F32 if epilogue == EPI_F32_OUT else _BF16, device=A.device)
    assert out.shape == (B, M, N) and out.stride(2) == 1
    assert out.dtype == (_F32 if epilogue == EPI_F32_OUT else _BF16)
    if bias is not None:
        _chk(bias, _F32, "bias")
    ldres = rbs = gbs = 0
    if epilogue == EPI_BIAS_GATE_RES:
        assert res is not None and gate is not None
        _chk(res, _BF16, "res"); _chk(gate, _F32, "gate")
        assert res.shape == (B, M, N) and res.stride(2) == 1 and gate.shape == (B, N) and gate.stride(1) == 1
        ldres, rbs, gbs = res.stride(1), res.stride(0), gate.stride(0)
    _timed("gemm", lambda: _lib.call("ea_gemm_bf16_w8" if w8 else "ea_gemm_bf16", _p(A), _p(W), _p(bias), _p(out), _p(res), _p(gate), B, M, N, K,
                                     A.stride(1), A.stride(0), out.stride(1), out.stride(0), ldres, rbs, gbs, epilogue,
                                     _stream()))
    return out.squeeze(0) if squeeze else out


LAYOUT_A, LAYOUT_W, LAYOUT_C = 1, 2, 4     # ea_gemm_bf16_kblocked: which operands are K-blocked ([K / 64][rows][64])


def kblocked_ok(B: int, M: int, N: int, K: int) -> bool:
    """Shapes the K-blocked GEMM serves: the 256 x 256 kernel's (enough tiles, N % 256 == 0) with K % 64 == 0 -- and only while
    the library's tuning switches leave that kernel selectable (ea_gemm_bf16_kblocked is the 16x16x32 256 x 256 kernel: with
    "gemm_mfma" = 32 or "gemm_tile" = 128 the callers fall back to ops.gemm, so results never depend on the switches)."""
    if not (N % 256 == 0 and K % 64 == 0 and ((M + 255) // 256) * (N // 256) * B >= 512):
        return False
    return _lib.get_option("gemm_mfma") == 16 and _lib.get_option("gemm_tile") in (0, 256)


def to_kblocked(w: torch.Tensor) -> torch.Tensor:
    """[rows, K] -> [K / 64, rows, 64] (contiguous): the K-blocked form of a row-major operand (weights: once, cached)."""
    rows, K = w.shape
    return w.view(rows, K // 64, 64).permute(1, 0, 2).contiguous()


def gemm_kblocked(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int, layout: int,
                  out: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ea_gemm_bf16 with K-blocked operands (layout = LAYOUT_A | LAYOUT_W | LAYOUT_C): A bf16 [B, M, K] (row-major, contiguous) or
    [B, K/64, M, 64]; W bf16 [N, K] or [K/64, N, 64]; out bf16 [B, M, N] or [B, N/64, M, 64]; res / gate as gemm()."""
    _dev(A, W, bias, out, res, gate)
    _chk(A, _BF16, "A"); _chk(W, _BF16, "W")
    if layout & LAYOUT_A:
        B, kb, M, c = A.shape
        K = kb * 64
        assert c == 64 and A[0].is_contiguous()
    else:
        B, M, K = A.shape
        assert A[0].is_contiguous()
    if layout & LAYOUT_W:
        kbw, N, c = W.shape
        assert c == 64 and kbw * 64 == K and W.is_contiguous()
    else:
        N = W.shape[0]
        assert W.shape == (N, K) and W.is_contiguous()
    oshape = (B, N // 64, M, 64) if layout & LAYOUT_C else (B, M, N)
    if out is None:
        out = torch.empty(oshape, dtype=_BF16, device=A.device)
    assert out.shape == oshape and out[0].is_contiguous() and kblocked_ok(B, M, N, K)
    if bias is not None:
        _chk(bias, _F32, "bias")
    ldres = rbs = gbs = 0
    if epilogue == EPI_BIAS_GATE_RES:
        assert res is not None and gate is not None and not (layout & LAYOUT_C)
        _chk(res, _BF16, "res"); _chk(gate, _F32, "gate")
        assert res.shape == (B, M, N) and res.stride(2) == 1 and gate.shape == (B, N) and gate.stride(1) == 1
        ldres, rbs, gbs = res.stride(1), res.stride(0), gate.stride(0)
    _timed("gemm", lambda: _lib.call("ea_gemm_bf16_kblocked", _p(A), _p(W), _p(bias), _p(out), _p(res), _p(gate), B, M, N, K, A.stride(0),
                                     out.stride(0), ldres, rbs, gbs, int(epilogue), int(layout), _stream()))
    return out


# Softmax scale of head_dim 64 folded into Q (exp2 domain): q_scale = 64^-1/2 * log2(e) at ea_qknorm_rope_bf16, and
# scale = ln(2) at ea_attention_fwd_* (scale * log2(e) == 1 selects the kernel that exponentiates raw scores).
FOLDED_Q_SCALE = 0.125 * 1.4426950408889634
FOLDED_ATTN_SCALE = 0.6931471805599453


def qknorm_rope(qkv: torch.Tensor, q_out: torch.Tensor, k_out: torch.Tensor, vt_out: torch.Tensor,
                nq_w, nq_b, nk_w, nk_b, cos: Optional[torch.Tensor], sin: Optional[torch.Tensor],
                seq_off: int, eps: float, q_scale: float = 1.0, kv_off: Optional[int] = None) -> None:
    """qkv bf16 [B,n_tok,3*H*64]; q_out bf16 [B,H,S_pad,64] (rows from seq_off); k_out bf16 [B,H,R,64], vt_out bf16
    [B,H,64,R] (rows / columns from kv_off; R and kv_off default to q_out's geometry).
    q_scale multiplies q ahead of its bf16 rounding (FOLDED_Q_SCALE folds the softmax scale of head_dim 64)."""
    _dev(qkv, q_out, k_out, vt_out, nq_w, nq_b, nk_w, nk_b, cos, sin)
    _chk(qkv, _BF16, "qkv")
    B, n_tok, three_inner = qkv.shape
    _, H, s_pad, dh = q_out.shape
    assert dh == 64 and three_inner == 3 * H * 64 and qkv.stride(2) == 1 and qkv.stride(1) == three_inner
    assert q_out.is_contiguous() and k_out.is_contiguous() and vt_out.is_contiguous()
    kv_rows = k_out.shape[2]
    kv_off = seq_off if kv_off is None else kv_off
    assert k_out.shape == (B, H, kv_rows, 64) and vt_out.shape == (B, H, 64, kv_rows) and kv_off + n_tok <= kv_rows
    if cos is not None:
        _chk(cos, _F32, "cos"); _chk(sin, _F32, "sin")
        assert cos.shape == (n_tok, 64) and cos.is_contiguous() and sin.is_contiguous()
    _lib.call("ea_qknorm_rope_bf16", _p(qkv), qkv.stride(0), _p(q_out), _p(k_out), _p(vt_out), _p(nq_w), _p(nq_b),
              _p(nk_w), _p(nk_b), _p(cos), _p(sin), B, H, n_tok, seq_off, s_pad, kv_off, kv_rows, float(eps), float(q_scale),
              _stream())


def qkv_fused_ok(n_tok: int, inner: int, k: int, seq_off: int, kv_off: Optional[int] = None) -> bool:
    """Shapes ea_qkv_gemm_norm_rope_bf16 is used for: whole 256-row tiles, or a long stream with a ragged last tile (the video
    stream of every benchmark configuration, of the reference's published shapes -- 384x672, 576x1008, 768x1344: none a
    multiple of 256 tokens -- and of sequence-parallel shards); short ragged streams (unaligned text) stay on the 128-row GEMMs."""
    return ((n_tok > 0 and n_tok % 256 == 0) or n_tok >= 512) and inner % 256 == 0 and k % 64 == 0 and seq_off % 8 == 0 \
        and (kv_off or 0) % 8 == 0


QKV_ALL, QKV_Q, QKV_KV = 7, 1, 6   # `parts` of qkv_gemm_norm_rope: which thirds of the q | k | v axis one launch computes


def qkv_gemm_norm_rope(x: torch.Tensor, wq, wk, wv, bq, bk, bv, q_out: torch.Tensor, k_out: torch.Tensor,
                       vt_out: torch.Tensor, nq_w, nq_b, nk_w, nk_b, cos: Optional[torch.Tensor],
                       sin: Optional[torch.Tensor], seq_off: int, eps: float, q_scale: float = 1.0,
                       kv_off: Optional[int] = None, parts: int = QKV_ALL, kv_group_stride: int = 0) -> None:
    """x bf16 [B, n_tok, K] -> rows [seq_off, seq_off+n_tok) of q_out [B,H,S_pad,64], rows [kv_off, kv_off+n_tok) of k_out
    [B,H,R,64] and columns of vt_out [B,H,64,R] (R, kv_off default to q_out's geometry): the three projections +
    qk-LayerNorm + RoPE + scatter in one launch (parts = QKV_KV / QKV_Q: two launches, K | V first).
    kv_group_stride > 0 (ea_qkv_gemm_norm_rope_grouped_bf16): k_out / vt_out are the buffers of head GROUP 0, [B, Hg, R, 64] /
    [B, Hg, 64, R] with Hg dividing H; group g lives kv_group_stride elements further on (the caller owns that memory)."""
    _dev(x, wq, wk, wv, bq, bk, bv, q_out, k_out, vt_out, nq_w, nq_b, nk_w, nk_b, cos, sin)
    _chk(x, _BF16, "x")
    B, n_tok, K = x.shape
    _, H, s_pad, dh = q_out.shape
    assert dh == 64 and x.stride(2) == 1 and q_out.is_contiguous() and k_out.is_contiguous() and vt_out.is_contiguous()
    kv_rows = k_out.shape[2]
    kv_off = seq_off if kv_off is None else kv_off
    Hg = k_out.shape[1] if kv_group_stride else H
    assert H % Hg == 0 and k_out.shape == (B, Hg, kv_rows, 64) and vt_out.shape == (B, Hg, 64, kv_rows) and kv_off + n_tok <= kv_rows
    w8 = wq.dtype == _FP8          # fp8 weight storage: all three or none
    for w in (wq, wk, wv):
        _chk(w, _FP8 if w8 else _BF16, "W")
        assert w.shape == (H * 64, K) and w.is_contiguous()
    if cos is not None:
        _chk(cos, _F32, "cos"); _chk(sin, _F32, "sin")
        assert cos.shape == (n_tok, 64) and cos.is_contiguous() and sin.is_contiguous()
    if kv_group_stride:
        assert not w8, "grouped K / V^T destination: bf16 weights only"
        _timed("gemm", lambda: _lib.call("ea_qkv_gemm_norm_rope_grouped_bf16", _p(x), _p(wq), _p(wk), _p(wv), _p(bq), _p(bk), _p(bv),
                                         _p(q_out), _p(k_out), _p(vt_out), _p(nq_w), _p(nq_b), _p(nk_w), _p(nk_b), _p(cos),
                                         _p(sin), B, n_tok, H, K, x.stride(1), x.stride(0), seq_off, s_pad, kv_off, kv_rows,
                                         int(parts), Hg, int(kv_group_stride), float(eps), float(q_scale), _stream()))
        return
    _timed("gemm", lambda: _lib.call("ea_qkv_gemm_norm_rope_bf16_w8" if w8 else "ea_qkv_gemm_norm_rope_bf16", _p(x), _p(wq), _p(wk), _p(wv), _p(bq), _p(bk), _p(bv),
                                     _p(q_out), _p(k_out), _p(vt_out), _p(nq_w), _p(nq_b), _p(nk_w), _p(nk_b), _p(cos),
                                     _p(sin), B, n_tok, H, K, x.stride(1), x.stride(0), seq_off, s_pad, kv_off, kv_rows,
                                     int(parts), float(eps), float(q_scale), _stream()))


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, seq: int, scale: float,
              out: Optional[torch.Tensor] = None, q_begin: int = 0, q_end: Optional[int] = None) -> torch.Tensor:
    """q,k bf16 [B,H,S_pad,64], vt bf16 [B,H,64,S_pad] -> out bf16 [B,seq,H*64]."""
    _dev(q, k, vt, out)
    B, H, s_pad, dh = q.shape
    assert dh == 64 and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    if out is None:
        out = torch.empty((B, seq, H * 64), dtype=_BF16, device=q.device)
    assert out.stride(2) == 1 and out.stride(1) == H * 64
    if q_end is None:
        q_end = seq
    _timed("attention", lambda: _lib.call("ea_attention_fwd_bf16", _p(q), _p(k), _p(vt), _p(out), out.stride(0), B, H,
                                          seq, s_pad, q_begin, q_end, float(scale), _stream()))
    return out


def attention_segments(q: torch.Tensor, gathered: torch.Tensor, n_seg: int, skip_seg: int, seg_rows: int, kv_valid: int,
                       q_begin: int, q_end: int, state: Optional[torch.Tensor] = None, load_state: bool = False,
                       store_state: bool = False, out: Optional[torch.Tensor] = None, first_row: int = 0,
                       used_rows: Optional[int] = None, head0: int = 0, group_heads: int = 0) -> Optional[torch.Tensor]:
    """Attention of query rows [q_begin, q_end) over the key segments of an exchange buffer `gathered`
    [n_seg, 2, B, H, seg_rows*64] (per rank: K rows [B,H,seg_rows,64], then V^T [B,H,64,seg_rows]), skipping segment
    skip_seg; of every segment the rows [first_row, first_row + used_rows) are keys; softmax scale folded into Q.
    group_heads > 0: a HEAD WINDOW -- `gathered` holds group_heads heads ([n_seg, 2, B, group_heads, seg_rows*64]) and the launch
    serves heads [head0, head0 + group_heads) of q / out / state (which keep all H heads)."""
    _dev(q, gathered, out, state)
    B, H, q_pad, dh = q.shape
    Hl = group_heads or H
    assert dh == 64 and q.is_contiguous() and gathered.is_contiguous() and gathered.dtype == _BF16
    assert gathered.numel() == n_seg * 2 * B * Hl * seg_rows * 64 and head0 + Hl <= H
    used_rows = seg_rows - first_row if used_rows is None else used_rows
    flags = (1 if load_state else 0) | (2 if store_state else 0)
    if flags:
        assert state is not None and state.dtype == _F32 and state.numel() * 4 >= _lib.load().ea_attention_state_bytes(B, H, q_begin, q_end)
    if not store_state:
        assert out is not None and out.stride(2) == 1 and out.stride(1) == H * 64 and out.shape[1] >= q_end
    half = B * Hl * seg_rows * 64
    base = gathered.data_ptr()
    if group_heads:
        _timed("attention", lambda: _lib.call("ea_attention_fwd_segments_heads_bf16", _p(q), ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * half),
                                              _p(out), out.stride(0) if out is not None else 0, B, Hl, q_pad, q_begin, q_end, seg_rows,
                                              n_seg, skip_seg, 2 * half, first_row, used_rows, kv_valid, FOLDED_ATTN_SCALE, _p(state),
                                              flags, head0, H, 0, _stream()))
        return out
    _timed("attention", lambda: _lib.call("ea_attention_fwd_segments_bf16", _p(q), ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * half),
                                          _p(out), out.stride(0) if out is not None else 0, B, H, q_pad, q_begin, q_end, seg_rows,
                                          n_seg, skip_seg, 2 * half, first_row, used_rows, kv_valid, FOLDED_ATTN_SCALE, _p(state),
                                          flags, _stream()))
    return out


def attention_window(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, seq: int, window: int, scale: float,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Band attention |i - j| <= window over rows [0, seq): q,k bf16 [B,H,S_pad,64], vt [B,H,64,S_pad] -> [B,seq,H*64]."""
    _dev(q, k, vt, out)
    B, H, s_pad, dh = q.shape
    assert dh == 64 and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous() and k.shape == q.shape
    if out is None:
        out = torch.empty((B, seq, H * 64), dtype=_BF16, device=q.device)
    assert out.stride(2) == 1 and out.stride(1) == H * 64
    _timed("attention", lambda: _lib.call("ea_attention_window_fwd_bf16", _p(q), _p(k), _p(vt), _p(out), out.stride(0), B, H,
                                          seq, s_pad, int(window), float(scale), _stream()))
    return out


def attention_window_mapped(q: torch.Tensor, k: torch.Tensor, vt_perm: torch.Tensor, cross: torch.Tensor, out: torch.Tensor,
                            seq: int, row_off: int, head_map: torch.Tensor, window: int, scale: float) -> torch.Tensor:
    """The SWA window pass without index copies: q, k bf16 [B,H,S_pad,64] in token order (video token n at row row_off + n),
    visited in the scan order head_map int32 [H, seq] (scan position -> token); vt_perm bf16 [B,H,64,vt_pad] already in scan
    order (permute_cols); out[b, row_off + token, h*64..] = window result + cross[same row] (out, cross bf16 [B, >= row_off+seq, H*64])."""
    _dev(q, k, vt_perm, cross, out, head_map)
    B, H, s_pad, dh = q.shape
    vt_pad = vt_perm.shape[3]
    assert dh == 64 and q.is_contiguous() and k.is_contiguous() and vt_perm.is_contiguous() and k.shape == q.shape
    assert vt_perm.shape == (B, H, 64, vt_pad) and head_map.dtype == torch.int32 and head_map.shape == (H, seq) and head_map.is_contiguous()
    assert out.shape == cross.shape and out.shape[0] == B and out.shape[1] >= row_off + seq and out.shape[2] == H * 64
    assert out.stride() == cross.stride() and out.stride(2) == 1 and out.stride(1) == H * 64
    _timed("attention", lambda: _lib.call("ea_attention_window_mapped_fwd_bf16", _p(q), _p(k), _p(vt_perm), _p(cross), _p(out), out.stride(0),
                                          B, H, seq, s_pad, vt_pad, int(row_off), _p(head_map), int(window), float(scale), _stream()))
    return out


def permute_cols(src: torch.Tensor, dst: torch.Tensor, head_order: torch.Tensor, grid, col_off: int) -> torch.Tensor:
    """dst[b,h,c,p] = src[b,h,c,col_off + token(p)]: V^T bf16 [B,H,64,src_pad] -> [B,H,64,dst_pad] with the token axis of head h
    re-ordered by scan order head_order[h] (int32 [H], 0..5 = (f h w), (f w h), (h f w), (h w f), (w f h), (w h f)) of the
    (frames, height, width) grid."""
    _dev(src, dst, head_order)
    _chk(src, _BF16, "src"); _chk(dst, _BF16, "dst")
    B, H, c, src_pad = src.shape
    F_, Hh, Ww = grid
    assert c == 64 and dst.shape[:3] == (B, H, 64) and src.is_contiguous() and dst.is_contiguous()
    assert head_order.dtype == torch.int32 and head_order.shape == (H,) and head_order.is_contiguous()
    _lib.call("ea_permute_cols_bf16", _p(src), _p(dst), _p(head_order), B, H, int(F_), int(Hh), int(Ww), src_pad, dst.shape[3], int(col_off),
              _stream())
    return dst


def attention_d512(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, n_keys: int, scale: float,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Single-head, head_dim 512 flash attention per frame (VAE mid block): q bf16 [T, n_q, 512], k bf16 [T, n_kpad, 512],
    vt bf16 [T, 512, n_kpad] (V^T) -> bf16 [T, n_q, 512]; keys [0, n_keys) of every frame are attended."""
    _dev(q, k, vt, out)
    _chk(q, _BF16, "q"); _chk(k, _BF16, "k"); _chk(vt, _BF16, "vt")
    T, n_q, D = q.shape
    n_kpad = k.shape[1]
    assert D == 512 and k.shape == (T, n_kpad, 512) and vt.shape == (T, 512, n_kpad) and n_kpad % 32 == 0 and 0 < n_keys <= n_kpad
    assert q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    if out is None:
        out = torch.empty_like(q)
    assert out.shape == q.shape and out.is_contiguous()
    _timed("attention", lambda: _lib.call("ea_attention_d512_fwd_bf16", _p(q), _p(k), _p(vt), _p(out), T, n_q, n_keys, n_kpad,
                                          q.stride(0), k.stride(0), vt.stride(0), out.stride(0), float(scale), _stream()))
    return out


def attention_state(B: int, H: int, q_begin: int, q_end: int, device) -> torch.Tensor:
    """fp32 scratch for the resumable attention (ea_attention_state_bytes)."""
    n = _lib.load().ea_attention_state_bytes(B, H, q_begin, q_end)
    return torch.empty(n // 4, dtype=_F32, device=device)


def attention_range(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, scale: float, q_begin: int, q_end: int,
                    kv_begin: int, kv_end: int, state: Optional[torch.Tensor] = None, load_state: bool = False,
                    store_state: bool = False, out: Optional[torch.Tensor] = None, head0: int = 0) -> Optional[torch.Tensor]:
    """Attention over the key range [kv_begin, kv_end) for query rows [q_begin, q_end); `state` carries the
    online-softmax state between calls.  out: bf16 [B, >= q_end, H*64] (written unless store_state).
    k / vt with fewer heads than q ([B, Hg, s_pad, 64] / [B, Hg, 64, s_pad]): a HEAD WINDOW -- the launch serves heads
    [head0, head0 + Hg) of q / out / state (ea_attention_fwd_range_heads_bf16)."""
    _dev(q, k, vt, out, state)
    B, H, s_pad, dh = q.shape
    Hl = k.shape[1]
    assert dh == 64 and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous() and head0 + Hl <= H and vt.shape[1] == Hl
    flags = (1 if load_state else 0) | (2 if store_state else 0)
    if flags:
        assert state is not None and state.dtype == _F32 and state.is_contiguous()
        assert state.numel() * 4 >= _lib.load().ea_attention_state_bytes(B, H, q_begin, q_end)
    if not store_state:
        assert out is not None and out.stride(2) == 1 and out.stride(1) == H * 64 and out.shape[1] >= q_end
    if Hl != H:
        _timed("attention", lambda: _lib.call("ea_attention_fwd_range_heads_bf16", _p(q), _p(k), _p(vt), _p(out),
                                              out.stride(0) if out is not None else 0, B, Hl, s_pad, q_begin, q_end, kv_begin,
                                              kv_end, float(scale), _p(state), flags, head0, H, 0, _stream()))
        return out
    _timed("attention", lambda: _lib.call("ea_attention_fwd_range_bf16", _p(q), _p(k), _p(vt), _p(out),
                                          out.stride(0) if out is not None else 0, B, H, s_pad, q_begin, q_end, kv_begin,
                                          kv_end, float(scale), _p(state), flags, _stream()))
    return out


_tc_ws = {}


def teacache_rel_l1_sums(cur: torch.Tensor, prev: torch.Tensor):
    """-> (fp64 device tensor [sum |bf16(cur-prev)|, sum |prev|], element count)."""
    _dev(cur, prev)
    _chk(cur, _BF16, "cur"); _chk(prev, _BF16, "prev")
    assert cur.shape == prev.shape and cur.is_contiguous() and prev.is_contiguous()
    n = cur.numel()
    nblk = max(1, min(2048, n // (8 * 256 * 4)))
    key = (nblk, str(cur.device))
    ws = _tc_ws.get(key)
    if ws is None:
        ws = (torch.empty(2 * nblk, dtype=_F32, device=cur.device), torch.empty(2, dtype=torch.float64, device=cur.device))
        _tc_ws[key] = ws
    sums = torch.empty(2, dtype=torch.float64, device=cur.device)
    _lib.call("ea_teacache_rel_l1_bf16", _p(cur), _p(prev), n, _p(ws[0]), nblk, _p(sums), _stream())
    return sums, n


def bf16_sub(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _dev(a, b)
    _chk(a, _BF16, "a"); _chk(b, _BF16, "b")
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    out = torch.empty_like(a)
    _lib.call("ea_bf16_binary", _p(a), _p(b), _p(out), a.numel(), 0, _stream())
    return out


def bf16_add_(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    _dev(x, y)
    _chk(x, _BF16, "x"); _chk(y, _BF16, "y")
    assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous()
    _lib.call("ea_bf16_binary", _p(x), _p(y), _p(x), x.numel(), 1, _stream())
    return x


def gated_residual(x: torch.Tensor, res: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """res + gate[b, :] * x;  x/res bf16 [B, R, D] contiguous, gate fp32 [B, D] (row view of the modulation table)."""
    _dev(x, res, gate)
    _chk(x, _BF16, "x"); _chk(res, _BF16, "res"); _chk(gate, _F32, "gate")
    assert x.shape == res.shape and x.dim() == 3 and x.is_contiguous() and res.is_contiguous()
    B, R, D = x.shape
    assert gate.shape == (B, D) and gate.stride(1) == 1
    out = torch.empty_like(x)
    _lib.call("ea_gated_residual_bf16", _p(x), _p(res), _p(gate), _p(out), B, R, D, gate.stride(0), _stream())
    return out


def rope_half_scatter(src: torch.Tensor, heads: int, head_dim: int, batch: int, seq: int, cos: Optional[torch.Tensor] = None,
                      sin: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src bf16 [batch*seq, >= heads*head_dim] (a column block of a projection output; row stride free) -> bf16 [batch, heads, seq,
    head_dim] with the rotate-half rotary embedding applied (cos / sin fp32 [batch*seq, head_dim]; None: scatter only)."""
    _dev(src, cos, sin)
    _chk(src, _BF16, "src")
    assert src.dim() == 2 and src.shape[0] == batch * seq and src.stride(1) == 1 and src.shape[1] >= heads * head_dim
    if cos is not None:
        _chk(cos, _F32, "cos"); _chk(sin, _F32, "sin")
        assert cos.shape == (batch * seq, head_dim) and sin.shape == cos.shape and cos.is_contiguous() and sin.is_contiguous()
    dst = torch.empty((batch, heads, seq, head_dim), dtype=_BF16, device=src.device)
    _lib.call("ea_rope_half_scatter_bf16", _p(src), _p(dst), _p(cos), _p(sin), batch, seq, heads, head_dim, src.stride(0), _stream())
    return dst


def silu_mul(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """SiLU(gate) * up; gate / up bf16 [rows, cols] (row strides free: two column blocks of one projection output)."""
    _dev(gate, up)
    _chk(gate, _BF16, "gate"); _chk(up, _BF16, "up")
    assert gate.dim() == 2 and gate.shape == up.shape and gate.stride(1) == 1 and up.stride(1) == 1
    out = torch.empty(tuple(gate.shape), dtype=_BF16, device=gate.device)
    _lib.call("ea_silu_mul_bf16", _p(gate), _p(up), _p(out), gate.shape[0], gate.shape[1], gate.stride(0), up.stride(0), _stream())
    return out


def attention_causal_gqa(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, seq: int, scale: float, valid: Optional[torch.Tensor] = None,
                         causal: bool = True) -> torch.Tensor:
    """q bf16 [B, Hq, S, D], k bf16 [B, Hkv, S, D], vt bf16 [B, Hkv, D, S_pad] (S_pad % 32 == 0, columns >= S finite), valid int32 [B]
    (real tokens of each right-padded prompt) -> bf16 [B, S, Hq * D].  Prompt-sized sequences (ea_attention_causal_gqa_bf16)."""
    _dev(q, k, vt, valid)
    _chk(q, _BF16, "q"); _chk(k, _BF16, "k"); _chk(vt, _BF16, "vt")
    B, Hq, S, D = q.shape
    Hkv = k.shape[1]
    assert S == seq and k.shape == (B, Hkv, S, D) and vt.shape[:3] == (B, Hkv, D) and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    if valid is not None:
        assert valid.dtype == torch.int32 and valid.shape == (B,) and valid.is_contiguous()
    out = torch.empty((B, S, Hq * D), dtype=_BF16, device=q.device)
    _lib.call("ea_attention_causal_gqa_bf16", _p(q), _p(k), _p(vt), _p(out), _p(valid), B, Hq, Hkv, S, vt.shape[3], D, int(causal), float(scale),
              _stream())
    return out


def tile_blend_(a: torch.Tensor, b: torch.Tensor, extent: int, axis: int) -> torch.Tensor:
    """In place on b: the seam blend of two neighbouring VAE tiles [B,C,T,H,W] (contiguous, same dtype bf16 / fp32), axis 3 =
    blend_v (a above b), axis 4 = blend_h (a left of b); autoencoder_magvit.py:319-337."""
    _dev(a, b)
    assert a.dim() == 5 and b.dim() == 5 and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.dtype in (_BF16, _F32)
    assert a.shape[:3] == b.shape[:3] and axis in (3, 4)
    if axis == 3:
        assert a.shape[4] == b.shape[4]
        extent = min(a.shape[3], b.shape[3], extent)
        outer, inner, a_n = a.shape[0] * a.shape[1] * a.shape[2], a.shape[4], a.shape[3]
        a_os, b_os = a.shape[3] * a.shape[4], b.shape[3] * b.shape[4]
    else:
        assert a.shape[3] == b.shape[3]
        extent = min(a.shape[4], b.shape[4], extent)
        outer, inner, a_n = a.shape[0] * a.shape[1] * a.shape[2] * a.shape[3], 1, a.shape[4]
        a_os, b_os = a.shape[4], b.shape[4]
    _lib.call("ea_tile_blend", _p(a), _p(b), int(a.dtype == _BF16), outer, extent, inner, a_os, a_n, b_os, _stream())
    return b


def tile_corner_blend_(q: torch.Tensor, dec: torch.Tensor) -> torch.Tensor:
    """In place on dec [B,C,T,H,W]: mix the separately decoded lower-right tile q [B,C,T,h,w] into its last h x w pixels with the
    weights min(linspace_x, linspace_y) (autoencoder_magvit.py:426-445)."""
    _dev(q, dec)
    assert q.dim() == 5 and dec.dim() == 5 and q.is_contiguous() and dec.is_contiguous() and q.dtype == dec.dtype and q.dtype in (_BF16, _F32)
    assert q.shape[:3] == dec.shape[:3] and q.shape[3] <= dec.shape[3] and q.shape[4] <= dec.shape[4]
    _lib.call("ea_tile_corner_blend", _p(q), _p(dec), int(q.dtype == _BF16), q.shape[0] * q.shape[1] * q.shape[2], q.shape[3], q.shape[4],
              dec.shape[3], dec.shape[4], _stream())
    return dec


def patchify(latents: torch.Tensor, extra: Optional[torch.Tensor], k_pad: int) -> torch.Tensor:
    """latents [B,C,F,H,W] (+ extra [B,C2,F,H,W]) -> bf16 [B, F*(H/2)*(W/2), k_pad]."""
    _dev(latents, extra)
    assert latents.is_contiguous() and latents.dtype in (_BF16, _F32)
    B, C, F, H, W = latents.shape
    c2 = 0
    if extra is not None:
        assert extra.is_contiguous() and extra.dtype == latents.dtype and extra.shape[0] == B and extra.shape[2:] == latents.shape[2:]
        c2 = extra.shape[1]
    cols = torch.empty((B, F * (H // 2) * (W // 2), k_pad), dtype=_BF16, device=latents.device)
    _lib.call("ea_patchify", _p(latents), _p(extra), _p(cols), B, C, c2, F, H, W, k_pad,
              int(latents.dtype == _BF16), _stream())
    return cols


def unpatchify(tokens: torch.Tensor, channels: int, frames: int, h: int, w: int, out_dtype) -> torch.Tensor:
    _dev(tokens)
    _chk(tokens, _BF16, "tokens")
    assert tokens.is_contiguous() and tokens.shape[1] == frames * h * w and tokens.shape[2] == channels * 4
    B = tokens.shape[0]
    out = torch.empty((B, channels, frames, 2 * h, 2 * w), dtype=out_dtype, device=tokens.device)
    _lib.call("ea_unpatchify", _p(tokens), _p(out), B, channels, frames, h, w, int(out_dtype == _BF16), _stream())
    return out


def cfg_euler_step(v: torch.Tensor, latents: torch.Tensor, guidance: float, dsigma: float, do_cfg: bool) -> None:
    """In-place: latents <- latents + dsigma * cfg(v).  v [2 or 1, ...] same dtype as latents."""
    _dev(v, latents)
    assert v.is_contiguous() and latents.is_contiguous() and v.dtype == latents.dtype
    n = latents.numel()
    assert v.numel() == (2 * n if do_cfg else n)
    _lib.call("ea_cfg_euler_step", _p(v), _p(latents), n, float(guidance), float(dsigma), int(do_cfg),
              int(latents.dtype == _BF16), _stream())


def cfg_rescale_euler_step(v: torch.Tensor, latents: torch.Tensor, guidance: float, dsigma: float, rescale: float) -> None:
    """In-place: latents <- latents + dsigma * rescale_noise_cfg(cfg(v), v_text, rescale).  v [2, ...], one sample."""
    _dev(v, latents)
    assert v.is_contiguous() and latents.is_contiguous() and v.dtype == latents.dtype
    n = latents.numel()
    assert v.numel() == 2 * n and latents.shape[0] == 1, "guidance_rescale: one sample (a CFG pair) per call"
    nblk = max(1, min(1024, n // 4096))
    partial = torch.empty(4 * nblk, dtype=_F32, device=v.device)
    sums = torch.empty(4, dtype=torch.float64, device=v.device)
    _lib.call("ea_cfg_rescale_euler_step", _p(v), _p(latents), n, float(guidance), float(dsigma), float(rescale), _p(partial),
              nblk, _p(sums), int(latents.dtype == _BF16), _stream())


# ---------------------------------------------------------------------------------------------------
# VAE ops: channels-last activations [T, H, W, C] bf16 (one sample)
# ---------------------------------------------------------------------------------------------------
_zero_pages = {}


def _zeros_page(device) -> torch.Tensor:
    z = _zero_pages.get(str(device))
    if z is None:
        z = torch.zeros(256, dtype=_BF16, device=device)
        _zero_pages[str(device)] = z
    return z


def conv_out_shape(T, H, W, k: int, st: int, ss: int, pad: int, ups: bool = False):
    He, We = (2 * H, 2 * W) if ups else (H, W)
    To = (T + (k - 1) - k) // st + 1
    pad_hi = 0 if k == 1 else (pad if pad else 1)
    Ho = (He + pad + pad_hi - k) // ss + 1
    Wo = (We + pad + pad_hi - k) // ss + 1
    return To, Ho, Wo


def conv3d_cl(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], k: int, st: int = 1, ss: int = 1,
              pad: int = 1, ups: bool = False, tdup: bool = False, res: Optional[torch.Tensor] = None,
              want_stats: bool = True, vin: bool = False, vres: bool = False, tmerge: bool = False,
              blocked: bool = False) -> torch.Tensor:
    """x bf16 [T,H,W,Cin] (Cin % 64 == 0), w_packed bf16 [Cout, k^3*Cin] -> bf16 [T',H',W',Cout].
    Cin == 8 (RGB padded to one 16-byte chunk per voxel; k = 3): w_packed [Cout, 256] = 32 tap slots x 8 channels, zero beyond
    tap 26 / the real channels (pack_conv_weight_c8).
    If the row-slab kernel serves the call it also leaves the per-frame GroupNorm partial sums of its output on the
    returned tensor (`y.gn_partial = (partial, nblk)`): groupnorm_silu() then skips its statistics pass.
    vin / vres: x / res hold the PHYSICAL frames of a virtually duplicated clip (temporal nearest x2 never materialised:
    logical frame f = physical frame (f + 1) >> 1; ea_mi355x.h, tdup bits 2 / 4)."""
    _dev(x, w_packed, bias, res)
    _chk(x, _BF16, "x"); _chk(w_packed, _BF16, "w")
    if blocked:     # channel-blocked input [Cin/32, T, H, W, 32] (groupnorm_silu(..., blocked=True); tdup bit 16; conv3d_blocked_ok)
        assert x.is_contiguous() and w_packed.is_contiguous() and x.dim() == 5 and x.shape[4] == 32
        T, H, W, Cin = x.shape[1], x.shape[2], x.shape[3], x.shape[0] * 32
    else:
        assert x.is_contiguous() and w_packed.is_contiguous() and x.dim() == 4
        T, H, W, Cin = x.shape
    if tmerge:      # merged temporal taps (tdup bit 8): w_packed [2, Cout, 18*Cin], see vae_modules._pack_tmerge_weight
        assert vin and k == 3 and w_packed.dim() == 3 and w_packed.shape[0] == 2 and w_packed.shape[2] == 18 * Cin
        Cout = w_packed.shape[1]
    else:
        Cout = w_packed.shape[0]
        assert w_packed.shape[1] == (256 if Cin == 8 else k * k * k * Cin)
    Tl = (2 * T - 1 if T > 1 else T) if vin else T          # logical input frames
    To, Ho, Wo = conv_out_shape(Tl, H, W, k, st, ss, pad, ups)
    Ty = 2 * To - 1 if (tdup and To > 1) else To
    y = torch.empty((Ty, Ho, Wo, Cout), dtype=_BF16, device=x.device)
    if res is not None:
        assert res.is_contiguous() and res.shape == ((To + 1) // 2 if vres else To, Ho, Wo, Cout) and res.dtype == _BF16
    dup = int(tdup and To > 1) | (2 if vin else 0) | (4 if (vres and res is not None) else 0) | (8 if tmerge else 0) | (16 if blocked else 0)
    if want_stats and k == 3 and ((st, ss, pad) == (1, 1, 1) or (ss == 2 and pad == 0 and not ups)) and Wo % 256 == 0 and Cout % 128 == 0:
        cap = Ty * Ho * (Wo // 256) * 4 * (Cout // 4) * 2       # the largest layout the kernel may choose
        partial = torch.empty(cap, dtype=_F32, device=x.device)
        nblk = ctypes.c_int(0)
        _timed("conv3d", lambda: _lib.call("ea_conv3d_cl_stats_bf16", _p(x), _p(w_packed), _p(bias), _p(res), _p(y),
                                           _p(_zeros_page(x.device)), T, H, W, Cin, Cout, k, k, k, st, ss, pad, int(ups), dup,
                                           _p(partial), cap, ctypes.byref(nblk), _stream()))
        if nblk.value:
            y.gn_partial = (partial, nblk.value)
        return y
    _timed("conv3d", lambda: _lib.call("ea_conv3d_cl_bf16", _p(x), _p(w_packed), _p(bias), _p(res), _p(y),
                                       _p(_zeros_page(x.device)), T, H, W, Cin, Cout, k, k, k, st, ss, pad, int(ups),
                                       dup, _stream()))
    return y


def conv3d_blocked_ok(T: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    """Will this 3x3x3 / stride 1 / pad 1 layer (T output frames) be served by a kernel that reads a channel-blocked input?"""
    return bool(_lib.load().ea_conv3d_cl_blocked_ok(int(T), int(H), int(W), int(Cin), int(Cout)))


def conv3d_tmerge_ok(T_logical: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    """Does the kernel that will serve this 3x3x3 layer accept merged temporal taps (ea_conv3d_cl_tmerge_ok)?"""
    return bool(_lib.load().ea_conv3d_cl_tmerge_ok(int(T_logical), int(H), int(W), int(Cin), int(Cout)))


def conv3d_subpixel(x: torch.Tensor, w4: torch.Tensor, bias: Optional[torch.Tensor], tdup: bool = False) -> torch.Tensor:
    """Nearest x2 up-sampling + 3x3x3 causal convolution in sub-pixel form (ea_conv3d_cl_subpixel_bf16): x bf16 [T,H,W,Cin]
    (W % 256 == 0), w4 bf16 [4, Cout, 12*Cin] (vae_modules._pack_subpixel_weight) -> bf16 [T', 2H, 2W, Cout]; leaves the
    GroupNorm partial sums of its output on the result like conv3d_cl."""
    _dev(x, w4, bias)
    _chk(x, _BF16, "x"); _chk(w4, _BF16, "w4")
    assert x.is_contiguous() and w4.is_contiguous() and x.dim() == 4
    T, H, W, Cin = x.shape
    Cout = w4.shape[1]
    assert w4.shape == (4, Cout, 12 * Cin) and W % 256 == 0 and Cin % 64 == 0 and Cout % 256 == 0
    dup = int(tdup and T > 1)
    Ty = 2 * T - 1 if dup else T
    y = torch.empty((Ty, 2 * H, 2 * W, Cout), dtype=_BF16, device=x.device)
    cap = Ty * H * (W // 256) * 8 * (Cout // 4) * 2
    partial = torch.empty(cap, dtype=_F32, device=x.device)
    nblk = ctypes.c_int(0)
    _timed("conv3d", lambda: _lib.call("ea_conv3d_cl_subpixel_bf16", _p(x), _p(w4), _p(bias), _p(y), T, H, W, Cin, Cout, dup,
                                       _p(partial), cap, ctypes.byref(nblk), _stream()))
    if nblk.value:
        y.gn_partial = (partial, nblk.value)
    return y


def conv3d_narrow(x: torch.Tensor, wz: torch.Tensor, bias: Optional[torch.Tensor], c_out: int, c_pad: int) -> torch.Tensor:
    """3x3x3 / stride 1 / pad 1 causal convolution with C_out <= 4 as one GEMM (weight rows x voxels, fp32, voxel-minor) + a
    tap-gather pass (ea_conv3d_tap_gather_f32).  x bf16 [T,H,W,Cin]; wz bf16 [27*c_out, Cin] -> bf16 [T,H,W,c_pad]."""
    _dev(x, wz, bias)
    _chk(x, _BF16, "x"); _chk(wz, _BF16, "wz")
    assert x.is_contiguous() and x.dim() == 4 and wz.is_contiguous() and wz.shape == (27 * c_out, x.shape[-1])
    T, H, W, Cin = x.shape
    y = torch.empty((T, H, W, c_pad), dtype=_BF16, device=x.device)
    # the fp32 tap planes cost 27 * c_out * 4 bytes per voxel (16.6 GB at 49 x 1024^2): above NARROW_SCRATCH_BYTES the clip is
    # processed in frame chunks; a chunk brings the two frames in front of it along (the causal taps look back two frames) and
    # the outputs of those two are dropped
    per_frame = H * W * 27 * c_out * 4
    chunk = T if T * per_frame <= NARROW_SCRATCH_BYTES else max(4, NARROW_SCRATCH_BYTES // per_frame - 2)
    for a in range(0, T, chunk):
        b = min(T, a + chunk)
        lo = max(0, a - 2)
        xs = x[lo:b]
        n = b - lo
        z = gemm(wz, xs.reshape(n * H * W, Cin), None, EPI_F32_OUT)          # fp32 [27*c_out, voxels of the chunk]
        yc = y if (a == 0 and b == T) else torch.empty((n, H, W, c_pad), dtype=_BF16, device=x.device)
        _timed("conv3d", lambda: _lib.call("ea_conv3d_tap_gather_f32", _p(z), _p(bias), _p(yc), n, H, W, z.stride(0), c_out, c_pad, _stream()))
        if yc is not y:
            y[a:b] = yc[a - lo:]
        del z
    return y


NARROW_SCRATCH_BYTES = 9 << 30   # fp32 scratch bound of conv3d_narrow: 49 x 1024^2 runs as two chunks (8.8 GB instead of 16.6)


def im2col3d(x: torch.Tensor, k: int, st: int, ss: int, pad: int, k_pad: int):
    """x bf16 [T,H,W,Cin] -> (cols bf16 [M, k_pad], (T',H',W'))."""
    _dev(x)
    _chk(x, _BF16, "x")
    assert x.is_contiguous()
    T, H, W, Cin = x.shape
    To, Ho, Wo = conv_out_shape(T, H, W, k, st, ss, pad)
    cols = torch.empty((To * Ho * Wo, k_pad), dtype=_BF16, device=x.device)
    _lib.call("ea_im2col3d_bf16", _p(x), _p(cols), T, H, W, Cin, k, k, k, st, ss, pad, k_pad, _stream())
    return cols, (To, Ho, Wo)


def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                   act: bool = True, blocked: bool = False) -> torch.Tensor:
    """Per-frame GroupNorm (+SiLU) of x bf16 [T, H, W, C] (or [T, HW, C]).  When x came out of a convolution that already
    reduced its output (x.gn_partial, see conv3d_cl) only the finalize kernel runs; otherwise the statistics pass.
    blocked: the result is written channel-blocked, bf16 [C / 32, T, H, W, 32] (for conv3d_cl(..., blocked=True))."""
    _dev(x, gamma, beta)
    _chk(x, _BF16, "x"); _chk(gamma, _F32, "gamma"); _chk(beta, _F32, "beta")
    assert x.is_contiguous()
    T, C = x.shape[0], x.shape[-1]
    hw = x.numel() // (T * C)
    stats = torch.empty((T, groups, 2), dtype=_F32, device=x.device)
    fused = getattr(x, "gn_partial", None)
    if fused is not None and FUSED_GN_STATS:
        partial, nblk = fused
        _lib.call("ea_groupnorm_finalize_bf16", _p(partial), _p(stats), T, hw, C, groups, nblk, float(eps), _stream())
    else:
        nblk = max(1, min(256, hw // 2048))
        partial = torch.empty((T, nblk, C // 4, 2), dtype=_F32, device=x.device)
        _lib.call("ea_groupnorm_stats_bf16", _p(x), _p(partial), _p(stats), T, hw, C, groups, nblk, float(eps), _stream())
    if blocked:
        assert x.dim() == 4 and C % 32 == 0
        y = torch.empty((C // 32, T, x.shape[1], x.shape[2], 32), dtype=_BF16, device=x.device)
    else:
        y = torch.empty_like(x)
    _lib.call("ea_groupnorm_apply_bf16", _p(x), _p(y), _p(stats), _p(gamma), _p(beta), T, hw, C, groups, int(act) | (2 if blocked else 0), _stream())
    return y


FUSED_GN_STATS = True   # False: always run the separate statistics pass (A/B, tests)


def groupnorm_local_sums(x: torch.Tensor, groups: int) -> torch.Tensor:
    """fp64 [T, groups, 2] = (sum, sum of squares) of x bf16 [T, H, W, C] per frame and group over THIS tensor's voxels: the
    additive half of the GroupNorm statistics (a spatially split VAE all-reduces it over the ranks that share a frame)."""
    _dev(x)
    _chk(x, _BF16, "x")
    assert x.is_contiguous()
    T, C = x.shape[0], x.shape[-1]
    hw = x.numel() // (T * C)
    nblk = max(1, min(256, hw // 2048))
    partial = torch.empty((T, nblk, C // 4, 2), dtype=_F32, device=x.device)
    stats = torch.empty((T, groups, 2), dtype=_F32, device=x.device)
    _lib.call("ea_groupnorm_stats_bf16", _p(x), _p(partial), _p(stats), T, hw, C, groups, nblk, 1e-6, _stream())
    return partial.double().sum(1).view(T, groups, C // 4 // groups, 2).sum(2)


def groupnorm_stats_from_sums(sums: torch.Tensor, n: int, eps: float) -> torch.Tensor:
    """(sum, sumsq) fp64 [T, groups, 2] over n elements per (frame, group) -> fp32 [T, groups, 2] = (mean, rstd), the
    arithmetic of gn_finalize_kernel."""
    mean = sums[..., 0] / n
    var = (sums[..., 1] / n - mean * mean).clamp_min(0)
    return torch.stack([mean, 1.0 / torch.sqrt(var + eps)], -1).to(torch.float32).contiguous()


def groupnorm_apply(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int,
                    act: bool = True) -> torch.Tensor:
    """y = act((x - mean) * rstd * gamma + beta) with given per-(frame, group) statistics fp32 [T, groups, 2]."""
    _dev(x, stats, gamma, beta)
    _chk(x, _BF16, "x"); _chk(stats, _F32, "stats"); _chk(gamma, _F32, "gamma"); _chk(beta, _F32, "beta")
    assert x.is_contiguous() and stats.is_contiguous()
    T, C = x.shape[0], x.shape[-1]
    hw = x.numel() // (T * C)
    assert stats.shape == (T, groups, 2)
    y = torch.empty_like(x)
    _lib.call("ea_groupnorm_apply_bf16", _p(x), _p(y), _p(stats), _p(gamma), _p(beta), T, hw, C, groups, int(act), _stream())
    return y


def softmax_rows(x: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(x * scale) per row; x bf16 or fp32 [rows, cols] -> bf16."""
    _dev(x, out)
    assert x.is_contiguous() and x.dim() == 2 and x.dtype in (_BF16, _F32)
    if out is None:
        out = torch.empty(x.shape, dtype=_BF16, device=x.device)
    name = "ea_softmax_rows_f32in" if x.dtype == _F32 else "ea_softmax_rows_bf16"
    _lib.call(name, _p(x), _p(out), x.shape[0], x.shape[1], float(scale), _stream())
    return out


def ncdhw_to_ndhwc(x: torch.Tensor, c_pad: Optional[int] = None) -> torch.Tensor:
    """x [C,T,H,W] fp32/bf16 -> bf16 [T,H,W,c_pad]."""
    _dev(x)
    assert x.is_contiguous() and x.dim() == 4 and x.dtype in (_BF16, _F32)
    C, T, H, W = x.shape
    cp = c_pad or C
    y = torch.empty((T, H, W, cp), dtype=_BF16, device=x.device)
    _lib.call("ea_ncdhw_to_ndhwc", _p(x), _p(y), C, cp, T * H * W, int(x.dtype == _BF16), _stream())
    return y


def ndhwc_to_ncdhw(x: torch.Tensor, channels: int, out_dtype, post: int = 0) -> torch.Tensor:
    """x bf16 [T,H,W,Cs] -> [channels,T,H,W] in out_dtype (first `channels` channels)."""
    _dev(x)
    _chk(x, _BF16, "x")
    assert x.is_contiguous() and x.dim() == 4
    T, H, W, Cs = x.shape
    y = torch.empty((channels, T, H, W), dtype=out_dtype, device=x.device)
    _lib.call("ea_ndhwc_to_ncdhw", _p(x), _p(y), channels, Cs, T * H * W, int(out_dtype == _BF16), post, _stream())
    return y
