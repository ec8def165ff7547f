"""TEST INFRASTRUCTURE ONLY.  `flash_attn.flash_attn_func` restated in plain PyTorch so that the reference's UNCHANGED
EasyAnimateSWAttnProcessor2_0 (/root/reference/easyanimate/models/processor.py:320-459) can run on the CPU: flash_attn
(Dao-AILab, `pip install flash-attn`, version not pinned by the reference's requirements.txt -- the import is wrapped in
try/except at processor.py:314-318) is a CUDA-only package that is absent here, so this piece of the SWA path is
"parity unpinned": the semantics below are the published ones of flash-attn >= 2.3 --

    flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1))
      q: [B, Sq, H, D], k / v: [B, Sk, H, D] -> [B, Sq, H, D];  softmax_scale defaults to D ** -0.5;
      window_size = (left, right): query i attends key j iff  i + (Sk - Sq) - left <= j <= i + (Sk - Sq) + right
      (a negative side is unbounded); scores and softmax in fp32, output in the input dtype.

oracle/gen_golden.py assigns it to `easyanimate.models.processor.flash_attn_func` before running the reference processor;
everything else in that processor (the strided cross keys, the six scan orders, the text-row doubling) is reference code."""
import torch


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), **unused):
    assert dropout_p == 0.0 and not causal
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    scale = D ** -0.5 if softmax_scale is None else softmax_scale
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
    left, right = window_size
    if left >= 0 or right >= 0:
        i = torch.arange(Sq)[:, None] + (Sk - Sq)
        j = torch.arange(Sk)[None, :]
        keep = torch.ones(Sq, Sk, dtype=torch.bool)
        if left >= 0:
            keep &= j >= i - left
        if right >= 0:
            keep &= j <= i + right
        s = s.masked_fill(~keep[None, None], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v.float()).to(q.dtype)
