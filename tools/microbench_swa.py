"""Sliding-window block at config-3 size (13 x 64 x 64 token grid, 48 heads, B = 2): where does the time of
EasyAnimateSWAttnProcessor2_0._swa go -- the two attention kernels or the torch index copies around them (strided cross-key
gather, six scan-order permutations of q / k / v^T and the inverse scatter)?  VERDICT r2 weak #15.
    python tools/microbench_swa.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import ops
from easyanimate_amd.processor import EasyAnimateSWAttnProcessor2_0

B, H, T = 2, 48, 256
F_, Hh, Ww = 13, 64, 64
N = F_ * Hh * Ww
S = T + N
s_pad = ops.round_up(S, 256)
dev = "cuda"
q = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=dev)
k = torch.zeros_like(q)
vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=dev)
q[:, :, :S] = (torch.randn(B, H, S, 64, device=dev) * 0.3 * ops.FOLDED_Q_SCALE).to(torch.bfloat16)
k[:, :, :S] = torch.randn(B, H, S, 64, device=dev).to(torch.bfloat16)
vt[:, :, :, :S] = torch.randn(B, H, 64, S, device=dev).to(torch.bfloat16)
proc = EasyAnimateSWAttnProcessor2_0()
with torch.no_grad():
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with ops.KernelTimer("attention") as kt:
            o = proc._swa(q, k, vt, B, H, 0, H, T, N, dev, (F_, Hh, Ww))
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) * 1e3
        kern = sum(kt.durations_ms())
        full = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.attention(q, k, vt, S, ops.FOLDED_ATTN_SCALE, out=full)
        torch.cuda.synchronize()
        t_full = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"rep": rep, "swa_attend_total_ms": round(total, 2), "attention_kernels_ms": round(kern, 2),
                          "index_copies_ms": round(total - kern, 2), "full_attention_same_shape_ms": round(t_full, 2),
                          "copies_share_of_a_full_block(74ms)": round((total - kern) / 74.0, 3)}), flush=True)
