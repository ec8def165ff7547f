"""Per-kernel timing on the GPU box (writes JSON lines to stdout). Usage: python tools/microbench.py [gemm] [attn] [ln]"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from easyanimate_amd import ops

DEV = "cuda"


def timeit(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    return sorted(ts)[len(ts) // 2]


def bench_gemm():
    from easyanimate_amd import _lib
    _lib.set_option("gemm_tile", 256)
    for mfma in (32, 16, 32, 16):
        _lib.set_option("gemm_mfma", mfma)
        _bench_gemm(f"256/mfma{mfma}")
    _lib.set_option("gemm_tile", 0)
    _lib.set_option("gemm_mfma", 16)


def _bench_gemm(tile):
    for (M, N, K, epi) in [(106496 + 512, 3072, 3072, 0), (106496, 12288, 3072, 1), (106496, 3072, 12288, 2), (8192, 8192, 8192, 0), (4096, 4096, 4096, 0), (512, 3072, 3072, 0)]:
        A = (torch.randn(M, K, device=DEV) ).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        gate = torch.randn(1, N, device=DEV)
        if epi == 2:
            fn = lambda: ops.gemm(A, W, bias, 2, out=out, res=out, gate=gate)
        else:
            fn = lambda: ops.gemm(A, W, bias, epi, out=out)
        ms = timeit(fn)
        print(json.dumps({"kernel": "gemm", "tile": tile, "M": M, "N": N, "K": K, "epi": epi, "ms": ms, "TFLOPs": 2.0 * M * N * K / ms / 1e9}), flush=True)
        del A, W, out


def bench_attn():
    from easyanimate_amd import _lib
    _bench_attn(_lib.get_option("attn_variant"))


def _bench_attn(var, waves=0):
    for (B, H, S) in [(2, 48, 13568), (1, 48, 53504), (2, 8, 53504)]:
        s_pad = ops.round_up(S, 256)
        q = torch.randn(B, H, s_pad, 64, device=DEV).to(torch.bfloat16)
        k = torch.randn(B, H, s_pad, 64, device=DEV).to(torch.bfloat16)
        vt = torch.randn(B, H, 64, s_pad, device=DEV).to(torch.bfloat16)
        out = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device=DEV)
        for folded in ((False, True) if var == 2 else (False,)):
            qq = (q.float() * ops.FOLDED_Q_SCALE).to(torch.bfloat16) if folded else q
            sc = ops.FOLDED_ATTN_SCALE if folded else 0.125
            fn = lambda: ops.attention(qq, k, vt, S, sc, out=out)
            ms = timeit(fn, warm=1, iters=3)
            print(json.dumps({"kernel": "attention", "variant": var, "waves": waves, "folded": folded, "B": B, "H": H, "S": S, "ms": ms, "TFLOPs": 4.0 * B * H * S * S * 64 / ms / 1e9}), flush=True)
        continue
        print(json.dumps({"kernel": "attention", "variant": var, "B": B, "H": H, "S": S, "ms": ms, "TFLOPs": 4.0 * B * H * S * S * 64 / ms / 1e9}), flush=True)
        del q, k, vt, out


def bench_ln():
    B, R, D = 2, 53248, 3072
    x = torch.randn(B, R, D, device=DEV).to(torch.bfloat16)
    y = torch.empty_like(x)
    gamma = torch.ones(D, device=DEV); beta = torch.zeros(D, device=DEV)
    tab = torch.randn(B, 6 * D, device=DEV)
    fn = lambda: ops.layernorm_modulate(x, gamma, beta, tab[:, D:2 * D], tab[:, :D], 1e-6, out=y)
    ms = timeit(fn)
    print(json.dumps({"kernel": "layernorm_modulate", "rows": B * R, "D": D, "ms": ms, "GBps": 2 * x.numel() * 2 / ms / 1e6}), flush=True)
    H = 48
    qkv = torch.randn(B, R, 3 * D, device=DEV).to(torch.bfloat16)
    s_pad = ops.round_up(R + 256, 256)
    q = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV); k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(64, device=DEV); bz = torch.zeros(64, device=DEV)
    cos = torch.rand(R, 64, device=DEV); sin = torch.rand(R, 64, device=DEV)
    fn = lambda: ops.qknorm_rope(qkv, q, k, vt, w, bz, w, bz, cos, sin, 256, 1e-6)
    ms = timeit(fn)
    print(json.dumps({"kernel": "qknorm_rope", "tokens": B * R, "ms": ms, "GBps": 2 * qkv.numel() * 2 / ms / 1e6}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ln", "gemm", "attn"]
    print(json.dumps({"device": torch.cuda.get_device_name(0), "cpus": os.cpu_count()}), flush=True)
    if "ln" in which:
        bench_ln()
    if "gemm" in which:
        bench_gemm()
    if "attn" in which:
        bench_attn()
