"""Context numbers on the same box, same session: this library's kernels next to the vendor libraries that ship with
PyTorch-ROCm (hipBLASLt through torch.matmul, the flash SDPA backend through F.scaled_dot_product_attention).
Not part of the product or of any test -- a yardstick for DESIGN.md.   python tools/vendor_compare.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from easyanimate_amd import ops


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
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))[iters // 2]


dev = "cuda"
for (M, N, K) in [(8192, 8192, 8192), (106496, 12288, 3072), (106496, 3072, 12288), (106496, 3072, 3072)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    for rep in range(2):
        ms = timeit(lambda: ops.gemm(a, w, bias, out=c))
        print(json.dumps({"op": "gemm", "impl": "ea_gemm_bf16", "M": M, "N": N, "K": K, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
        ms = timeit(lambda: torch.matmul(a, w.t(), out=c))
        print(json.dumps({"op": "gemm", "impl": "torch.matmul (hipBLASLt)", "M": M, "N": N, "K": K, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
    del a, w, c
B, H, S = 1, 48, 53504
q = torch.randn(B, H, S, 64, device=dev).to(torch.bfloat16)
k = torch.randn(B, H, S, 64, device=dev).to(torch.bfloat16)
v = torch.randn(B, H, S, 64, device=dev).to(torch.bfloat16)
vt = v.transpose(2, 3).contiguous()
qf = (q.float() * ops.FOLDED_Q_SCALE).to(torch.bfloat16)
out = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device=dev)
fl = 4.0 * B * H * S * S * 64
for rep in range(2):
    ms = timeit(lambda: ops.attention(qf, k, vt, S, ops.FOLDED_ATTN_SCALE, out=out), warm=1, iters=3)
    print(json.dumps({"op": "attention", "impl": "ea_attention_fwd_bf16", "B": B, "H": H, "S": S, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
    try:
        ms = timeit(lambda: F.scaled_dot_product_attention(q, k, v), warm=1, iters=3)
        print(json.dumps({"op": "attention", "impl": "F.scaled_dot_product_attention (torch " + torch.__version__ + ")", "B": B, "H": H, "S": S, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
    except Exception as e:  # noqa
        print(json.dumps({"op": "attention", "impl": "F.scaled_dot_product_attention", "error": str(e)[:200]}), flush=True)
