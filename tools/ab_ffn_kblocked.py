"""In-process A/B of the feed-forward GEMM pair at config-3 / config-2 size: row-major inner activation and weight against the
K-blocked form (ea_gemm_bf16_kblocked: the first GEMM writes [B, inner / 64, M, 64], the second reads it and a K-blocked weight),
alternating, bitwise comparison.      python tools/ab_ffn_kblocked.py
"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
if len(sys.argv) > 1:      # python tools/ab_ffn_kblocked.py 1  -> the four-wave hand-placed kernel (ea_set_option("gemm_w4a", 1))
    _lib.set_option("gemm_w4a", int(sys.argv[1]))
from microbench_vae_common import timeit

d, inner = 3072, 12288
for (B, M, what) in [(2, 53248, "config 3 (B = 2 x 53 248 video tokens)"), (2, 13312, "config 2 (B = 2 x 13 312)"), (1, 13312, "one rank of 8")]:
    x = torch.randn(B, M, d, device="cuda").to(torch.bfloat16)
    w1 = (torch.randn(inner, d, device="cuda") / math.sqrt(d)).to(torch.bfloat16)
    w2 = (torch.randn(d, inner, device="cuda") / math.sqrt(inner)).to(torch.bfloat16)
    b1, b2 = torch.randn(inner, device="cuda"), torch.randn(d, device="cuda")
    res = torch.randn(B, M, d, device="cuda").to(torch.bfloat16)
    gate = torch.randn(B, d, device="cuda")
    w2b = ops.to_kblocked(w2)
    h_r = torch.empty(B, M, inner, dtype=torch.bfloat16, device="cuda")
    h_b = torch.empty(B, inner // 64, M, 64, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(B, M, d, dtype=torch.bfloat16, device="cuda")
    up_r = lambda: ops.gemm(x, w1, b1, ops.EPI_BIAS_GELU_TANH, out=h_r)
    dn_r = lambda: ops.gemm(h_r, w2, b2, ops.EPI_BIAS_GATE_RES, out=y, res=res, gate=gate)
    up_b = lambda: ops.gemm_kblocked(x, w1, b1, ops.EPI_BIAS_GELU_TANH, ops.LAYOUT_C, out=h_b)
    dn_b = lambda: ops.gemm_kblocked(h_b, w2b, b2, ops.EPI_BIAS_GATE_RES, ops.LAYOUT_A | ops.LAYOUT_W, out=y, res=res, gate=gate)
    dn_bA = lambda: ops.gemm_kblocked(h_b, w2, b2, ops.EPI_BIAS_GATE_RES, ops.LAYOUT_A, out=y, res=res, gate=gate)
    up_r(); dn_r(); y_r = y.clone()
    up_b(); dn_b(); y_b = y.clone()
    print(json.dumps({"what": what, "bit_identical": bool(torch.equal(y_r, y_b)),
                      "inner_identical": bool(torch.equal(h_b.permute(0, 2, 1, 3).reshape(B, M, inner), h_r))}), flush=True)
    fl = 2.0 * B * M * d * inner
    for rep in range(3):
        for name, fn in (("FFN-up, row-major C", up_r), ("FFN-up, K-blocked C", up_b), ("FFN-down, row-major A / W", dn_r),
                         ("FFN-down, K-blocked A, row-major W", dn_bA), ("FFN-down, K-blocked A / W", dn_b)):
            ms = timeit(fn, warm=2, iters=7)
            print(json.dumps({"what": what, "kernel": name, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    del x, w1, w2, w2b, h_r, h_b, y, res
