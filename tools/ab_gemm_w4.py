"""In-process A/B of the two 256^2 bf16 GEMM kernels at the DiT shapes of config 3: the eight-wave ping-pong kernel
(gemm256_mi16_kernel, 128 x 64 wave tiles) against the four-wave kernel (gemm256_w4_kernel, 128 x 128 wave tiles, accumulators
in AGPRs; both of its main-loop schedules), alternating, with a bitwise comparison of the outputs (same products summed in the same order).
    python tools/ab_gemm_w4.py
"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from microbench_vae_common import timeit

if _lib.get_option("build_variants") != 1:
    raise SystemExit("the four-wave kernel is compiled into EA_BUILD_VARIANTS=1 libraries only: EA_BUILD_VARIANTS=1 python -m easyanimate_amd.build --force")

SHAPES = [(106496, 12288, 3072, 1, "FFN-up (GELU)"), (106496, 3072, 12288, 2, "FFN-down (gate + residual)"),
          (106496, 3072, 3072, 2, "out-proj (gate + residual)"), (8192, 8192, 8192, 0, "8192^3"), (13312, 12288, 3072, 1, "FFN-up, one rank of 8")]
for (M, N, K, epi, what) in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    gate = torch.randn(1, N, device="cuda")
    outs = {}
    for w4 in (0, 1, 2):
        _lib.set_option("gemm_w4", w4)
        _lib.reset_counters()
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        if epi == 2:
            ops.gemm(A, W, bias, 2, out=out, res=res, gate=gate)
        else:
            ops.gemm(A, W, bias, epi, out=out)
        torch.cuda.synchronize()
        outs[w4] = (out, dict(_lib.counters()))
    same = [bool(torch.equal(outs[0][0], outs[w][0])) for w in (1, 2)]
    print(json.dumps({"what": what, "M": M, "N": N, "K": K, "epi": epi, "bit_identical": same, "kernels": [outs[w][1] for w in (0, 1, 2)],
                      "finite": bool(torch.isfinite(outs[2][0].float()).all())}), flush=True)
    out = outs[1][0]
    fn = (lambda: ops.gemm(A, W, bias, 2, out=out, res=res, gate=gate)) if epi == 2 else (lambda: ops.gemm(A, W, bias, epi, out=out))
    for rep in range(3):
        for w4 in (0, 1, 2):
            _lib.set_option("gemm_w4", w4)
            ms = timeit(fn, warm=2, iters=7)
            print(json.dumps({"what": what, "kernel": ["mi16 (8 waves x 128x64)", "w4 (4 waves x 128x128)", "w4, second schedule"][w4], "ms": round(ms, 4),
                              "TFLOPs": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del A, W, res, outs, out
_lib.set_option("gemm_w4", 0)
