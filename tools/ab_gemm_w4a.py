"""In-process A/B of the four-wave hand-placed GEMM (ea_set_option("gemm_w4a", 1)) against the eight-wave product kernel at the DiT
shapes of config 3 / config 2 and 8192^3, alternating, bitwise comparison.      python tools/ab_gemm_w4a.py [reps]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from microbench_vae_common import timeit

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d, inner = 3072, 12288
SHAPES = [("FFN-up c3", 2, 53248, inner, d, ops.EPI_BIAS_GELU_TANH), ("FFN-down c3", 2, 53248, d, inner, ops.EPI_BIAS_GATE_RES),
          ("out-proj c3", 2, 53248, d, d, ops.EPI_BIAS_GATE_RES), ("plain QKV-width c3", 2, 53248, 3 * d, d, ops.EPI_BIAS),
          ("FFN-up c2", 2, 13312, inner, d, ops.EPI_BIAS_GELU_TANH), ("FFN-down c2", 2, 13312, d, inner, ops.EPI_BIAS_GATE_RES),
          ("FFN-up 1 rank of 8", 1, 13312, inner, d, ops.EPI_BIAS_GELU_TANH), ("8192^3", 1, 8192, 8192, 8192, ops.EPI_BIAS)]
for what, B, M, N, K, epi in SHAPES:
    x = torch.randn(B, M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(B, M, N, device="cuda").to(torch.bfloat16) if epi == ops.EPI_BIAS_GATE_RES else None
    gate = torch.randn(B, N, device="cuda") if epi == ops.EPI_BIAS_GATE_RES else None
    y = torch.empty(B, M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: ops.gemm(x, w, bias, epi, out=y, res=res, gate=gate)
    outs = {}
    for v in (0, 1):
        _lib.set_option("gemm_w4a", v)
        fn()
        torch.cuda.synchronize()
        outs[v] = y.clone()
    print(json.dumps({"what": what, "bit_identical": bool(torch.equal(outs[0], outs[1])), "finite": bool(torch.isfinite(outs[1].float()).all())}), flush=True)
    fl = 2.0 * B * M * N * K
    for rep in range(reps):
        for v, name in ((0, "eight-wave (product)"), (1, "four-wave hand-placed")):
            _lib.set_option("gemm_w4a", v)
            ms = timeit(fn, warm=2, iters=7)
            print(json.dumps({"what": what, "kernel": name, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    _lib.set_option("gemm_w4a", 0)
    del x, w, y, res
