"""GEMM timings at the DiT shapes of config 3; EA_LIB_PATH selects another library build for an A/B (one library per process).
    python tools/ab_gemm.py            # FFN-up (GELU), FFN-down (gate + residual), out-proj, fused QKV
"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from microbench_vae_common import timeit

lib = os.path.basename(os.environ.get("EA_LIB_PATH", "default"))
for (M, N, K, epi) in [(106496, 12288, 3072, 1), (106496, 3072, 12288, 2), (106496, 3072, 3072, 0), (512, 3072, 3072, 0)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    gate = torch.randn(1, N, device="cuda")
    fn = (lambda: ops.gemm(A, W, bias, 2, out=out, res=out, gate=gate)) if epi == 2 else (lambda: ops.gemm(A, W, bias, epi, out=out))
    W8 = W.to(torch.float8_e4m3fn)
    fn8 = (lambda: ops.gemm(A, W8, bias, 2, out=out, res=out, gate=gate)) if epi == 2 else (lambda: ops.gemm(A, W8, bias, epi, out=out))
    for rep in range(2):
        ms = timeit(fn8, warm=2, iters=7)
        print(json.dumps({"lib": lib, "kernel": "gemm, fp8-stored weight", "M": M, "N": N, "K": K, "epi": epi, "ms": round(ms, 4), "TFLOPs": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
        ms = timeit(fn, warm=2, iters=7)
        print(json.dumps({"lib": lib, "kernel": "gemm", "M": M, "N": N, "K": K, "epi": epi, "ms": round(ms, 4), "TFLOPs": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del A, W, out
B, M, K, H = 2, 53248, 3072, 48
d = H * 64
x = torch.randn(B, M, K, device="cuda").to(torch.bfloat16)
ws = [(torch.randn(d, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(3)]
bs = [torch.randn(d, device="cuda") * 0.1 for _ in range(3)]
nw = [torch.ones(64, device="cuda") for _ in range(2)]
nb = [torch.zeros(64, device="cuda") for _ in range(2)]
cos, sin = torch.rand(M, 64, device="cuda"), torch.rand(M, 64, device="cuda")
s_pad = 53504
q = torch.zeros(B, H, s_pad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q)
vt = torch.zeros(B, H, 64, s_pad, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * B * M * 3 * d * K
fused = lambda: ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q, k, vt, nw[0], nb[0], nw[1], nb[1], cos, sin, 256, 1e-6, q_scale=ops.FOLDED_Q_SCALE)
w8s = [w.to(torch.float8_e4m3fn) for w in ws]
fused8 = lambda: ops.qkv_gemm_norm_rope(x, w8s[0], w8s[1], w8s[2], bs[0], bs[1], bs[2], q, k, vt, nw[0], nb[0], nw[1], nb[1], cos, sin, 256, 1e-6, q_scale=ops.FOLDED_Q_SCALE)
for rep in range(2):
    ms = timeit(fused8, warm=2, iters=7)
    print(json.dumps({"lib": lib, "kernel": "qkv fused, fp8-stored weights", "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    ms = timeit(fused, warm=2, iters=7)
    print(json.dumps({"lib": lib, "kernel": "qkv fused", "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
