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
        _lib.set_option("gemm_w4a", 3 * v)
        fn()
        torch.cuda.synchronize()
        outs[v] = y.clone()
    print(json.dumps({"what": what, "bit_identical": bool(torch.equal(outs[0], outs[1])), "finite": bool(torch.isfinite(outs[1].float()).all())}), flush=True)
    fl = 2.0 * B * M * N * K
    for rep in range(reps):
        for v, name in ((0, "eight-wave (product)"), (1, "four-wave hand-placed")):
            _lib.set_option("gemm_w4a", 3 * v)
            ms = timeit(fn, warm=2, iters=7)
            print(json.dumps({"what": what, "kernel": name, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    _lib.set_option("gemm_w4a", 3)
    del x, w, y, res

# ---- the fused QKV projection (config-3 video stream, one rank of four, config 2)
for what, B, M in (("fused QKV c3 video", 2, 53248), ("fused QKV one rank of 4", 1, 13312), ("fused QKV c2 video", 2, 13312)):
    H, K = 48, 3072
    dd = H * 64
    x = torch.randn(B, M, K, device="cuda").to(torch.bfloat16)
    ws = [(torch.randn(dd, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16) for _ in range(3)]
    bs = [torch.randn(dd, device="cuda") * 0.1 for _ in range(3)]
    n1 = [torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"), torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")]
    ang = torch.rand(M, 32, device="cuda") * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
    s_pad = ops.round_up(256 + M, 256)
    q = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device="cuda")
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device="cuda")
    fn = lambda: ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q, k, vt, n1[0], n1[1], n1[2], n1[3], cos, sin, 256, 1e-6,
                                        q_scale=ops.FOLDED_Q_SCALE)
    outs = {}
    for v in (0, 1):
        _lib.set_option("gemm_w4a", 3 * v)
        fn()
        torch.cuda.synchronize()
        outs[v] = (q.clone(), k.clone(), vt.clone())
    print(json.dumps({"what": what, "vt_bit_identical": bool(torch.equal(outs[0][2], outs[1][2])),
                      "q_k_elements_differing": [int((a != b).sum().item()) for a, b in zip(outs[0][:2], outs[1][:2])],
                      "q_k_max_abs_diff": [float((a.float() - b.float()).abs().max().item()) for a, b in zip(outs[0][:2], outs[1][:2])]}), flush=True)
    fl = 2.0 * B * M * 3 * dd * K
    for rep in range(reps):
        for v, name in ((0, "eight-wave (product)"), (1, "four-wave hand-placed")):
            _lib.set_option("gemm_w4a", 3 * v)
            ms = timeit(fn, warm=2, iters=7)
            print(json.dumps({"what": what, "kernel": name, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    _lib.set_option("gemm_w4a", 3)
    del x, ws, q, k, vt
