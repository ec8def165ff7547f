"""One-kernel driver for rocprofv3 runs: python tools/prof_attn.py [attn|gemm] [variant]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops

what = sys.argv[1] if len(sys.argv) > 1 else "attn"
if what == "attn":
    var = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    _lib.set_option("attn_variant", var)
    # EA_PROF_ATTN_SHAPE="nw,stages": the workgroup shape of v3 (round 6: 8,2 = one K / V^T stream per CU; 4,3 = a three-stage ring)
    nw, stages = (int(v) for v in os.environ.get("EA_PROF_ATTN_SHAPE", "4,2").split(","))
    _lib.set_option("attn_nw", nw)
    _lib.set_option("attn_stages", stages)
    B, H, S = 2, 48, 53504
    s_pad = ops.round_up(S, 256)
    q = (torch.randn(B, H, s_pad, 64, device="cuda") * ops.FOLDED_Q_SCALE).to(torch.bfloat16)   # scale folded into Q
    k = torch.randn(B, H, s_pad, 64, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, H, 64, s_pad, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.attention(q, k, vt, S, ops.FOLDED_ATTN_SCALE, out=out)
else:
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    _lib.set_option("gemm_tile", tile)
    for (M, N, K, epi) in [(106496, 3072, 3072, 0), (106496, 12288, 3072, 1), (106496, 3072, 12288, 2)]:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        gate = torch.randn(1, N, device="cuda")
        for _ in range(3):
            if epi == 2:
                ops.gemm(A, W, bias, 2, out=o, res=o, gate=gate)
            else:
                ops.gemm(A, W, bias, epi, out=o)
        del A, W, o
torch.cuda.synchronize()
