"""Fused QKV projection kernel vs the three GEMMs + ea_qknorm_rope_bf16 at the config-3 video-stream shape (default), or at
    python tools/microbench_qkv.py 52416     # tokens: e.g. the reference's published 768 x 1344 x 49 shape (204.75 x 256 rows)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import ops
from microbench_vae_common import timeit
B, M, K, H = 2, (int(sys.argv[1]) if len(sys.argv) > 1 else 53248), 3072, 48
d = H * 64
x = torch.randn(B, M, K, device="cuda").to(torch.bfloat16)
ws = [(torch.randn(d, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(3)]
bs = [torch.randn(d, device="cuda") * 0.1 for _ in range(3)]
nw = [torch.ones(64, device="cuda") for _ in range(2)]
nb = [torch.zeros(64, device="cuda") for _ in range(2)]
cos, sin = torch.rand(M, 64, device="cuda"), torch.rand(M, 64, device="cuda")
s_pad = ops.round_up(256 + M, 256)
q = torch.zeros(B, H, s_pad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q)
vt = torch.zeros(B, H, 64, s_pad, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * B * M * 3 * d * K
def fused():
    ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q, k, vt, nw[0], nb[0], nw[1], nb[1], cos, sin, 256, 1e-6, q_scale=ops.FOLDED_Q_SCALE)
qkv = torch.empty(B, M, 3 * d, device="cuda", dtype=torch.bfloat16)
def unfused():
    for i in range(3):
        ops.gemm(x, ws[i], bs[i], ops.EPI_BIAS, out=qkv[:, :, i * d:(i + 1) * d])
    ops.qknorm_rope(qkv, q, k, vt, nw[0], nb[0], nw[1], nb[1], cos, sin, 256, 1e-6, q_scale=ops.FOLDED_Q_SCALE)
for name, fn in (("fused", fused), ("unfused", unfused), ("fused", fused), ("unfused", unfused)):
    ms = timeit(fn, warm=2, iters=5)
    print(json.dumps({"M": M, "path": name, "ms": ms, "TFLOPs(gemm flops only)": fl / ms / 1e9}), flush=True)
