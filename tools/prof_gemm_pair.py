"""One-shape driver for rocprofv3 counter runs: this library's two 256 x 256 GEMM kernels (the four-wave hand-placed one, the eight-wave
one) and the vendor kernel torch.matmul dispatches (hipBLASLt) on the same operands, three launches each, plain bias-free epilogue.
    python tools/prof_gemm_pair.py [M N K]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (106496, 12288, 3072)
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for v in (3, 0):
    _lib.set_option("gemm_w4a", v)
    for _ in range(3):
        ops.gemm(a, w, None, 0, out=c)
_lib.set_option("gemm_w4a", 3)
for _ in range(3):
    torch.matmul(a, w.t(), out=c)
torch.cuda.synchronize()
