"""Which hipBLASLt kernel torch.matmul picks for the DiT GEMM shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
for (M, N, K) in [(8192, 8192, 8192), (106496, 12288, 3072), (106496, 3072, 12288), (106496, 3072, 3072)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        torch.matmul(a, w.t(), out=c)
    torch.cuda.synchronize()
    del a, w, c
