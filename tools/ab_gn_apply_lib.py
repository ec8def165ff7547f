"""GroupNorm-apply (+SiLU) at the VAE's full-resolution shape (4 frames x 1024^2 x 128 channels) with the library EA_LIB_PATH names:
time, effective HBM rate (4 B per element) and a checksum of the output (two builds that print the same checksum on the same seed are
bit-identical).      EA_LIB_PATH=... [EA_GN_CLIP=1] python tools/ab_gn_apply_lib.py
EA_GN_CLIP=1: the whole-clip shapes a 49 x 1024^2 decode launches (T = 49 / 25 / 13) instead of 4-frame chunks."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import ops
from microbench_vae_common import timeit

lib = os.path.basename(os.environ.get("EA_LIB_PATH", "default"))
torch.manual_seed(11)
SHAPES = ((49, 1024, 1024, 128, 32), (25, 512, 512, 256, 32), (13, 256, 256, 512, 32)) if os.environ.get("EA_GN_CLIP") else ((4, 1024, 1024, 128, 32), (4, 512, 512, 256, 32), (4, 256, 256, 512, 32))
for T, H, W, C, G in SHAPES:
    x = (torch.randn(T, H, W, C, device="cuda") * 2 + 0.5).to(torch.bfloat16)
    gamma = 1 + 0.1 * torch.randn(C, device="cuda")
    beta = 0.1 * torch.randn(C, device="cuda")
    stats = torch.stack([0.3 * torch.randn(T, G, device="cuda"), 0.5 + torch.rand(T, G, device="cuda")], -1).contiguous()
    for act in ((True,) if os.environ.get("EA_GN_CLIP") else (True, False)):
        y = ops.groupnorm_apply(x, stats, gamma, beta, G, act=act)
        torch.cuda.synchronize()
        h = hashlib.sha256(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
        ms = timeit(lambda: ops.groupnorm_apply(x, stats, gamma, beta, G, act=act), warm=2, iters=10)
        print(json.dumps({"lib": lib, "shape": [T, H, W, C], "silu": act, "ms": round(ms, 4), "TB_per_s": round(x.numel() * 4 / ms / 1e9, 3), "sha256_16": h}), flush=True)
    del x
