"""Per-layer timing of the VAE kernels on the GPU box: conv tiles A/B, GroupNorm apply.  python tools/microbench_vae.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops


def timeit(fn, warm=1, iters=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))[iters // 2]


zeros = None
for (T, H, W, Ci, Co) in [(13, 1024, 1024, 128, 128), (13, 1024, 1024, 256, 128), (13, 512, 512, 256, 256), (13, 256, 256, 512, 512)]:
    x = torch.randn(T, H, W, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 27 * Ci, device="cuda") / (27 * Ci) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    fl = 2.0 * 27 * Ci * Co * T * H * W
    for tile, mfma in (((512, 32), (1024, 32), (1024, 16)) if Co == 128 else ((256, 32), (1024, 32), (1024, 16))):
        _lib.set_option("conv_tile", tile)
        _lib.set_option("conv_mfma", mfma)
        ms = timeit(lambda: ops.conv3d_cl(x, w, b, 3, 1, 1, 1))
        print(json.dumps({"kernel": "conv3d_cl", "tile": tile, "mfma": mfma, "T": T, "HW": H, "Cin": Ci, "Cout": Co, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
    _lib.set_option("conv_tile", 0)
    _lib.set_option("conv_mfma", 16)
    del x, w
for (T, HW, C) in [(13, 1024 * 1024, 128), (13, 512 * 512, 256)]:
    x = torch.randn(T, HW, C, device="cuda").to(torch.bfloat16)
    g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ms = timeit(lambda: ops.groupnorm_silu(x, g, be, 32, 1e-6, act=True))
    print(json.dumps({"kernel": "groupnorm_silu (stats + apply)", "T": T, "HW": HW, "C": C, "ms": ms, "GBps(3 passes: 2 reads + 1 write)": 3 * x.numel() * 2 / ms / 1e6}), flush=True)
