"""Per-layer timing of the row-slab convolution at the VAE's large-layer shapes (run once per library build:
EA_LIB_PATH=... for A/B).    python tools/microbench_conv_rows.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from microbench_vae_common import timeit

for (T, H, W, Ci, Co) in [(13, 1024, 1024, 128, 128), (13, 1024, 1024, 256, 128), (13, 512, 512, 256, 256), (13, 256, 256, 512, 512)]:
    x = torch.randn(T, H, W, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 27 * Ci, device="cuda") / (27 * Ci) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    fl = 2.0 * 27 * Ci * Co * T * H * W
    res = torch.randn(T, H, W, Co, device="cuda").to(torch.bfloat16) if Ci == Co else None
    for m512 in (3, 0, 3, 0):
        _lib.set_option("conv_m512", m512)
        for r in ((None, res) if res is not None else (None,)):
            ms = timeit(lambda: ops.conv3d_cl(x, w, b, 3, 1, 1, 1, res=r, want_stats=r is not None))
            print(json.dumps({"kernel": "conv3d_cl row16", "lib": os.environ.get("EA_LIB_PATH", "default"), "m512": m512, "res+stats": r is not None,
                              "T": T, "HW": H, "Cin": Ci, "Cout": Co, "ms": round(ms, 3), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    _lib.set_option("conv_m512", 1)
    del x, w
