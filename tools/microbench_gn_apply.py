"""GroupNorm + SiLU apply pass (statistics given) at the decoder's full-resolution shapes; EA_LIB_PATH selects the library."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from easyanimate_amd.ops import _p, _stream
from microbench_vae_common import timeit

lib = os.path.basename(os.environ.get("EA_LIB_PATH", "default"))
for (T, HW, C) in [(25, 1024 * 1024, 128), (25, 1024 * 1024, 256), (13, 512 * 512, 256), (13, 256 * 256, 512)]:
    x = torch.randn(T, HW, C, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    stats = torch.rand(T, 32, 2, device="cuda") + 0.5
    g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    fn = lambda: _lib.call("ea_groupnorm_apply_bf16", _p(x), _p(y), _p(stats), _p(g), _p(b), T, HW, C, 32, 1, _stream())
    for rep in range(2):
        ms = timeit(fn, warm=2, iters=7)
        print(json.dumps({"lib": lib, "kernel": "groupnorm apply + SiLU", "T": T, "HW": HW, "C": C, "ms": round(ms, 3), "TB/s": round(x.numel() * 4 / ms / 1e9, 2)}), flush=True)
    del x, y
