"""In-process A/B of the four-wave hand-placed row-slab convolution (ea_set_option("conv_w4a", 3)) against the eight-wave kernels at
the layer shapes of a 49 x 1024^2 decode (4 frames per chunk), alternating, bitwise comparison.      python tools/ab_conv_w4a.py [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from easyanimate_amd.vae_modules import _pack_conv_weight
from microbench_vae_common import timeit

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
# (what, T, H, W, Cin, Cout, residual, eight-wave conv_m512 setting)
SHAPES = [("128 -> 128 @ 1024^2 (512 x 128 tiles)", 4, 1024, 1024, 128, 128, True, 1), ("256 -> 128 @ 1024^2 (512 x 128 tiles)", 4, 1024, 1024, 256, 128, False, 1),
          ("256 -> 256 @ 512^2 (256 x 256 tiles)", 4, 512, 512, 256, 256, True, 1), ("512 -> 256 @ 512^2 (256 x 256 tiles)", 4, 512, 512, 512, 256, False, 1),
          ("512 -> 512 @ 256^2 (256 x 256 tiles)", 4, 256, 256, 512, 512, True, 1)]
for what, T, H, W, Ci, Co, res, m512 in SHAPES:
    x = torch.randn(T, H, W, Ci, device="cuda").to(torch.bfloat16)
    w = _pack_conv_weight((torch.randn(Co, Ci, 3, 3, 3) / (Ci * 27) ** 0.5).to(torch.bfloat16)).cuda()
    b = torch.randn(Co, device="cuda")
    r = torch.randn(T, H, W, Co, device="cuda").to(torch.bfloat16) if res else None
    fn = lambda: ops.conv3d_cl(x, w, b, 3, 1, 1, 1, res=r)
    outs, kern = {}, {}
    for v in (0, 3):
        _lib.set_option("conv_w4a", v)
        _lib.reset_counters()
        y = fn()
        torch.cuda.synchronize()
        kern[v] = _lib.counters()
        outs[v] = y.clone()
    mx = (outs[0].float() - outs[3].float()).abs().max().item()
    print(json.dumps({"what": what, "bit_identical": bool(torch.equal(outs[0], outs[3])), "max_abs_diff": mx, "kernels": kern}), flush=True)
    fl = 2.0 * T * H * W * Co * Ci * 27
    for rep in range(reps):
        for v, name in ((0, "eight-wave (product)"), (3, "four-wave hand-placed")):
            _lib.set_option("conv_w4a", v)
            ms = timeit(fn, warm=2, iters=5)
            print(json.dumps({"what": what, "kernel": name, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    _lib.set_option("conv_w4a", 3)
    del x, w, r, outs
