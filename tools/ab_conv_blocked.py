"""The four-wave row-slab convolutions on a CHANNEL-BLOCKED input ([C/32][T][H][W][32], ops.conv3d_cl(..., blocked=True)) against the
voxel-major input, at the shapes of the 49 x 1024^2 VAE passes: bit-identity and alternating timings.   python tools/ab_conv_blocked.py [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from easyanimate_amd.vae_modules import _pack_conv_weight
from microbench_vae_common import timeit

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
#          T   H     W    Cin  Cout
SHAPES = [(8, 1024, 1024, 128, 128), (8, 512, 512, 256, 256), (8, 512, 512, 128, 256), (8, 256, 256, 512, 512), (8, 256, 256, 256, 512)]
for T, H, W, Ci, Co in SHAPES:
    assert ops.conv3d_blocked_ok(T, H, W, Ci, Co), (T, H, W, Ci, Co)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(T, H, W, Ci, device="cuda", generator=g).to(torch.bfloat16)
    xb = x.view(T, H, W, Ci // 32, 32).permute(3, 0, 1, 2, 4).contiguous()
    w = (torch.randn(Co, Ci, 3, 3, 3, device="cuda", generator=g) / (Ci * 27) ** 0.5).to(torch.bfloat16)
    wp = _pack_conv_weight(w.cpu()).cuda()
    b = torch.randn(Co, device="cuda", generator=g)
    _lib.reset_counters()
    y0 = ops.conv3d_cl(x, wp, b, 3)
    y1 = ops.conv3d_cl(xb, wp, b, 3, blocked=True)
    torch.cuda.synchronize()
    cnt = _lib.counters()
    same = bool(torch.equal(y0, y1)) and bool(torch.equal(y0.gn_partial[0][:8192], y1.gn_partial[0][:8192]))
    print(json.dumps({"shape": [T, H, W, Ci, Co], "bit_identical": same, "counters": cnt}), flush=True)
    fl = 2.0 * 27 * Ci * Co * T * H * W
    for rep in range(reps):
        for name, fn in (("voxel-major", lambda: ops.conv3d_cl(x, wp, b, 3)), ("channel-blocked", lambda: ops.conv3d_cl(xb, wp, b, 3, blocked=True))):
            ms = timeit(fn, warm=1, iters=5)
            print(json.dumps({"shape": [T, H, W, Ci, Co], "input": name, "ms": round(ms, 3), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)
    del x, xb, y0, y1
