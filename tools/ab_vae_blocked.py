"""Whole VAE passes at 49 x 1024^2 with and without the channel-blocked GroupNorm -> convolution edges (vae_modules.BLOCKED_GN_OUTPUT),
alternating in one process; checksums of the outputs must agree (bit-identical).      python tools/ab_vae_blocked.py [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae
from easyanimate_amd import _lib, vae_modules

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
vae = bench_vae.build_vae()
g = torch.Generator().manual_seed(3)
z = (torch.randn(1, 16, 13, 128, 128, generator=g) / 0.1825).to("cuda").bfloat16()
video = (torch.rand(1, 3, 49, 1024, 1024, generator=g) * 2 - 1).to("cuda").bfloat16()
sums = {}
for rep in range(reps + 1):
    for flag in (False, True):
        vae_modules.BLOCKED_GN_OUTPUT = flag
        res = {}
        for name, fn in (("decode", lambda: vae.decode(z)[0]), ("encode", lambda: vae.encode(video)[0].parameters)):
            _lib.reset_counters()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                y = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name] = {"s": round(dt, 4), "MPix_s": round(49 * 1024 * 1024 / 1e6 / dt, 2), "blocked_convs": _lib.counters().get("conv_blocked_input", 0),
                         "checksum": float(y.float().double().sum().item())}
            del y
        if rep:      # the first pass warms both paths up
            print(json.dumps({"blocked_gn_output": flag, **res}), flush=True)
