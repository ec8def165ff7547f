"""In-process A/B of the four-wave hand-placed row-slab convolutions inside the whole VAE at 49 x 1024^2: conv_w4a = 0 (eight-wave
kernels), 1 (512 x 128 tiles), 3 (also the 256 x 256 tiles), alternating, same weights and inputs; decode and encode times, the
change of the outputs against conv_w4a = 0 and the conv kernels that served each setting.      python tools/ab_vae_w4a.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae
from easyanimate_amd import _lib

vae = bench_vae.build_vae()
z = (torch.randn(1, 16, 13, 128, 128, device="cuda") / 0.1825).to(torch.bfloat16)
video = (torch.rand(1, 3, 49, 1024, 1024, device="cuda") * 2 - 1).to(torch.bfloat16)
base = {}
with torch.no_grad():
    for rep in range(3):
        for v in (0, 1, 3):
            _lib.set_option("conv_w4a", v)
            for what, fn in (("decode", lambda: vae.decode(z)[0]), ("encode", lambda: vae.encode(video)[0].parameters)):
                _lib.reset_counters()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                y = fn()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                rec = {"conv_w4a": v, "pass": what, "rep": rep, "seconds": round(dt, 4), "MPix_per_s": round(51.380224 / dt, 2)}
                if rep == 0:
                    sub = (y[:, :, ::4, ::8, ::8] if what == "decode" else y).float().clone()
                    base.setdefault(what, sub)
                    rec["mse_vs_w4a_0"] = float(((sub - base[what]) ** 2).mean())
                    rec["output_std"] = float(base[what].std())
                    rec["conv_kernels"] = {k: n for k, n in _lib.counters().items() if k.startswith("conv")}
                print(json.dumps(rec), flush=True)
                del y
_lib.set_option("conv_w4a", 3)
