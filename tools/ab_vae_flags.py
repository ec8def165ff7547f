"""In-process A/B of the VAE's algebraic shortcuts at 49 x 1024^2 decode: vae_modules.SUBPIXEL_UPSAMPLE (up-samplers as four
12-tap parity classes) and vae_modules.TEMPORAL_TAP_MERGE (18 merged taps behind a virtual temporal x2), alternating
configurations, same weights and input; reports the decode time, the change of the output against the all-off run (the
bf16 rounding of the summed weights) and the conv kernels that served each configuration.
    python tools/ab_vae_flags.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae
from easyanimate_amd import _lib, vae_modules

vae = bench_vae.build_vae()
z = (torch.randn(1, 16, 13, 128, 128, device="cuda") / 0.1825).to(torch.bfloat16)
CONFIGS = [("all off", False, False), ("sub-pixel up-samplers", True, False), ("merged temporal taps", False, True), ("both", True, True)]
base = None
with torch.no_grad():
    for rep in range(3):
        for name, sub, tm in CONFIGS:
            vae_modules.SUBPIXEL_UPSAMPLE, vae_modules.TEMPORAL_TAP_MERGE = sub, tm
            _lib.reset_counters()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = vae.decode(z)[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rec = {"config": name, "rep": rep, "decode_s": round(dt, 4), "MPix_per_s": round(51.380224 / dt, 2)}
            if rep == 0:
                sub_y = y[:, :, ::4, ::8, ::8].float().clone()
                if base is None:
                    base = sub_y
                rec["mse_vs_all_off"] = float(((sub_y - base) ** 2).mean())
                rec["output_std"] = float(base.std())
                rec["conv_kernels"] = {k: v for k, v in _lib.counters().items() if k.startswith("conv")}
            print(json.dumps(rec), flush=True)
            del y
vae_modules.SUBPIXEL_UPSAMPLE = vae_modules.TEMPORAL_TAP_MERGE = True
