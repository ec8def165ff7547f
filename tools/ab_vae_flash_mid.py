"""In-process A/B: VAE decode + encode at 49 x 1024^2 with the mid-block attention on the head_dim-512 flash kernel vs the
three-GEMM route (vae_modules.FLASH_MID_BLOCK), alternating, same weights and inputs.
    python tools/ab_vae_flash_mid.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae
from easyanimate_amd import ops, vae_modules

vae = bench_vae.build_vae()
z = (torch.randn(1, 16, 13, 128, 128, device="cuda") / 0.1825).to(torch.bfloat16)
video = (torch.rand(1, 3, 49, 1024, 1024, device="cuda") * 2 - 1).to(torch.bfloat16)
with torch.no_grad():
    for rep in range(4):
        for flash in (True, False):
            vae_modules.FLASH_MID_BLOCK = flash
            res = {}
            for name, fn in (("decode", lambda: vae.decode(z)[0]), ("encode", lambda: vae.encode(video)[0].mode())):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with ops.KernelTimer("attention") as kt:
                    y = fn()
                torch.cuda.synchronize()
                res[name + "_s"] = round(time.perf_counter() - t0, 4)
                res[name + "_flash_kernel_ms"] = round(sum(kt.durations_ms()), 3)
                del y
            print(json.dumps({"flash_mid_block": flash, "rep": rep, **res, "decode_MPix_per_s": round(51.380224 / res["decode_s"], 2),
                              "encode_MPix_per_s": round(51.380224 / res["encode_s"], 2)}), flush=True)
vae_modules.FLASH_MID_BLOCK = True
