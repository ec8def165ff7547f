"""In-process A/B: VAE decode at 49 x 1024^2 with the up-samplers' temporal duplication kept virtual vs materialised
(vae_modules.VIRTUAL_TDUP), alternating, same weights and input.
    python tools/ab_vae_virtual_tdup.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae
from easyanimate_amd import vae_modules

vae = bench_vae.build_vae()
z = (torch.randn(1, 16, 13, 128, 128, device="cuda") / 0.1825).to(torch.bfloat16)
outs = {}
with torch.no_grad():
    for rep in range(4):
        for virt in (True, False):
            vae_modules.VIRTUAL_TDUP = virt
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            t0 = time.perf_counter()
            y = vae.decode(z)[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep == 0:
                outs[virt] = y[:, :, ::7, ::16, ::16].clone()
            print(json.dumps({"virtual_tdup": virt, "rep": rep, "decode_s": round(dt, 4), "MPix_per_s": round(49 * 1024 * 1024 / 1e6 / dt, 2),
                              "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)
            del y
print(json.dumps({"bit_identical_subsample": bool(torch.equal(outs[True], outs[False]))}))
vae_modules.VIRTUAL_TDUP = True
