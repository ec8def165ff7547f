"""One 49 x 1024^2 VAE decode + encode for rocprofv3 counter runs (the product's default dispatch).   python tools/prof_vae_decode.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae

vae = bench_vae.build_vae()
z = (torch.randn(1, 16, 13, 128, 128, device="cuda") / 0.1825).to(torch.bfloat16)
video = (torch.rand(1, 3, 49, 1024, 1024, device="cuda") * 2 - 1).to(torch.bfloat16)
with torch.no_grad():
    y = vae.decode(z)[0]
    m = vae.encode(video)[0].parameters
torch.cuda.synchronize()
