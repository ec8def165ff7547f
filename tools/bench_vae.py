"""VAE-only benchmark (BASELINE.json config 4): AutoencoderKLMagvit encode + decode of 49 x 1024 x 1024 RGB on one
MI355X, bf16, random-init full-width weights.  Prints one JSON line with MPix/s (MPix = F*H*W/1e6 of the pixel
video) for decode and encode, the algorithmic FLOPs (decode 7.50e14, encode 4.77e14 at this shape) and
the implied MFMA fraction.

    python tools/bench_vae.py [--frames 49 --size 1024 --iters 2]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PEAK = 2500.0


def conv_flops(vae, frames, size):
    """2*k^3*Cin*Cout per output voxel for every conv + linears/attention of the mid blocks (SURVEY 8d formula)."""
    import torch.nn as nn
    from easyanimate_amd import ops
    from easyanimate_amd.vae_modules import SpatialAttention
    from easyanimate_amd import vae_modules as VM
    total = {"enc": 0.0, "dec": 0.0}
    executed = {"enc": 0.0, "dec": 0.0}   # MFMA work actually issued: the sub-pixel up-samplers run 12 of 27 taps, a layer behind a
    virt = [False]                        # virtual temporal x2 18 of 27 (vae_modules.SUBPIXEL_UPSAMPLE / TEMPORAL_TAP_MERGE)
    traffic = {"enc": 0.0, "dec": 0.0}   # minimal bf16 activation traffic: every conv / GroupNorm reads its input once and
    side = ["enc"]                        # writes its output once, no fusion credit
    # walk the graph symbolically with shapes only
    def conv(c, T, H, W, ups=False, tdup=False):
        co, ci, k = c.weight.shape[0], c.weight.shape[1], c.weight.shape[2]
        st, ss = c.stride[0], c.stride[1]
        pad = c.padding[1] if k == 3 else 0
        To, Ho, Wo = ops.conv_out_shape(T, H, W, k, st, ss, pad, ups)
        fl = 2.0 * k ** 3 * ci * co * To * Ho * Wo
        ex = fl
        if ups and k == 3 and VM.SUBPIXEL_UPSAMPLE and W % 256 == 0 and ci % 64 == 0 and co % 256 == 0:
            ex = fl * 12.0 / 27.0
        elif virt[0] and k == 3 and VM.TEMPORAL_TAP_MERGE and To > 1 and ops.conv3d_tmerge_ok(To, H, W, ci, ops.round_up(co, 8)):
            ex = fl * 18.0 / 27.0
        if k == 3:
            virt[0] = bool(tdup and To > 1 and VM.VIRTUAL_TDUP)     # a temporal up-sampler leaves a virtual clip for the next 3x3x3
        executed[side[0]] += ex
        if tdup and To > 1:
            To = 2 * To - 1
        traffic[side[0]] += 2.0 * (T * H * W * ci + To * Ho * Wo * co)
        return fl, (To, Ho, Wo)
    def gn(shp, c):
        traffic[side[0]] += 4.0 * shp[0] * shp[1] * shp[2] * c
    def res(r, shp):
        fl = 0.0
        if not isinstance(r.shortcut, nn.Identity):
            f, _ = conv(r.shortcut, *shp); fl += f
        gn(shp, r.conv1.weight.shape[1])
        f, s1 = conv(r.conv1, *shp); fl += f
        gn(s1, r.conv2.weight.shape[1])
        f, s2 = conv(r.conv2, *s1); fl += f
        return fl, s2
    def mid(m, shp):
        fl, s = res(m.convs[0], shp)
        for a, r in zip(m.attentions, m.convs[1:]):
            if a is not None:
                T, H, W = s
                n, C = H * W, a.inner_dim
                fl += T * (8.0 * n * C * C + 4.0 * n * n * C)
                executed[side[0]] += T * (8.0 * n * C * C + 4.0 * n * n * C)
            f, s = res(r, s); fl += f
        return fl, s
    T, H, W = frames, size, size
    f, s = conv(vae.encoder.conv_in, T, H, W); total["enc"] += f
    for b in vae.encoder.down_blocks:
        for r in b.convs:
            f, s = res(r, s); total["enc"] += f
        if b.downsampler is not None:
            f, s = conv(b.downsampler.conv, *s); total["enc"] += f
    f, s = mid(vae.encoder.mid_block, s); total["enc"] += f
    gn(s, vae.encoder.conv_out.weight.shape[1])
    f, s = conv(vae.encoder.conv_out, *s); total["enc"] += f
    f, _ = conv(vae.quant_conv, *s); total["enc"] += f
    lat = s
    side[0] = "dec"
    f, s = conv(vae.post_quant_conv, *lat); total["dec"] += f
    f, s = conv(vae.decoder.conv_in, *s); total["dec"] += f
    f, s = mid(vae.decoder.mid_block, s); total["dec"] += f
    for b in vae.decoder.up_blocks:
        for r in b.convs:
            f, s = res(r, s); total["dec"] += f
        if b.upsampler is not None:
            f, s = conv(b.upsampler.conv, *s, ups=True, tdup=hasattr(b.upsampler, "padding_flag")); total["dec"] += f
    gn(s, vae.decoder.conv_out.weight.shape[1])
    f, s = conv(vae.decoder.conv_out, *s); total["dec"] += f
    total["traffic"] = traffic
    total["executed"] = executed
    return total, lat, s


FULL = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512],
            down_block_types=("SpatialDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D", "SpatialTemporalDownBlock3D"),
            up_block_types=("SpatialUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D", "SpatialTemporalUpBlock3D"),
            mid_block_attention_type="spatial", latent_channels=16, norm_num_groups=32, spatial_group_norm=True,
            cache_mag_vae=True, slice_mag_vae=False, mini_batch_encoder=4, mini_batch_decoder=1)


def build_vae(small: bool = False, device="cuda"):
    """Random-init full-width V5/V5.1 VAE, bf16, on the device."""
    from easyanimate_amd import AutoencoderKLMagvit
    cfg = dict(FULL)
    if small:
        cfg.update(block_out_channels=[64, 64, 128, 128], norm_num_groups=16)
    with torch.device("meta"):
        vae = AutoencoderKLMagvit(**cfg)
    vae = vae.to(torch.bfloat16).to_empty(device=device)
    torch.cuda.manual_seed(2)
    with torch.no_grad():
        for n, p in vae.named_parameters():
            if p.dim() >= 2:
                b = 1.0 / (p[0].numel() ** 0.5)
                p.uniform_(-b, b)
            elif "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.uniform_(-0.02, 0.02)
    return vae.eval()


def run(vae, frames: int = 49, size: int = 1024, iters: int = 2, kernel_breakdown: bool = True):
    """Times decode of randn[1,16,F',h,w]/scaling_factor and encode of U(-1,1)[1,3,F,H,W]: `iters`
    timed passes each after one warm-up, then (kernel_breakdown) one more pass with HIP events around every convolution
    launch, labelled with the kernel variant that served it (ea_last_dispatch)."""
    from easyanimate_amd import ops
    fl, lat, out = conv_flops(vae, frames, size)
    mpix = frames * size * size / 1e6
    z = torch.randn(1, 16, *lat, device="cuda").to(torch.bfloat16) / 0.1825
    video = (torch.rand(1, 3, frames, size, size, device="cuda") * 2 - 1).to(torch.bfloat16)
    res = {}
    with torch.no_grad():
        for name, fn in (("decode", lambda: vae.decode(z)[0]), ("encode", lambda: vae.encode(video)[0].mode())):
            y = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                y = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            key = "dec" if name == "decode" else "enc"
            res[name] = {"seconds": dt, "MPix_per_s": mpix / dt, "algorithmic_flop": fl[key],
                         "TFLOPs": fl[key] / dt / 1e12, "mfma_frac": fl[key] / dt / 1e12 / PEAK,
                         "executed_flop": fl["executed"][key], "executed_mfma_frac": fl["executed"][key] / dt / 1e12 / PEAK,
                         "min_activation_bytes": fl["traffic"][key], "hbm_frac": fl["traffic"][key] / dt / 8e12,
                         "finite": bool(torch.isfinite(y.float()).all().item()), "out_shape": list(y.shape)}
            del y
            if kernel_breakdown:
                with ops.KernelTimer("conv3d") as kt:
                    y = fn()
                torch.cuda.synchronize()
                del y
                by = {}
                for lab, ms in zip(kt.labels, kt.durations_ms()):
                    e = by.setdefault(lab, [0, 0.0])
                    e[0] += 1
                    e[1] += ms
                dom = max(by.items(), key=lambda kv: kv[1][1])
                res[name]["conv_kernels_ms"] = {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}
                res[name]["dominant_kernel"] = dom[0]
                res[name]["dominant_avg_launch_ms"] = dom[1][1] / dom[1][0]
                res[name]["dominant_share_of_pass"] = dom[1][1] * 1e-3 / dt
    return {"frames": frames, "size": size, "MPix": mpix, "latent_shape": list(lat),
            "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, **res}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=49)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--small", action="store_true", help="reduced widths (debug)")
    a = ap.parse_args()
    vae = build_vae(a.small)
    print(json.dumps({"metric": "VAE MPix/s (AutoencoderKLMagvit, bf16, 1 GPU)", **run(vae, a.frames, a.size, a.iters)}))


if __name__ == "__main__":
    main()
