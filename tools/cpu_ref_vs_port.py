"""TEST / MEASUREMENT INFRASTRUCTURE (build container only: /root/reference does not travel to the GPU box).

bench.py's `cpu_baseline` is `kind: "port"`: oracle/restatement.dit_block timed on the host cores.  This script backs that label
with a measurement (VERDICT r3 next #8a): the UNCHANGED reference block -- easyanimate/models/attention.py:1028-1163
(EasyAnimateDiTBlock, its EasyAnimateAttnProcessor2_0), hosted by oracle/diffusers_shim -- and the port on the SAME weights and
the SAME block sample bench.py uses (one full-width MMDiT block, B = 1, 1024 video + 256 text tokens, fp32, all host cores),
timed alternately, plus their output difference.

    python tools/cpu_ref_vs_port.py > profiles/r04b_cpu_reference_vs_port.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyanimate_amd.synthetic import synth_state_dict  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle import restatement as R  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ns = ref_loader.load()
    d, H, T, N = 3072, 48, 256, 1024
    blk = ns.attention.EasyAnimateDiTBlock(dim=d, num_attention_heads=H, attention_head_dim=64, time_embed_dim=512, norm_eps=1e-5,
                                           is_mmdit_block=True).eval()
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    sd = synth_state_dict(shapes, 0)
    blk.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(0)
    h, e, temb = torch.randn(1, N, d, generator=g), torch.randn(1, T, d, generator=g), torch.randn(1, 512, generator=g)
    rope = ns.shim.get_3d_rotary_pos_embed(64, ((0, 8), (30, 38)), (32, 32), 1, use_real=True)
    rope_p = R.rope_3d(64, ((0, 8), (30, 38)), (32, 32), 1)
    ref = lambda: blk(h, e, temb, image_rotary_emb=rope)
    port = lambda: R.dit_block(sd, "", h, e, temb, rope_p, H, 1e-5)
    with torch.no_grad():
        ho, eo = ref()
        hp, ep = port()
        times = {"reference": [], "port": []}
        for _ in range(4):                      # alternating, so that clock / cache state is shared
            for name, fn in (("reference", ref), ("port", port)):
                t0 = time.perf_counter()
                fn()
                times[name].append(time.perf_counter() - t0)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print(json.dumps({
        "what": "one full-width MMDiT block (d = 3072, 48 heads), B = 1, 1024 video + 256 text tokens, fp32, CPU: the unchanged reference "
                "(easyanimate/models/attention.py:1028-1163 through oracle/diffusers_shim) against oracle/restatement.dit_block "
                "(the cpu_baseline of bench.py, kind 'port') on the same weights and inputs",
        "threads": torch.get_num_threads(), "seconds_reference": times["reference"], "seconds_port": times["port"],
        "median_reference_s": med["reference"], "median_port_s": med["port"], "port_over_reference": med["port"] / med["reference"],
        "max_abs_diff_hidden": (ho - hp).abs().max().item(), "max_abs_diff_text": (eo - ep).abs().max().item(),
        "bit_identical": bool(torch.equal(ho, hp) and torch.equal(eo, ep))}))


if __name__ == "__main__":
    main()
