"""LayerNorm-modulate at the config-3 video stream (2 x 53 248 x 3072) and text stream, in-process A/B of the round-1 kernel
(ea_set_option("ln_wgs", 0): 32 rows per workgroup) against the sweeping kernel over grid sizes and streaming flags, alternating;
GB/s = (read + write of the stream) / time.      python tools/ab_layernorm.py [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from microbench_vae_common import timeit

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D = 3072
for what, B, R in (("video stream c3", 2, 53248), ("one rank of 8 (cfg2 x sp4)", 1, 13312), ("video stream c2", 2, 13312), ("text stream", 2, 256)):
    x = torch.randn(B, R, D, device="cuda").to(torch.bfloat16)
    gamma, beta = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    table = torch.randn(B, 6 * D, device="cuda")
    y = torch.empty_like(x)
    fn = lambda: ops.layernorm_modulate(x, gamma, beta, table[:, D:2 * D], table[:, :D], 1e-5, out=y)
    _lib.set_option("ln_wgs", 0)
    fn(); torch.cuda.synchronize()
    ref = y.clone()
    variants = [(0, 3)] + [(w, nt) for w in (512, 640, 704, 768, 832, 896, 1024, 1536) for nt in (2, 3)] + [(768, 0), (768, 1)]
    for rep in range(reps):
        for wgs, nt in variants:
            _lib.set_option("ln_wgs", wgs)
            _lib.set_option("ln_nt", nt)
            ms = timeit(fn, warm=2, iters=9)
            same = bool(torch.equal(y, ref))
            print(json.dumps({"what": what, "ln_wgs": wgs, "ln_nt": nt, "us": round(ms * 1e3, 1), "GB_s": round(4.0 * B * R * D / ms / 1e6, 1),
                              "bit_identical": same}), flush=True)
    _lib.set_option("ln_wgs", 768); _lib.set_option("ln_nt", 2)
