"""Attention timing at the config-3 shape (1 x 48 heads x 53 504 tokens, scale folded into Q); EA_LIB_PATH selects the library
build for an A/B (one library per process; alternate the processes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops
from microbench_vae_common import timeit

lib = os.path.basename(os.environ.get("EA_LIB_PATH", "default"))
B, H, S = 1, 48, 53504
q = torch.randn(B, H, S, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, S, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, S, device="cuda").to(torch.bfloat16)
qq = (q.float() * ops.FOLDED_Q_SCALE).to(torch.bfloat16)
out = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device="cuda")
variants = [3] + ([2] if _lib.get_option("build_variants") == 1 else [])    # 2: the 32x32x16 generation (EA_BUILD_VARIANTS=1 libraries)
# EA_AB_NW=1: alternate the workgroup shapes of v3 -- (waves, LDS stages) = (4, 2) product, (8, 2) one stream per CU, (4, 3) deeper ring
nws = ([(4, 2), (16, 2)] if os.environ.get("EA_AB_NW") == "pp" else [(4, 2), (8, 2), (4, 3)]) if os.environ.get("EA_AB_NW") else [(_lib.get_option("attn_nw"), _lib.get_option("attn_stages"))]
ref = None
for rep in range(3):
  for variant in variants:
    for nw, stages in (nws if variant == 3 else [(4, 2)]):
        _lib.set_option("attn_variant", variant)
        _lib.set_option("attn_nw", nw)
        _lib.set_option("attn_stages", stages)
        ms = timeit(lambda: ops.attention(qq, k, vt, S, ops.FOLDED_ATTN_SCALE, out=out), warm=1, iters=5)
        same = None
        if variant == 3:
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref)) if nw != 16 else round((out.float() - ref.float()).abs().max().item(), 5)
        print(json.dumps({"lib": lib, "kernel": f"attention v{variant}", "nw": nw, "stages": stages, "ms": round(ms, 3), "TFLOPs": round(4.0 * B * H * S * S * 64 / ms / 1e9, 1),
                          "bit_identical_to_first": same}), flush=True)
