"""GroupNorm-apply launches of one 49 x 1024^2 VAE decode, by shape: count, total ms (HIP events around the C-ABI call), effective TB/s
(4 B per element).      [EA_LIB_PATH=...] python tools/prof_gn_in_vae.py"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_vae
from easyanimate_amd import _lib

vae = bench_vae.build_vae()
z = (torch.randn(1, 16, 13, 128, 128, device="cuda") / 0.1825).to(torch.bfloat16)
rec = []
orig = _lib.call
def call(name, *a):
    if name != "ea_groupnorm_apply_bf16":
        return orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(name, *a)
    e1.record()
    rec.append((a[5:10], e0, e1))      # T, hw, C, groups, act
    return r
with torch.no_grad():
    vae.decode(z)
    torch.cuda.synchronize()
    _lib.call = call
    vae.decode(z)
    torch.cuda.synchronize()
_lib.call = orig
by = collections.OrderedDict()
for key, e0, e1 in rec:
    d = by.setdefault(tuple(int(k) for k in key), [0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
lib = os.path.basename(os.environ.get("EA_LIB_PATH", "default"))
for (T, hw, C, G, act), (n, ms) in by.items():
    print(json.dumps({"lib": lib, "T": T, "hw": hw, "C": C, "act": act, "launches": n, "total_ms": round(ms, 3), "TB_per_s": round(n * T * hw * C * 4 / ms / 1e9, 2)}))
print(json.dumps({"lib": lib, "all_launches": len(rec), "total_ms": round(sum(v[1] for v in by.values()), 2)}))
