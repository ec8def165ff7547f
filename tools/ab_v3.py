import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from easyanimate_amd import ops, _lib
def timeit(fn, warm=1, iters=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(iters+1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i+1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i+1]) for i in range(iters))[iters//2]
B,H,S=1,48,53504
q=torch.randn(B,H,S,64,device="cuda").to(torch.bfloat16); k=torch.randn(B,H,S,64,device="cuda").to(torch.bfloat16); vt=torch.randn(B,H,64,S,device="cuda").to(torch.bfloat16)
qq=(q.float()*ops.FOLDED_Q_SCALE).to(torch.bfloat16)
out=torch.empty(B,S,H*64,dtype=torch.bfloat16,device="cuda")
res={}
for var in (2,3,2,3):
    _lib.set_option("attn_variant", var)
    ms=timeit(lambda: ops.attention(qq,k,vt,S,ops.FOLDED_ATTN_SCALE,out=out))
    o1=out.clone()
    ms2=timeit(lambda: ops.attention(q,k,vt,S,0.125,out=out))
    res[var]=(o1,out.clone())
    print(json.dumps({"variant":var,"folded_TF":4.0*B*H*S*S*64/ms/1e9,"general_TF":4.0*B*H*S*S*64/ms2/1e9}),flush=True)
for i in (0,1):
    d=(res[2][i].float()-res[3][i].float()).abs()
    print("v2 vs v3", "folded" if i==0 else "general", "max abs diff", d.max().item(), "mean", d.mean().item(), "finite", bool(torch.isfinite(res[3][i].float()).all()))
