import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mtamrecommender_amd import hip_ops as ops
D=128
def dev(a): return torch.as_tensor(np.ascontiguousarray(a)).cuda()
for (B,L,wu) in [(1,2,1),(1,2,0),(5,7,1),(128,50,1)]:
    rng=np.random.default_rng(0)
    vi,vc,vp,vu=97,13,L+3,31
    sl=rng.integers(2,L+1,size=B).astype(np.int32)
    ids=[np.zeros((B,L),np.int32) for _ in range(3)]
    for b in range(B):
        for t,v in zip(ids,(vi,vc,vp)): t[b,:sl[b]]=rng.integers(0,v,size=sl[b])
    uid=rng.integers(0,vu,size=B).astype(np.int32)
    R=B*L
    d_ic=dev(rng.standard_normal((R,2*D)).astype(np.float32)); d_pos=dev(rng.standard_normal((R,D)).astype(np.float32))
    ic=dev(rng.standard_normal((R,2*D)).astype(np.float32)); pos=dev(rng.standard_normal((R,D)).astype(np.float32)); usr=dev(rng.standard_normal((B,D)).astype(np.float32))
    g=[torch.zeros((v,D),device="cuda") for v in (vi,vc,vp,vu)]
    n=ops.emb_scatter_partials(B,L); print("case",B,L,wu,"partials",n,flush=True)
    part=torch.zeros(n,device="cuda")
    a=[dev(ids[0]),dev(ids[1]),dev(ids[2]),dev(uid),dev(sl)]
    torch.cuda.synchronize(); print("launch",flush=True)
    ops.emb_scatter_add_bwd(d_ic,d_pos,ic,pos,usr,a[0],a[1],a[2],a[3],a[4],B,L,0.5,wu,g[0],g[1],g[2],g[3],part)
    torch.cuda.synchronize(); print("ok", float(part.sum()), [float(x.abs().sum()) for x in g],flush=True)
