import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
import flash_attention_metal_amd as fa, oracle
from util import make_qkv, to_dev
def run(q,k,v,do,dt,causal):
    qd,kd,vd,dod=(to_dev(x,dt) for x in (q,k,v,do))
    o,lse=fa.flash_attention_forward(qd,kd,vd,is_causal=causal)
    g=fa.flash_attention_backward(qd,kd,vd,o,dod,lse,is_causal=causal); torch.cuda.synchronize()
    return [x.cpu().numpy() for x in g]
N=64
q,k,v=make_qkv(oracle,1,1,N,64,"f16")
q=oracle.round_to(q*np.linspace(0.3,3.0,N).astype(np.float32)[None,None,:,None],"f16")
s=np.einsum("id,jd->ij",q[0,0].astype(np.float64),k[0,0].astype(np.float64))*0.125
P=np.exp(s-s.max(1,keepdims=True)); P/=P.sum(1,keepdims=True)
res=[]
for i0 in list(range(0,16))+[20,31,32,33,40,63]:
    do=np.zeros_like(q); do[0,0,i0,:]=1.0
    g=run(q,k,v,do,"f16",False)
    col=g[2][0,0,:,0]            # = P[i?, :] as the kernel sees it
    best=int(np.argmin([np.abs(col-P[i]).max() for i in range(N)]))
    # also try: kernel used S row i0 but lse of row i1: col = exp(s[i0]-lse[i1]) -> ratio constant
    ratio=col/np.maximum(P[i0],1e-30)
    res.append((i0,best,float(np.abs(col-P[i0]).max()),float(ratio.min()),float(ratio.max())))
for r in res: print("i0=%2d best-matching P row=%2d  err-vs-own=%.4f  ratio col/P[i0] in [%.4f, %.4f]"%r)
