import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flash_attention_metal_amd as fa, oracle
sys.path.insert(0, "tests")
from util import make_qkv, run_op
np.set_printoptions(linewidth=200, precision=4, suppress=True)
D=64
def rep(name, o, ref):
    e = np.abs(o-ref)
    rows = np.where(e.max(-1).reshape(-1) > 2e-3)[0]
    cols = np.where(e.reshape(-1, e.shape[-1]).max(0) > 2e-3)[0]
    print(f"{name}: max err {e.max():.4g}; bad rows {len(rows)} {rows[:24]}; bad cols {len(cols)} {cols[:32]}")
for N in (64, 128):
    q,k,v = make_qkv(oracle, 1,1,N,D,"bf16")
    o64,l64 = oracle.attn_fwd_f64(q,k,v,False)
    o,l = run_op(fa,q,k,v,"bf16",False,"mfma"); rep(f"N={N} random", o, o64); print("  lse err", np.abs(l-l64).max())
    ones = np.ones_like(v)
    o,l = run_op(fa,q,k,ones,"bf16",False,"mfma"); rep(f"N={N} V=1", o, np.ones_like(o))
    z = np.zeros_like(q)
    o,l = run_op(fa,z,k,v,"bf16",False,"mfma"); rep(f"N={N} Q=0 (O=mean V)", o, np.broadcast_to(v.mean(2,keepdims=True), v.shape)); print("  lse", l.reshape(-1)[:4], "expect", np.log(N))
    # V[j,d] = j/64 for all d  -> O[i,:] = sum_j p_ij j/64
    vj = np.broadcast_to((np.arange(N)[:,None]/64.0).astype(np.float32), (N,D)).reshape(1,1,N,D).copy()
    o64,_ = oracle.attn_fwd_f64(q,k,vj,False)
    o,l = run_op(fa,q,k,vj,"bf16",False,"mfma"); rep(f"N={N} V=j/64", o, o64)
    # V[j,d] = d/64 -> O[i,d] = d/64
    vd = np.broadcast_to((np.arange(D)[None,:]/64.0).astype(np.float32), (N,D)).reshape(1,1,N,D).copy()
    o,l = run_op(fa,q,k,vd,"bf16",False,"mfma"); rep(f"N={N} V=d/64", o, vd)
    # one-hot V: V[t, :] = 1 else 0 with Q=0 -> O = 1/N everywhere; per key t check
    bad=[]
    for t in range(N):
        vt = np.zeros_like(v); vt[0,0,t,:]=1
        o,l = run_op(fa,z,k,vt,"bf16",False,"mfma")
        if np.abs(o-1.0/N).max() > 1e-3: bad.append((t, float(o.min()), float(o.max())))
    print(f"N={N} one-hot key probe bad keys: {len(bad)}", bad[:8])
