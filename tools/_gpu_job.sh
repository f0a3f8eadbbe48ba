#!/bin/bash
set -e
mkdir -p gpurun_out
python - > gpurun_out/ex_auto_gqa.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
import flash_attention_metal_amd as fa
for (B,Hq,Hkv,N,D,c) in ((1,64,8,4096,128,True),(2,32,8,2048,128,True),(1,64,8,2048,64,True),(4,32,8,1536,64,True),(2,64,8,1024,64,False),(1,64,8,8192,64,True)):
    q=torch.randn(B,Hq,N,D,device='cuda',dtype=torch.bfloat16); k=torch.randn(B,Hkv,N,D,device='cuda',dtype=torch.bfloat16); v=torch.randn_like(k)
    o=torch.empty_like(q); lse=torch.empty(B,Hq,N,dtype=torch.float32,device='cuda')
    res={}
    for rnd in range(3):
        for var in ("auto","mfma","mfma16"):
            f=lambda: fa.flash_attention_forward(q,k,v,is_causal=c,variant=var,out=o,lse=lse)
            for _ in range(10): f()
            torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): f()
            b.record(); torch.cuda.synchronize(); res.setdefault(var,[]).append(a.elapsed_time(b)/30*1e3)
    print((B,Hq,Hkv,N,D,c), {k_:round(sorted(v_)[1],1) for k_,v_ in res.items()}, flush=True)
PY
cat gpurun_out/ex_auto_gqa.log
