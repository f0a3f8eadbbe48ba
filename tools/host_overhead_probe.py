import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import flash_attention_metal_amd as fa
B,H,N,D=1,64,4096,64
dev=torch.device("cuda",0)
g=torch.Generator(device=dev).manual_seed(0)
q,k,v=((torch.rand(B,H,N,D,generator=g,device=dev)*2-1).to(torch.bfloat16) for _ in range(3))
o=torch.empty_like(q); lse=torch.empty(B,H,N,dtype=torch.float32,device=dev)
plan=fa.ForwardPlan(q,k,v,is_causal=True,out=o,lse=lse)
t0=time.perf_counter()
while time.perf_counter()-t0<0.4:
    for _ in range(20): plan.launch()
    torch.cuda.synchronize()
for trial in range(3):
    torch.cuda.synchronize(); torch.cuda.synchronize()
    e=[torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0=time.perf_counter()
    e[0].record()
    t1=time.perf_counter()
    for _ in range(20): plan.launch()
    t2=time.perf_counter()
    e[1].record()
    t3=time.perf_counter()
    torch.cuda.synchronize()
    t4=time.perf_counter()
    torch.cuda.synchronize()
    t5=time.perf_counter()
    print(f"wall {1e3*(t5-t0):.3f} ms | ev {e[0].elapsed_time(e[1]):.3f} ms | first record {1e6*(t1-t0):.0f} us, 20 launches (host) {1e6*(t2-t1):.0f} us, rec {1e6*(t3-t2):.0f}, sync1 {1e6*(t4-t3):.0f}, sync2 {1e6*(t5-t4):.0f}")
