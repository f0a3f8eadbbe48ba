#!/usr/bin/env python3
"""Time forward + backward for one shape. usage: run_bwd.py B H N dtype causal iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
B, H, N = map(int, sys.argv[1:4]); dtype = sys.argv[4]; causal = bool(int(sys.argv[5])); iters = int(sys.argv[6]); D = 64
tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(tdt) for _ in range(4))
o, lse = fa.flash_attention_forward(q, k, v, is_causal=causal)
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:  # warm up BY TIME: two launches end long before the clocks have settled (the first
    for _ in range(5): fa.flash_attention_backward(q, k, v, o, do, lse, is_causal=causal)  # version of this tool read 10 % low)
    torch.cuda.synchronize()
evs = []
for _ in range(iters):  # events around blocks of 4 back-to-back calls
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): fa.flash_attention_backward(q, k, v, o, do, lse, is_causal=causal)
    b.record(); evs.append((a, b))
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) / 4 for a, b in evs); fl = 2.5 * fa.algorithmic_flops(B, H, N, D, causal)
print(f"bwd B{B} H{H} N{N} D{D} {dtype} causal={int(causal)}: median {ms[len(ms)//2]:.4f} ms  {fl/ms[len(ms)//2]/1e9:.1f} TF (algorithmic 5 products)  best {fl/ms[0]/1e9:.1f} TF")
