#!/usr/bin/env python3
"""Run one shape of the operator repeatedly (target for rocprofv3 / quick A-B timing).
usage: run_shape.py B H N D dtype causal iters [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
B, H, N, D = map(int, sys.argv[1:5]); dtype = sys.argv[5]; causal = bool(int(sys.argv[6])); iters = int(sys.argv[7])
variant = sys.argv[8] if len(sys.argv) > 8 else "auto"
tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(tdt) for _ in range(3))
o = torch.empty_like(q); lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
for _ in range(3): fa.flash_attention_forward(q, k, v, is_causal=causal, out=o, lse=lse, variant=variant)
evs = []
for _ in range(iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fa.flash_attention_forward(q, k, v, is_causal=causal, out=o, lse=lse, variant=variant); b.record(); evs.append((a, b))
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)
fl = fa.algorithmic_flops(B, H, N, D, causal)
print(f"B{B} H{H} N{N} D{D} {dtype} causal={int(causal)} {variant}: median {ms[len(ms)//2]:.4f} ms  min {ms[0]:.4f} ms  "
      f"{fl/ms[len(ms)//2]/1e9:.1f} TF median  {fl/ms[0]/1e9:.1f} TF best")
