#!/usr/bin/env python3
"""Diagnosis only: the backward on random and all-zero inputs (same instruction stream; see tools/power_probe.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
B, H, N, D = 4, 16, 4096, 64
g = torch.Generator(device="cuda").manual_seed(0)
for kind in ("random", "zeros", "random"):
    sc = 1.0 if kind == "random" else 0.0
    q, k, v, do = (((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1) * sc).to(torch.bfloat16) for _ in range(4))
    o, lse = fa.flash_attention_forward(q, k, v, is_causal=True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(5): fa.flash_attention_backward(q, k, v, o, do, lse, is_causal=True)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fa.flash_attention_backward(q, k, v, o, do, lse, is_causal=True)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"bwd config-3 shape {kind:7s}: {ms*1e3:7.1f} us  {2.5*fa.algorithmic_flops(B,H,N,D,True)/ms/1e9:6.1f} TF", flush=True)
