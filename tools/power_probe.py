#!/usr/bin/env python3
"""Diagnosis only (never a reported number): the same launch on random, small-amplitude and all-zero inputs. Cycles per
launch do not depend on the data; if the zero run is much faster the chip is holding its clock down under load
(MI355X_MICROARCH.md, DVFS give-back) and further cycle savings return only partly as throughput."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import flash_attention_metal_amd as fa
shapes = {"c3": (4, 16, 4096, 64), "c4": (1, 32, 16384, 128), "c16k": (1, 64, 16384, 64)}
for name in (sys.argv[1:] or ["c3", "c4"]):
    B, H, N, D = shapes[name]
    g = torch.Generator(device="cuda").manual_seed(0)
    for kind in ("random", "amp1e-3", "zeros", "random"):
        mk = lambda: (torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1)
        scale = {"random": 1.0, "amp1e-3": 1e-3, "zeros": 0.0}[kind]
        q, k, v = ((mk() * scale).to(torch.bfloat16) for _ in range(3))
        plan = fa.ForwardPlan(q, k, v, is_causal=True)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            for _ in range(10): plan.launch()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if N <= 4096 else 10
        a.record()
        for _ in range(n): plan.launch()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        print(f"{name} {kind:8s}: {ms*1e3:8.1f} us  {fa.algorithmic_flops(B,H,N,D,True)/ms/1e9:7.1f} TF", flush=True)
