#!/usr/bin/env python3
"""Does FA_VARIANT_AUTO pick (close to) the fastest kernel? Times auto and every matrix-core variant that supports the shape,
interleaved, and flags shapes where auto is more than --tol slower than the best one. usage: auto_check.py [--tol 0.07]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
ap = argparse.ArgumentParser(); ap.add_argument("--tol", type=float, default=0.07); ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
VARS = ["auto", "mfma", "mfma16", "mfma_splitkv", "mfma_split2", "mfma_h64s2", "mfma_fp8pv"]
shapes = []
for D in (64, 128):
    for dt in ("bf16", "fp8"):
        for causal in (1, 0):
            for (B, H, N) in ((1, 8, 256), (1, 8, 1024), (1, 8, 4096), (1, 32, 512), (1, 32, 2048), (1, 64, 1024), (2, 64, 512), (4, 16, 2048),
                              (1, 128, 1024), (4, 16, 4096), (1, 16, 16384), (8, 32, 256), (1, 256, 512)):
                shapes.append((B, H, N, D, dt, causal))
tdt = {"bf16": torch.bfloat16, "fp8": getattr(torch, "float8_e4m3fn", None)}
lib = fa.load_library()
bad = 0
for (B, H, N, D, dt, causal) in shapes:
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16).to(tdt[dt]) for _ in range(3))
    vs = [x for x in VARS if x == "auto" or fa.supported({"fp8": "fp8_e4m3"}.get(dt, dt), x, D)]
    plans = {x: fa.ForwardPlan(q, k, v, is_causal=bool(causal), variant=x) for x in vs}
    iters = max(3, min(50, int(2e-3 / (4.0 * B * H * N * N * D / (2 if causal else 1) / 600e12 + 6e-6))))
    for p in plans.values():
        for _ in range(3): p.launch()
    res = {x: [] for x in vs}
    for _ in range(a.rounds):
        for x in vs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): plans[x].launch()
            e1.record(); torch.cuda.synchronize()
            res[x].append(e0.elapsed_time(e1) / iters * 1e3)
    med = {x: sorted(r)[len(r) // 2] for x, r in res.items()}
    best = min((x for x in vs if x != "auto"), key=lambda x: med[x])
    chosen = lib.fa_variant_name(lib.fa_resolve_variant_for(fa.DTYPES[{"fp8": "fp8_e4m3"}.get(dt, dt)], D, B, H, N, causal)).decode()
    flag = "  <-- auto slower" if med["auto"] > med[best] * (1 + a.tol) else ""
    bad += bool(flag)
    print(f"B{B} H{H} N{N} D{D} {dt} c{causal}: auto={chosen} {med['auto']:.1f}us  best={best} {med[best]:.1f}us  " +
          " ".join(f"{x[5:] or x}={med[x]:.1f}" for x in vs if x != "auto") + flag, flush=True)
print(f"{bad} of {len(shapes)} shapes where auto is more than {a.tol:.0%} behind the best kernel")
