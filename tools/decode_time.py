#!/usr/bin/env python3
"""Decode-step shapes (Nq = 1..16 against a long key cache, grouped-query heads) through fa_fwd_ex: time per launch and
the K+V bytes it streams. usage: decode_time.py [B Hq Hkv Nq Nk D]... (default: a sweep)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
a = [int(x) for x in sys.argv[1:]]
shapes = [tuple(a[i:i + 6]) for i in range(0, len(a), 6)] or \
    [(1, 32, 8, 1, nk, d) for d in (64, 128) for nk in (256, 512, 1024, 2048, 4096, 8192, 16384)] + [(1, 32, 8, 16, 2048, 128), (1, 64, 8, 1, 4096, 64), (2, 32, 8, 1, 4096, 128)]
for (B, Hq, Hkv, Nq, Nk, D) in shapes:
    q = torch.randn(B, Hq, Nq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Hkv, Nk, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    for _ in range(5): fa.flash_attention_forward(q, k, v, is_causal=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fa.flash_attention_forward(q, k, v, is_causal=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byts = 2 * B * Hkv * Nk * D * 2
    print(f"B{B} Hq{Hq} Hkv{Hkv} Nq{Nq} Nk{Nk} D{D}: {us:8.1f} us   K+V {byts / 1e6:7.1f} MB  {byts / us / 1e6:.2f} TB/s", flush=True)
