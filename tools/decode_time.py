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
    lib = fa.load_library(); st = torch.cuda.current_stream().cuda_stream
    o0 = torch.empty_like(q); lse0 = torch.empty(B, Hq, Nq, dtype=torch.float32, device="cuda")
    xargs = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o0.data_ptr(), lse0.data_ptr(), B, Hq, Hkv, Nq, Nk, D, D ** -0.5, Hq * Nq * D, Nq * D,
             Hkv * Nk * D, Nk * D, 1, 2, st)
    for _ in range(5): assert lib.fa_fwd_ex(*xargs) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.fa_fwd_ex(*xargs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byts = 2 * B * Hkv * Nk * D * 2
    line = f"B{B} Hq{Hq} Hkv{Hkv} Nq{Nq} Nk{Nk} D{D}: fa_fwd_ex {us:8.1f} us {byts / us / 1e6:.2f} TB/s"
    if (Hq // Hkv) * Nq <= 32:  # fa_fwd_decode (round 4): packed query heads, key splits, workspace + combine
        ws = torch.empty(fa.decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D), dtype=torch.uint8, device="cuda")
        o = torch.empty_like(q); lse = torch.empty(B, Hq, Nq, dtype=torch.float32, device="cuda")
        lib = fa.load_library(); st = torch.cuda.current_stream().cuda_stream  # raw C-ABI calls: the Python wrapper's checks cost more than a short launch
        args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Hq, Hkv, Nq, Nk, D, D ** -0.5, Hq * Nq * D, Nq * D,
                Hkv * Nk * D, Nk * D, 1, 2, ws.data_ptr(), ws.numel(), st)
        for _ in range(5): assert lib.fa_fwd_decode(*args) == 0
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50): lib.fa_fwd_decode(*args)
        e1.record(); torch.cuda.synchronize()
        us2 = e0.elapsed_time(e1) / 50 * 1e3
        line += f" | fa_fwd_decode {us2:8.1f} us {byts / us2 / 1e6:.2f} TB/s ({us / us2:.1f}x)"
        # the same step on an e4m3 KV cache (dtype 3: Q, K, V e4m3, bf16 output): half the bytes
        q8, k8, v8 = (x.to(torch.float8_e4m3fn) for x in (q, k, v))
        a8 = (q8.data_ptr(), k8.data_ptr(), v8.data_ptr()) + args[3:17] + (3,) + args[18:]
        for _ in range(5): assert lib.fa_fwd_decode(*a8) == 0
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50): lib.fa_fwd_decode(*a8)
        e1.record(); torch.cuda.synchronize()
        us3 = e0.elapsed_time(e1) / 50 * 1e3
        line += f" | e4m3 {us3:8.1f} us {byts / 2 / us3 / 1e6:.2f} TB/s ({us2 / us3:.2f}x)"
    print(line + f"   K+V {byts / 1e6:7.1f} MB", flush=True)
