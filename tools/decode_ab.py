#!/usr/bin/env python3
"""fa_fwd_decode of several builds, interleaved in one process (raw C-ABI calls). usage: [DECODE_AB_FP8=1] decode_ab.py lib1.so lib2.so ...
(DECODE_AB_FP8=1: e4m3 Q / K / V, dtype 3)"""
import ctypes, os, sys
from ctypes import c_int, c_float, c_longlong, c_void_p
import torch
libs = []
for p in sys.argv[1:]:
    l = ctypes.CDLL(os.path.abspath(p))
    l.fa_fwd_decode.restype = c_int
    l.fa_fwd_decode.argtypes = [c_void_p] * 5 + [c_int] * 6 + [c_float] + [c_longlong] * 4 + [c_int, c_int, c_void_p, c_longlong, c_void_p]
    l.fa_fwd_decode_workspace_bytes.restype = c_longlong; l.fa_fwd_decode_workspace_bytes.argtypes = [c_int] * 6
    libs.append(l)
shapes = [(1, 32, 8, 1, 4096, 64), (1, 32, 8, 1, 16384, 64), (1, 32, 32, 1, 16384, 64), (1, 32, 8, 1, 16384, 128), (1, 32, 32, 1, 16384, 128),
          (8, 32, 8, 1, 4096, 128), (1, 32, 8, 4, 8192, 64), (1, 8, 1, 1, 32768, 128)]
for (B, Hq, Hkv, Nq, Nk, D) in shapes:
    q = torch.randn(B, Hq, Nq, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, Hkv, Nk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
    o = torch.empty_like(q); lse = torch.empty(B, Hq, Nq, dtype=torch.float32, device="cuda")
    FP8 = bool(os.environ.get("DECODE_AB_FP8"))
    if FP8: q, k, v = (x.to(torch.float8_e4m3fn) for x in (q, k, v))
    st = torch.cuda.current_stream().cuda_stream
    byts = 2 * B * Hkv * Nk * D * (1 if FP8 else 2)
    line = f"B{B} Hq{Hq} Hkv{Hkv} Nq{Nq} Nk{Nk} D{D} ({byts/1e6:.0f} MB):"
    res = [[] for _ in libs]
    wss = [torch.empty(max(16, l.fa_fwd_decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D)), dtype=torch.uint8, device="cuda") for l in libs]
    for rnd in range(6):
        for i, l in enumerate(libs):
            args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Hq, Hkv, Nq, Nk, D, D ** -0.5, Hq * Nq * D, Nq * D,
                    Hkv * Nk * D, Nk * D, 1, 3 if FP8 else 2, wss[i].data_ptr(), wss[i].numel(), st)
            for _ in range(3): assert l.fa_fwd_decode(*args) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): l.fa_fwd_decode(*args)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / 30 * 1e3)
    for i, p in enumerate(sys.argv[1:]):
        us = sorted(res[i])[len(res[i]) // 2]
        line += f" | {os.path.basename(p)[4:-3]} {us:6.1f}us {byts/us/1e6:.2f}TB/s"
    print(line, flush=True)
