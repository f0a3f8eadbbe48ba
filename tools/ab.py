#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libfa_mi355.so in ONE process (cdna guide rule 24).
usage: ab.py libA.so[:variant] libB.so[:variant] [...] [--shapes c3,nc8k,c16k] [--rounds 8] [--iters 20]
`lib.so:5` times variant 5 (mfma_pp) of that build; without a suffix --variant applies (0 = auto)."""
import argparse, ctypes, os, sys
from ctypes import c_int, c_float, c_longlong, c_void_p
import torch

SHAPES = {  # name: (B, H, N, D, dtype, causal)
    "c3": (4, 16, 4096, 64, "bf16", 1), "nc4k": (4, 16, 4096, 64, "bf16", 0), "nc8k": (1, 64, 8192, 64, "bf16", 0),
    "c16k": (1, 64, 16384, 64, "bf16", 1), "c2": (1, 8, 1024, 64, "f16", 0), "c4": (1, 32, 16384, 128, "bf16", 1),
    "c5": (4, 16, 8192, 64, "fp8", 1), "c5bf": (4, 16, 8192, 64, "bf16", 1), "c5d128": (2, 16, 8192, 128, "fp8", 1), "c5d128bf": (2, 16, 8192, 128, "bf16", 1), "c8k": (4, 16, 8192, 64, "bf16", 1), "c1k": (4, 16, 1024, 64, "bf16", 1), "c2bf": (1, 8, 1024, 64, "bf16", 0), "c2c": (1, 8, 1024, 64, "bf16", 1), "h8n2k": (1, 8, 2048, 64, "bf16", 1), "h16n512": (1, 16, 512, 64, "bf16", 1), "h32n1k": (1, 32, 1024, 64, "bf16", 1), "c512": (4, 16, 512, 64, "bf16", 1), "c256": (4, 16, 256, 64, "bf16", 1), "c128": (4, 16, 128, 64, "bf16", 1), "c2k": (4, 16, 2048, 64, "bf16", 1), "d128nc": (1, 32, 8192, 128, "bf16", 0),
    "h32n8k": (1, 32, 8192, 64, "bf16", 1), "h16n16k": (1, 16, 16384, 64, "bf16", 1), "h8n16k": (1, 8, 16384, 64, "bf16", 1), "h16n8k": (1, 16, 8192, 64, "bf16", 1),
    "nc2k": (4, 16, 2048, 64, "bf16", 0), "c3f16": (4, 16, 4096, 64, "f16", 1), "h32nc4k": (1, 32, 4096, 64, "bf16", 0), "c3k": (4, 16, 3072, 64, "bf16", 1),
    "b5n4k": (5, 16, 4096, 64, "bf16", 1), "b6n4k": (6, 16, 4096, 64, "bf16", 1), "b7n4k": (7, 16, 4096, 64, "bf16", 1), "c6k": (4, 16, 6144, 64, "bf16", 1), "c5k": (4, 16, 5120, 64, "bf16", 1),
    "b2nc2k": (2, 16, 2048, 64, "bf16", 0), "nc1k": (4, 16, 1024, 64, "bf16", 0), "b8nc1k": (8, 16, 1024, 64, "bf16", 0), "b8nc2k": (8, 16, 2048, 64, "bf16", 0), "b3n8k": (3, 16, 8192, 64, "bf16", 1),
    "b16nc1k": (16, 16, 1024, 64, "bf16", 0), "b16nc512": (16, 16, 512, 64, "bf16", 0), "b32nc512": (32, 16, 512, 64, "bf16", 0), "b8nc1536": (8, 16, 1536, 64, "bf16", 0),
    "b16c1k": (16, 16, 1024, 64, "bf16", 1), "b8c1k": (8, 16, 1024, 64, "bf16", 1), "b8c1536": (8, 16, 1536, 64, "bf16", 1), "b32c512": (32, 16, 512, 64, "bf16", 1), "b16c1kf16": (16, 16, 1024, 64, "f16", 1),
    "c3x4": (16, 16, 4096, 64, "bf16", 1), "c3x3": (12, 16, 4096, 64, "bf16", 1), "c3x2": (8, 16, 4096, 64, "bf16", 1), "c3h": (2, 16, 4096, 64, "bf16", 1), "c3h48": (3, 16, 4096, 64, "bf16", 1),
    "d128c1k": (2, 32, 1024, 128, "bf16", 1), "d128c2k": (2, 32, 2048, 128, "bf16", 1), "d128c4k": (1, 32, 4096, 128, "bf16", 1),
    "d128h8n1k": (1, 8, 1024, 128, "bf16", 0), "d128h8n1kc": (1, 8, 1024, 128, "bf16", 1), "d128h8n2kc": (1, 8, 2048, 128, "bf16", 1), "d128h16n512c": (1, 16, 512, 128, "bf16", 1), "d128h32n256": (1, 32, 256, 128, "bf16", 0), "d128h8n4kc": (1, 8, 4096, 128, "bf16", 1),
    "d128c8k": (1, 32, 8192, 128, "bf16", 1), "d128c4k8h": (1, 8, 4096, 128, "bf16", 1), "d128nc2k": (2, 32, 2048, 128, "bf16", 0),
}
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--shapes", default="c3,nc8k,c16k")
ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--dump", action="store_true", help="print every round's time per build")
ap.add_argument("--warm-ms", type=float, default=0.0, help="alternate the builds for this long before timing (the clocks take ~10 rounds to settle)")
a = ap.parse_args()
libs = []
variants = []
for p in a.libs:
    p, _, vs = p.partition(":")
    variants.append(int(vs) if vs else a.variant)
    l = ctypes.CDLL(os.path.abspath(p))
    l.fa_fwd.restype = c_int
    l.fa_fwd.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p]
    libs.append(l)
for name in a.shapes.split(","):
    B, H, N, D, dt, causal = SHAPES[name]
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "fp8": torch.float8_e4m3fn}[dt]; fdt = {"f16": 1, "bf16": 2, "fp8": 3}[dt]
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(tdt) for _ in range(3))
    o = torch.empty_like(q, dtype=torch.bfloat16 if dt == "fp8" else tdt); lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def launch(l):
        rc = l.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, D, D ** -0.5,
                      H * N * D, N * D, causal, fdt, variants[libs.index(l)], st)
        assert rc == 0, rc
    outs = []
    for l in libs:
        for _ in range(3): launch(l)
        torch.cuda.synchronize(); outs.append(o.clone())
    if a.warm_ms > 0:
        import time
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < a.warm_ms:
            for l in libs:
                for _ in range(a.iters): launch(l)
            torch.cuda.synchronize()
    res = [[] for _ in libs]
    for r in range(a.rounds):
        for i, l in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): launch(l)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / a.iters)
    fl = (2.0 if causal else 4.0) * B * H * N * N * D
    line = f"{name:7s}"
    for i, p in enumerate(a.libs):
        ms = sorted(res[i]); med = ms[len(ms) // 2]
        same = "" if i == 0 else (" same-bits" if torch.equal(outs[i], outs[0]) else f" maxdiff={(outs[i].float()-outs[0].float()).abs().max().item():.2e}")
        line += f" | {os.path.basename(p.partition(':')[0])}:v{variants[i]}: med {med*1e3:8.1f}us {fl/med/1e9:7.1f}TF best {fl/ms[0]/1e9:7.1f}TF{same}"
    print(line, flush=True)
    if a.dump:
        for i, p in enumerate(a.libs):
            print("        ", os.path.basename(p.partition(':')[0]), " ".join(f"{x*1e3:.1f}" for x in res[i]), flush=True)
