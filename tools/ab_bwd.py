#!/usr/bin/env python3
"""Interleaved A/B timing of the BACKWARD of several builds of libfa_mi355.so in one process (like tools/ab.py).
usage: ab_bwd.py libA.so libB.so [...] [--shapes c3,nc4k,b16h8] [--rounds 6] [--iters 6]
Each build's forward produces the O / LSE its own backward is fed with; gradients are compared with the first build's."""
import argparse, ctypes, os, time
from ctypes import c_int, c_float, c_longlong, c_void_p
import torch
SHAPES = {"c3": (4, 16, 4096, 64, "bf16", 1), "nc4k": (4, 16, 4096, 64, "bf16", 0), "b16h8": (16, 8, 4096, 64, "bf16", 1),
          "c8k": (4, 16, 8192, 64, "bf16", 1), "c1k": (4, 16, 1024, 64, "bf16", 1), "d128c4k": (1, 32, 4096, 128, "bf16", 1), "d128nc4k": (1, 32, 4096, 128, "bf16", 0), "d128c8k": (2, 16, 8192, 128, "bf16", 1),
          "c3f16": (4, 16, 4096, 64, "f16", 1)}
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--shapes", default="c3,nc4k,b16h8")
ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--iters", type=int, default=6)
a = ap.parse_args()
libs = []
for p in a.libs:
    l = ctypes.CDLL(os.path.abspath(p))
    l.fa_fwd.restype = c_int
    l.fa_fwd.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p]
    l.fa_bwd.restype = c_int
    l.fa_bwd.argtypes = [c_void_p] * 10 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_void_p]
    l.fa_bwd_workspace_bytes.restype = c_longlong; l.fa_bwd_workspace_bytes.argtypes = [c_int] * 3
    libs.append(l)
for name in a.shapes.split(","):
    B, H, N, D, dt, causal = SHAPES[name]
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dt]; fdt = {"f16": 1, "bf16": 2}[dt]
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(tdt) for _ in range(4))
    st = torch.cuda.current_stream().cuda_stream
    state = []
    for l in libs:
        o = torch.empty_like(q); lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
        assert l.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, D, D ** -0.5, H * N * D, N * D, causal, fdt, 0, st) == 0
        gr = [torch.empty(B, H, N, D, dtype=torch.float32, device="cuda") for _ in range(3)]
        ws = torch.empty(l.fa_bwd_workspace_bytes(B, H, N), dtype=torch.uint8, device="cuda")
        state.append((o, lse, gr, ws))
    def launch(i):
        l = libs[i]; o, lse, gr, ws = state[i]
        rc = l.fa_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), gr[0].data_ptr(), gr[1].data_ptr(),
                      gr[2].data_ptr(), ws.data_ptr(), B, H, N, D, D ** -0.5, H * N * D, N * D, causal, fdt, st)
        assert rc == 0, rc
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for i in range(len(libs)): launch(i)
        torch.cuda.synchronize()
    res = [[] for _ in libs]
    for r in range(a.rounds):
        for i in range(len(libs)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): launch(i)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / a.iters)
    fl = 2.5 * (2.0 if causal else 4.0) * B * H * N * N * D
    line = f"{name:7s}"
    for i, p in enumerate(a.libs):
        ms = sorted(res[i]); med = ms[len(ms) // 2]
        diff = "" if i == 0 else " maxreldiff dq/dk/dv " + "/".join(f"{((state[i][2][j]-state[0][2][j]).abs().max()/state[0][2][j].abs().max()).item():.1e}" for j in range(3))
        line += f" | {os.path.basename(p)}: med {med*1e3:8.1f}us {fl/med/1e9:7.1f}TF best {fl/ms[0]/1e9:7.1f}TF{diff}"
    print(line, flush=True)
