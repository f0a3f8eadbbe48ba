#!/usr/bin/env python3
"""Accuracy of several builds of libfa_mi355.so at FULL benchmark shapes, on the GPU, against an fp64 statement of the
operator evaluated with torch on the same device (a development aid: the parity tests proper compare with oracle/ at
sizes the CPU oracle finishes in seconds; this table exists to price numerics changes such as the pre-scaled operand).
usage: accuracy.py libA.so[:variant] [libB.so[:variant] ...] [--shapes c3,c4,c5] [--amps 1,3] [--heads 4] [--bwd]
Prints per (shape, amp, lib): max|O - ref|, rms, max|LSE - ref|, and with --bwd the gradient errors of that build's
fa_bwd fed with that build's own forward O / LSE (relative to max|grad|)."""
import argparse, ctypes, os
from ctypes import c_int, c_float, c_longlong, c_void_p
import torch

SHAPES = {  # name: (B, H, N, D, dtype, causal) -- heads are cut to --heads (the oracle is O(N^2) per head in fp64)
    "c3": (4, 16, 4096, 64, "bf16", 1), "c4": (1, 32, 16384, 128, "bf16", 1), "c5": (4, 16, 8192, 64, "fp8", 1),
    "c2": (1, 8, 1024, 64, "f16", 0), "nc4k": (4, 16, 4096, 64, "bf16", 0), "c3f16": (4, 16, 4096, 64, "f16", 1),
    "c1k": (4, 16, 1024, 64, "bf16", 1), "d128c4k": (1, 32, 4096, 128, "bf16", 1), "d32": (1, 8, 2048, 32, "bf16", 1),
    "d96": (1, 8, 2048, 96, "bf16", 1),
}
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--shapes", default="c3,c4,c5")
ap.add_argument("--amps", default="1,3"); ap.add_argument("--heads", type=int, default=4); ap.add_argument("--bwd", action="store_true")
ap.add_argument("--variant", type=int, default=0)
a = ap.parse_args()
libs, variants = [], []
for p in a.libs:
    p, _, vs = p.partition(":")
    variants.append(int(vs) if vs else a.variant)
    l = ctypes.CDLL(os.path.abspath(p))
    l.fa_fwd.restype = c_int
    l.fa_fwd.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p]
    l.fa_bwd.restype = c_int
    l.fa_bwd.argtypes = [c_void_p] * 10 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_void_p]
    l.fa_bwd_workspace_bytes.restype = c_longlong
    l.fa_bwd_workspace_bytes.argtypes = [c_int] * 3
    libs.append(l)


def ref64(q, k, v, causal, scale, d_o=None):
    """fp64 reference per head (and autograd gradients when d_o is given)."""
    B, H, N, D = q.shape
    O = torch.empty(B, H, N, D, dtype=torch.float64, device=q.device); L = torch.empty(B, H, N, dtype=torch.float64, device=q.device)
    grads = [torch.empty_like(O) for _ in range(3)] if d_o is not None else None
    for b in range(B):
        for h in range(H):
            qq, kk, vv = (t[b, h].double().requires_grad_(d_o is not None) for t in (q, k, v))
            s = (qq @ kk.T) * scale
            if causal:
                s = s.masked_fill(torch.ones(N, N, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
            L[b, h] = torch.logsumexp(s.detach(), dim=-1)
            o = torch.softmax(s, dim=-1) @ vv
            O[b, h] = o.detach()
            if d_o is not None:
                g = torch.autograd.grad(o, (qq, kk, vv), d_o[b, h].double())
                for i in range(3): grads[i][b, h] = g[i]
            del s, o
    return O, L, grads


for name in a.shapes.split(","):
    B, H, N, D, dt, causal = SHAPES[name]
    H = min(H, a.heads); B = 1
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "fp8": torch.float8_e4m3fn}[dt]; fdt = {"f16": 1, "bf16": 2, "fp8": 3}[dt]
    odt = torch.bfloat16 if dt == "fp8" else tdt
    for amp in [float(x) for x in a.amps.split(",")]:
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v = (((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1) * amp).to(tdt) for _ in range(3))
        scale = D ** -0.5
        do_bwd = a.bwd and dt != "fp8" and D in (64, 128) and N <= 4096
        d_o = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1)).to(tdt) if do_bwd else None
        O64, L64, G64 = ref64(q.float(), k.float(), v.float(), causal, scale, d_o.float() if do_bwd else None)
        smax = float((L64.abs()).max())
        for i, l in enumerate(libs):
            o = torch.empty(B, H, N, D, dtype=odt, device="cuda"); lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            rc = l.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, D, scale, H * N * D, N * D, causal, fdt, variants[i], st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            eo = (o.double() - O64).abs(); el = (lse.double() - L64).abs()
            line = (f"{name:6s} amp {amp:3.0f} {os.path.basename(a.libs[i]):28s} max|O-ref| {eo.max().item():.3e} rms {eo.pow(2).mean().sqrt().item():.3e} "
                    f"max|O| {O64.abs().max().item():.2f} | max|LSE-ref| {el.max().item():.3e} rms {el.pow(2).mean().sqrt().item():.3e} max|LSE| {smax:.2f}")
            if do_bwd:
                dq, dk, dv = (torch.empty(B, H, N, D, dtype=torch.float32, device="cuda") for _ in range(3))
                ws = torch.empty(l.fa_bwd_workspace_bytes(B, H, N), dtype=torch.uint8, device="cuda")
                rc = l.fa_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                              dv.data_ptr(), ws.data_ptr(), B, H, N, D, scale, H * N * D, N * D, causal, fdt, st)
                assert rc == 0, rc
                torch.cuda.synchronize()
                rel = [((x.double() - r).abs().max() / r.abs().max()).item() for x, r in zip((dq, dk, dv), G64)]
                line += f" | bwd rel err dQ {rel[0]:.2e} dK {rel[1]:.2e} dV {rel[2]:.2e}"
            print(line, flush=True)
