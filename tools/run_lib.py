#!/usr/bin/env python3
"""run_shape.py for an arbitrary build of the library (tools/ab/lib_<name>.so): the target of per-arm PMC passes.
usage: run_lib.py B H N D dtype causal iters variant_number LIB.so"""
import ctypes, os, sys
from ctypes import c_int, c_float, c_longlong, c_void_p
import torch
B, H, N, D = map(int, sys.argv[1:5]); dt = sys.argv[5]; causal = int(sys.argv[6]); iters = int(sys.argv[7])
variant = int(sys.argv[8]); lib = ctypes.CDLL(os.path.abspath(sys.argv[9]))
lib.fa_fwd.restype = c_int
lib.fa_fwd.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p]
tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "fp8": torch.float8_e4m3fn}[dt]; fdt = {"f16": 1, "bf16": 2, "fp8": 3}[dt]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(tdt) for _ in range(3))
o = torch.empty_like(q, dtype=torch.bfloat16 if dt == "fp8" else tdt); lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3 + iters):
    rc = lib.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, D, D ** -0.5,
                    H * N * D, N * D, causal, fdt, variant, st)
    assert rc == 0, rc
torch.cuda.synchronize()
print("done")
