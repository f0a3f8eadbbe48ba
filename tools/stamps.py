#!/usr/bin/env python3
"""Per-block phase times of the 128-row matrix-core kernel from a -DFA_STAMPS build (the LSE buffer carries s_memtime
stamps of wave 0: 0 block start, 1 after tile 0, 2 after the tile loop, 3 after the epilogue, 4 kernel entry, 7 tile count).
usage: stamps.py lib.so persist|plain [shape]"""
import ctypes, os, sys
from ctypes import c_int, c_float, c_longlong, c_void_p
import numpy as np, torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1])); persist = sys.argv[2] == "persist"
shape = {"c3": (4, 16, 4096), "c8k": (4, 16, 8192), "c16k": (1, 64, 16384)}[sys.argv[3] if len(sys.argv) > 3 else "c3"]
B, H, N = shape; D = 64
lib.fa_fwd.restype = c_int
lib.fa_fwd.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_float, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(3))
o = torch.empty_like(q); lse = torch.zeros(max(B * H * N, 768 * 40 * 8 + 64), dtype=torch.int32, device="cuda")
for it in range(3):
    lse.zero_(); torch.cuda.synchronize()
    rc = lib.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, D, D ** -0.5, H * N * D, N * D, 1, 2, 4, None)
    assert rc == 0; torch.cuda.synchronize()
a = lse.cpu().numpy().astype(np.int64)
per = 40 if persist else 1
nwg = 768 if persist else (N // 128) * B * H
a = a[: nwg * per * 8].reshape(nwg, per, 8)
valid = a[:, :, 7] > 0
d = lambda x, y: ((a[:, :, x] - a[:, :, y]) & 0xFFFFFFFF)[valid]  # noqa: E731
nT = a[:, :, 7][valid]
print(f"blocks stamped {valid.sum()}  tiles/block mean {nT.mean():.1f}")
t0_1, t1_2, t2_3 = d(1, 0), d(2, 1), d(3, 2)
print(f"block start -> after tile 0 : mean {t0_1.mean():8.0f}  median {np.median(t0_1):8.0f} ticks (10 ns)")
print(f"per later tile              : mean {(t1_2 / np.maximum(nT - 1, 1)).mean():8.0f}  median {np.median(t1_2 / np.maximum(nT - 1, 1)):8.0f}")
print(f"tile loop end -> epilogue end: mean {t2_3.mean():8.0f}  median {np.median(t2_3):8.0f}")
ref = int(a[0, 0, 4])
rel = lambda x: (x - ref + (1 << 31)) % (1 << 32) - (1 << 31)  # noqa: E731  (stamps relative to workgroup 0's entry)
if persist:
    gap = ((a[:, 1:, 0] - a[:, :-1, 3]) & 0xFFFFFFFF)[valid[:, 1:] & valid[:, :-1]]
    print(f"epilogue end -> next block start: mean {gap.mean():8.0f} median {np.median(gap):8.0f}  (n={gap.size})")
    first = ((a[:, 0, 0] - a[:, 0, 4]) & 0xFFFFFFFF)
    print(f"kernel entry -> first block start: mean {first.mean():8.0f}")
    nb = valid.sum(axis=1)
    end = np.array([rel(int(a[i, nb[i] - 1, 3])) for i in range(nwg)]); start = rel(a[:, 0, 4])
    print(f"slot entry: min {start.min()} max {start.max()};  slot end: min {end.min()} mean {end.mean():.0f} max {end.max()}")
    print(f"blocks per slot: min {nb.min()} max {nb.max()};  tiles per slot: min {(a[:, :, 7] * valid).sum(1).min()} max {(a[:, :, 7] * valid).sum(1).max()}")
    for x in range(8):
        e = end[x::8]; print(f"  xcd class {x}: end min {e.min()} mean {e.mean():.0f} max {e.max()}  tiles {(a[x::8, :, 7] * valid[x::8]).sum()}")
else:
    pro = d(0, 4)
    print(f"kernel entry -> block start (prologue): mean {pro.mean():8.0f} median {np.median(pro):8.0f}")
    life = d(3, 4)
    start = rel(a[:, 0, 4]); end = rel(a[:, 0, 3])
    print(f"workgroup lifetime: mean {life.mean():.0f}; sum over workgroups / 768 slots = {life.sum() / 768:.0f} ticks (10 ns)")
    print(f"entry: min {start.min()} max {start.max()};  end: max {end.max()}")
    for x in range(8):
        print(f"  xcd class {x}: last end {end[x::8].max()}")
    # concurrency: how many workgroups are alive over time
    ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]; alive = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0]); print(f"time-averaged resident workgroups: {(alive[:-1] * dt).sum() / dt.sum():.1f}; max {alive.max()}")
