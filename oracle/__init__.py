"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the timed CPU baseline.
The product (``flash_attention_metal_amd``) never imports it.

``libfa_oracle.so``           our C restatement (oracle/attn_oracle.c) of
                              /root/reference/main.mm:24-30,128-159,551-578.
``_ref/libfa_ref_slices.so``  the reference's own loops, sliced by line number
                              at build time (oracle/build_ref.sh); optional.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_longlong, c_uint32

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfa_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libfa_ref_slices.so")

_fp = POINTER(c_float)
_dp = POINTER(c_double)


def build(force: bool = False) -> None:
    """Compile the oracle (and the reference slices when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "attn_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libfa_oracle.so"])
    if os.path.exists("/root/reference/main.mm") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


def _load() -> ctypes.CDLL:
    if not os.path.exists(_LIB_PATH):
        build()
    lib = ctypes.CDLL(_LIB_PATH)
    lib.oracle_init_random.argtypes = [_fp, c_longlong, c_uint32]
    lib.oracle_noncausal_faithful.argtypes = [_fp, _fp, _fp, _fp, c_int, c_int, c_float]
    lib.oracle_noncausal_hoisted.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, c_float]
    lib.oracle_causal.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, c_float]
    lib.oracle_attn_fwd.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, c_float,
                                    c_longlong, c_longlong, c_int, c_int]
    lib.oracle_attn_fwd_f64.argtypes = [_fp, _fp, _fp, _dp, _dp, c_int, c_int, c_int, c_int,
                                        c_float, c_longlong, c_longlong, c_int, c_int]
    lib.oracle_attn_rows_f64.argtypes = [_fp, _fp, _fp, _dp, _dp, c_int, c_int, c_float, c_int,
                                         POINTER(c_int), c_int, c_int]
    lib.oracle_attn_fwd_ex_f64.argtypes = [_fp, _fp, _fp, _dp, _dp, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int]
    lib.oracle_attn_bwd_f64.argtypes = [_fp, _fp, _fp, _fp, _dp, _dp, _dp, c_int, c_int, c_int, c_int, c_float, c_int, c_int]
    for name in ("oracle_round_f16", "oracle_round_bf16", "oracle_round_fp8_e4m3"):
        getattr(lib, name).argtypes = [_fp, c_longlong]
    lib.oracle_max_threads.restype = c_int
    return lib


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def _d(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def init_random(size: int, seed: int = 42) -> np.ndarray:
    """main.mm:24-30: mt19937(seed) -> U(-1,1) floats (reference seed is 42)."""
    out = np.empty(size, dtype=np.float32)
    lib().oracle_init_random(_f(out), size, seed)
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def noncausal_faithful(q, k, v, scale):
    N, D = q.shape
    o = np.empty_like(q)
    lib().oracle_noncausal_faithful(_f(q), _f(k), _f(v), _f(o), N, D, scale)
    return o


def noncausal_hoisted(q, k, v, scale):
    N, D = q.shape
    o = np.empty_like(q)
    lse = np.empty(N, dtype=np.float32)
    lib().oracle_noncausal_hoisted(_f(q), _f(k), _f(v), _f(o), _f(lse), N, D, scale)
    return o, lse


def causal(q, k, v, scale):
    N, D = q.shape
    o = np.empty_like(q)
    lse = np.empty(N, dtype=np.float32)
    lib().oracle_causal(_f(q), _f(k), _f(v), _f(o), _f(lse), N, D, scale)
    return o, lse


def attn_fwd(q, k, v, is_causal: bool, scale: float | None = None, threads: int = 0):
    """(Q,K,V,is_causal) -> (O, LSE) over contiguous [B,H,N,D] fp32 arrays."""
    B, H, N, D = q.shape
    if scale is None:
        scale = float(np.float32(1.0) / np.float32(np.sqrt(D)))
    if threads <= 0:
        threads = max_threads()
    o = np.empty_like(q)
    lse = np.empty((B, H, N), dtype=np.float32)
    lib().oracle_attn_fwd(_f(q), _f(k), _f(v), _f(o), _f(lse), B, H, N, D, scale,
                          H * N * D, N * D, int(is_causal), threads)
    return o, lse


def attn_fwd_f64(q, k, v, is_causal: bool, scale: float | None = None, threads: int = 0):
    B, H, N, D = q.shape
    if scale is None:
        scale = float(np.float32(1.0) / np.float32(np.sqrt(D)))
    if threads <= 0:
        threads = max_threads()
    o = np.empty(q.shape, dtype=np.float64)
    lse = np.empty((B, H, N), dtype=np.float64)
    lib().oracle_attn_fwd_f64(_f(q), _f(k), _f(v), _d(o), _d(lse), B, H, N, D, scale,
                              H * N * D, N * D, int(is_causal), threads)
    return o, lse


def attn_rows_f64(q, k, v, rows, is_causal: bool, scale: float | None = None, threads: int = 0):
    """fp64 (O rows, LSE rows) for the query rows `rows` of one contiguous [N,D] head."""
    N, D = q.shape
    if scale is None:
        scale = float(np.float32(1.0) / np.float32(np.sqrt(D)))
    if threads <= 0:
        threads = max_threads()
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    o = np.empty((len(rows), D), dtype=np.float64)
    lse = np.empty(len(rows), dtype=np.float64)
    lib().oracle_attn_rows_f64(_f(q), _f(k), _f(v), _d(o), _d(lse), N, D, scale, int(is_causal),
                               rows.ctypes.data_as(POINTER(c_int)), len(rows), threads)
    return o, lse


def attn_fwd_ex_f64(q, k, v, is_causal: bool, scale: float | None = None, threads: int = 0):
    """Generalised fp64 (O, LSE): q [B,Hq,Nq,D], k/v [B,Hkv,Nk,D] (grouped-query heads, Nq != Nk)."""
    B, Hq, Nq, D = q.shape
    _, Hkv, Nk, _ = k.shape
    assert Hq % Hkv == 0 and v.shape == k.shape
    if scale is None:
        scale = float(np.float32(1.0) / np.float32(np.sqrt(D)))
    if threads <= 0:
        threads = max_threads()
    o = np.empty(q.shape, dtype=np.float64)
    lse = np.empty((B, Hq, Nq), dtype=np.float64)
    lib().oracle_attn_fwd_ex_f64(_f(q), _f(k), _f(v), _d(o), _d(lse), B, Hq, Hkv, Nq, Nk, D, scale, int(is_causal), threads)
    return o, lse


def attn_bwd_f64(q, k, v, d_o, is_causal: bool, scale: float | None = None, threads: int = 0):
    """fp64 (dQ, dK, dV) over contiguous [B,H,N,D] fp32 inputs."""
    B, H, N, D = q.shape
    if scale is None:
        scale = float(np.float32(1.0) / np.float32(np.sqrt(D)))
    if threads <= 0:
        threads = max_threads()
    dq, dk, dv = (np.empty(q.shape, dtype=np.float64) for _ in range(3))
    lib().oracle_attn_bwd_f64(_f(q), _f(k), _f(v), _f(d_o), _d(dq), _d(dk), _d(dv), B, H, N, D, scale, int(is_causal), threads)
    return dq, dk, dv


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """fp32 -> dtype -> fp32 with round-to-nearest-even (fp8: saturating e4m3fn)."""
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    fn = {"f32": None, "f16": "oracle_round_f16", "bf16": "oracle_round_bf16",
          "fp8": "oracle_round_fp8_e4m3"}[dtype]
    if fn is not None:
        getattr(lib(), fn)(_f(y.reshape(-1)), y.size)
    return y


# ---------------------------------------------------------------------------
# the reference's own loops (optional; only where oracle/_ref was built)
# ---------------------------------------------------------------------------
def have_ref() -> bool:
    return os.path.exists(_REF_PATH)


_ref = None


def ref() -> ctypes.CDLL:
    global _ref
    if _ref is None:
        r = ctypes.CDLL(_REF_PATH)
        r.ref_head_dim.restype = c_int
        r.ref_scale.restype = c_float
        r.ref_init_random.argtypes = [_fp, c_int]
        r.ref_noncausal.argtypes = [_fp, _fp, _fp, _fp, c_int]
        r.ref_causal.argtypes = [_fp, _fp, _fp, _fp, c_int]
        _ref = r
    return _ref


def ref_init_random(size: int) -> np.ndarray:
    out = np.empty(size, dtype=np.float32)
    ref().ref_init_random(_f(out), size)
    return out


def ref_noncausal(q, k, v):
    N, D = q.shape
    assert D == ref().ref_head_dim()
    o = np.empty_like(q)
    ref().ref_noncausal(_f(q), _f(k), _f(v), _f(o), N)
    return o


def ref_causal(q, k, v):
    N, D = q.shape
    assert D == ref().ref_head_dim()
    o = np.empty_like(q)
    ref().ref_causal(_f(q), _f(k), _f(v), _f(o), N)
    return o
