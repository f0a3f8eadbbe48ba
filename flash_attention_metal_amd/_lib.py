"""ctypes loader of csrc/libfa_mi355.so (the C-ABI of include/fa_mi355.h)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, "libfa_mi355.so")

# every symbol include/fa_mi355.h declares: (restype, argtypes)
SYMBOLS = {
    "fa_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                       c_float, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p]),
    "fa_fwd_ex": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float] + [c_longlong] * 4 + [c_int, c_int, c_void_p]),
    "fa_fwd_exv": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float] + [c_longlong] * 4 + [c_int, c_int, c_int, c_void_p]),
    "fa_fwd_decode": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float] + [c_longlong] * 4 + [c_int, c_int, c_void_p, c_longlong, c_void_p]),
    "fa_fwd_decode_kv8": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float] + [c_longlong] * 4 + [c_int, c_int, c_void_p, c_longlong, c_void_p]),
    "fa_fwd_decode_workspace_bytes": (c_longlong, [c_int] * 6),
    "fa_fwd_decode_supported": (c_int, [c_int] * 5),
    "fa_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_float, c_longlong, c_longlong, c_int, c_int, c_void_p]),
    "fa_bwd_ex": (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_float] + [c_longlong] * 4 + [c_int, c_int, c_void_p]),
    "fa_bwd_workspace_bytes": (c_longlong, [c_int, c_int, c_int]),
    "fa_bwd_workspace_bytes_ex": (c_longlong, [c_int] * 7 + [c_longlong] * 4),
    "fa_bwd_supported": (c_int, [c_int, c_int]),
    "fa_bwd_algorithmic_flops": (c_double, [c_int, c_int, c_int, c_int, c_int]),
    "fa_supported": (c_int, [c_int, c_int, c_int]),
    "fa_resolve_variant": (c_int, [c_int, c_int]),
    "fa_resolve_variant_for": (c_int, [c_int] * 6),
    "fa_fwd_kernel_name": (c_char_p, [c_int] * 6),
    "fa_dtype_in_bytes": (c_int, [c_int]),
    "fa_dtype_out_bytes": (c_int, [c_int]),
    "fa_algorithmic_flops": (c_double, [c_int, c_int, c_int, c_int, c_int]),
    "fa_algorithmic_bytes": (c_double, [c_int, c_int, c_int, c_int, c_int]),
    "fa_last_error": (c_char_p, []),
    "fa_version": (c_int, []),
    "fa_variant_name": (c_char_p, [c_int]),
    "fa_dtype_name": (c_char_p, [c_int]),
}


class LibraryNotBuilt(RuntimeError):
    pass


def lib_path() -> str:
    return _SO


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the kernel library, in-tree."""
    args = ["make", "-C", _CSRC, "-j4"]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"])
    subprocess.check_call(args)
    return _SO


_lib = None


def load_library() -> ctypes.CDLL:
    """Load the HIP kernel library or raise -- never fall back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise LibraryNotBuilt(
            f"{_SO} is missing: build it with `make -C {_CSRC}` (or __graft_entry__.build()). "
            "There is no CPU fallback for the attention operator.")
    lib = ctypes.CDLL(_SO)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
