"""The operator: flash_attention_forward(Q, K, V, is_causal) -> (O, LSE).

Argument meaning follows the reference binding table
(/root/reference/kernels.metal:600-613, host side /root/reference/main.mm:821-852):
tensors are ``[B, H, N, D]`` with rows contiguous; batch/head strides are taken
from the tensors; ``scale`` defaults to ``1/sqrt(D)`` (main.mm:13); ``lse`` is
``[B, H, N]`` fp32 (kernels.metal:611). Errors follow the C-ABI: a negative
status becomes :class:`FaError` carrying ``fa_last_error()``.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from ._lib import load_library

DTYPES = {"f32": 0, "f16": 1, "bf16": 2, "fp8_e4m3": 3}
VARIANTS = {"auto": 0, "naive": 1, "tiled": 2, "tiled_v2": 3, "mfma": 4, "mfma_pp": 5, "mfma_splitkv": 6, "mfma_split2": 7, "mfma_exact": 8, "mfma_h64s2": 9, "mfma16": 10, "mfma_fp8pv": 11}

_TORCH2FA = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
if hasattr(torch, "float8_e4m3fn"):
    _TORCH2FA[torch.float8_e4m3fn] = 3


class FaError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"fa_fwd failed ({status}): {msg}")
        self.status = status


def supported(dtype: str, variant: str, D: int) -> bool:
    return bool(load_library().fa_supported(DTYPES[dtype], VARIANTS[variant], D))


def algorithmic_flops(B: int, H: int, N: int, D: int, is_causal: bool) -> float:
    return float(load_library().fa_algorithmic_flops(B, H, N, D, int(is_causal)))


def algorithmic_bytes(B: int, H: int, N: int, D: int, dtype: str) -> float:
    return float(load_library().fa_algorithmic_bytes(B, H, N, D, DTYPES[dtype]))


def forward_kernel_name(dtype: str, D: int, is_causal: bool, B: int = 4, H: int = 16, N: int = 4096) -> str:
    """Name (as rocprofv3 prints it) of the device kernel variant "auto" launches for the problem."""
    return load_library().fa_fwd_kernel_name(DTYPES[dtype], D, B, H, N, int(is_causal)).decode()


def _strides(t: torch.Tensor) -> Tuple[int, int]:
    """(batch_stride, head_stride) in elements; size-1 dims get the dense value."""
    B, H, N, D = t.shape
    sb, sh, sn, sd = t.stride()
    if sd != 1 or sn != D:
        raise ValueError("rows must be contiguous with pitch D (kernels.metal:622: offset = b*bs + h*hs)")
    hs = sh if H > 1 else N * D
    bs = sb if B > 1 else H * hs
    return bs, hs


def _prepare_forward(q, k, v, is_causal, scale, variant, return_lse, out, lse):
    """Validate one (Q,K,V,O,LSE) call and return (argument tuple of fa_fwd without the stream, out, lse)."""
    if q.dim() != 4 or q.shape != k.shape or q.shape != v.shape:
        raise ValueError(f"q, k, v must share one [B,H,N,D] shape, got {tuple(q.shape)} {tuple(k.shape)} {tuple(v.shape)}")
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise RuntimeError("flash_attention_forward needs device tensors: there is no CPU path "
                           "(the CPU oracle lives in oracle/ and is test infrastructure only)")
    if q.dtype not in _TORCH2FA or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError(f"unsupported / mixed dtypes {q.dtype} {k.dtype} {v.dtype}")
    B, H, N, D = q.shape
    bs, hs = _strides(q)
    if _strides(k) != (bs, hs) or _strides(v) != (bs, hs):
        raise ValueError("q, k, v must share batch/head strides (one stride pair in the binding table)")
    fa_dtype = _TORCH2FA[q.dtype]
    out_dtype = torch.bfloat16 if fa_dtype == 3 else q.dtype
    if out is None:
        out = torch.empty_strided((B, H, N, D), q.stride(), dtype=out_dtype, device=q.device)
    elif (not out.is_cuda or out.device != q.device or out.dtype != out_dtype or out.shape != q.shape
          or _strides(out) != (bs, hs)):
        raise ValueError("out must be a device tensor with q's shape/strides (bf16 for fp8 inputs)")
    if return_lse and lse is None:
        lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    if lse is not None and (not lse.is_cuda or lse.device != q.device or lse.dtype != torch.float32
                            or not lse.is_contiguous() or lse.numel() != B * H * N):
        raise ValueError("lse must be contiguous fp32 [B,H,N] on q's device")
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr() if lse is not None else None,
            B, H, N, D, float(scale), bs, hs, int(bool(is_causal)), fa_dtype, VARIANTS[variant])
    return args, out, lse


def _launch_forward(lib, args, device, stream):
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream
    if device.index == torch.cuda.current_device():  # the common case: no device switch (it costs microseconds,
        st = lib.fa_fwd(*args, stream)                # as much as a short-sequence kernel runs)
    else:
        with torch.cuda.device(device):
            st = lib.fa_fwd(*args, stream)
    if st != 0:
        raise FaError(st, lib.fa_last_error().decode())


def flash_attention_forward(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    is_causal: bool = False,
    scale: Optional[float] = None,
    variant: str = "auto",
    return_lse: bool = True,
    out: Optional[torch.Tensor] = None,
    lse: Optional[torch.Tensor] = None,
    stream: Optional[int] = None,
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Launch the gfx950 kernel on the current torch stream (asynchronous)."""
    lib = load_library()
    if q.dim() == 4 and k.dim() == 4 and k.shape == v.shape and q.shape != k.shape:
        return _forward_ex(lib, q, k, v, is_causal, scale, variant, return_lse, out, lse, stream)  # GQA / Nq != Nk
    args, out, lse = _prepare_forward(q, k, v, is_causal, scale, variant, return_lse, out, lse)
    _launch_forward(lib, args, q.device, stream)
    return out, lse


class ForwardPlan:
    """One validated forward call on fixed tensors, launched many times: the argument table is checked and encoded
    once (what the reference host does per dispatch, /root/reference/main.mm:821-852), ``launch()`` only issues
    ``fa_fwd``. For launch-bound shapes -- short sequences, where the Python-side checks of
    ``flash_attention_forward`` cost as much as the kernel runs -- and for replaying a step loop. The plan keeps the
    tensors alive; it does not notice if they are resized or freed behind its back (``Tensor.set_`` / ``resize_``)."""

    def __init__(self, q, k, v, is_causal=False, scale=None, variant="auto", return_lse=True, out=None, lse=None):
        self._lib = load_library()
        self._args, self.out, self.lse = _prepare_forward(q, k, v, is_causal, scale, variant, return_lse, out, lse)
        self._keep = (q, k, v)
        self._device = q.device

    def launch(self, stream: Optional[int] = None) -> None:
        """Issue the kernel on ``stream`` (default: the current torch stream of the tensors' device); asynchronous."""
        _launch_forward(self._lib, self._args, self._device, stream)


def _forward_ex(lib, q, k, v, is_causal, scale, variant, return_lse, out, lse, stream):
    """Generalised call (include/fa_mi355.h fa_fwd_exv): q [B,Hq,Nq,D], k/v [B,Hkv,Nk,D], Hq % Hkv == 0,
    causal bottom-right aligned (key j visible to query i iff j <= i + Nk - Nq). `variant`: auto, mfma, mfma_exact, mfma16 or
    mfma_splitkv -- the kernels that take grouped heads / Nq != Nk; any other raises FaError (unsupported), never a silent substitute."""
    B, Hq, Nq, D = q.shape
    Bk, Hkv, Nk, Dk = k.shape
    if Bk != B or Dk != D or Hq % Hkv:
        raise ValueError(f"incompatible shapes q {tuple(q.shape)} k/v {tuple(k.shape)}")
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise RuntimeError("flash_attention_forward needs device tensors: there is no CPU path")
    if q.dtype not in _TORCH2FA or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError(f"unsupported / mixed dtypes {q.dtype} {k.dtype} {v.dtype}")
    qbs, qhs = _strides(q)
    kbs, khs = _strides(k)
    if _strides(v) != (kbs, khs):
        raise ValueError("k and v must share batch/head strides")
    fa_dtype = _TORCH2FA[q.dtype]
    out_dtype = torch.bfloat16 if fa_dtype == 3 else q.dtype
    if out is None:
        out = torch.empty_strided((B, Hq, Nq, D), q.stride(), dtype=out_dtype, device=q.device)
    elif (not out.is_cuda or out.device != q.device or out.dtype != out_dtype or out.shape != q.shape
          or _strides(out) != (qbs, qhs)):
        # the kernel writes O with q's strides: anything else lands in the wrong place or out of bounds
        raise ValueError("out must be a device tensor with q's shape/strides (bf16 for fp8 inputs)")
    if return_lse and lse is None:
        lse = torch.empty((B, Hq, Nq), dtype=torch.float32, device=q.device)
    if lse is not None and (not lse.is_cuda or lse.device != q.device or lse.dtype != torch.float32
                            or not lse.is_contiguous() or lse.numel() != B * Hq * Nq):
        raise ValueError("lse must be contiguous fp32 [B,Hq,Nq] on q's device")
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if stream is None:
        stream = torch.cuda.current_stream(q.device).cuda_stream
    with torch.cuda.device(q.device):
        st = lib.fa_fwd_exv(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr() if lse is not None else None,
                            B, Hq, Hkv, Nq, Nk, D, float(scale), qbs, qhs, kbs, khs, int(bool(is_causal)), fa_dtype, VARIANTS[variant], stream)
    if st != 0:
        raise FaError(st, lib.fa_last_error().decode())
    return out, lse


def decode_workspace_bytes(B: int, Hq: int, Hkv: int, Nq: int, Nk: int, D: int) -> int:
    return int(load_library().fa_fwd_decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D))


def flash_attention_decode(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    is_causal: bool = False,
    scale: Optional[float] = None,
    return_lse: bool = True,
    out: Optional[torch.Tensor] = None,
    lse: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
    stream: Optional[int] = None,
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Few query rows against a long key sequence (include/fa_mi355.h fa_fwd_decode): q [B,Hq,Nq,D], k/v [B,Hkv,Nk,D] with
    (Hq / Hkv) * Nq <= 32, f16 / bf16 / e4m3 -- or bf16 queries on an e4m3 k / v cache --, D = 64 | 128; the same operator as
    flash_attention_forward on these shapes, laid out for the HBM roofline. `workspace`: a uint8 device tensor of at least decode_workspace_bytes(...) bytes (allocated here if None --
    the C entry point itself allocates nothing)."""
    lib = load_library()
    if q.dim() != 4 or k.dim() != 4 or k.shape != v.shape:
        raise ValueError("q [B,Hq,Nq,D], k / v [B,Hkv,Nk,D]")
    B, Hq, Nq, D = q.shape
    Bk, Hkv, Nk, Dk = k.shape
    if Bk != B or Dk != D or Hq % Hkv:
        raise ValueError(f"incompatible shapes q {tuple(q.shape)} k/v {tuple(k.shape)}")
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise RuntimeError("flash_attention_decode needs device tensors: there is no CPU path")
    fp8 = getattr(torch, "float8_e4m3fn", None)
    kv8 = fp8 is not None and k.dtype == fp8 and v.dtype == fp8 and q.dtype == torch.bfloat16  # an e4m3 KV cache under bf16 queries (fa_fwd_decode_kv8)
    if q.dtype not in (torch.float16, torch.bfloat16, fp8) or v.dtype != k.dtype or (k.dtype != q.dtype and not kv8):
        raise ValueError(f"unsupported / mixed dtypes {q.dtype} {k.dtype} {v.dtype} (one of f16 / bf16 / e4m3, or bf16 queries on an e4m3 cache)")
    odt = torch.bfloat16 if q.dtype == fp8 else q.dtype  # e4m3 inputs: bf16 output, as in flash_attention_forward
    qbs, qhs = _strides(q)
    kbs, khs = _strides(k)
    if _strides(v) != (kbs, khs):
        raise ValueError("k and v must share batch/head strides")
    if out is None:
        out = torch.empty_strided((B, Hq, Nq, D), q.stride(), dtype=odt, device=q.device)
    elif not out.is_cuda or out.device != q.device or out.dtype != odt or out.shape != q.shape or _strides(out) != (qbs, qhs):
        raise ValueError("out must be a device tensor with q's shape/strides")
    if return_lse and lse is None:
        lse = torch.empty((B, Hq, Nq), dtype=torch.float32, device=q.device)
    if lse is not None and (not lse.is_cuda or lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != B * Hq * Nq):
        raise ValueError("lse must be contiguous fp32 [B,Hq,Nq] on q's device")
    need = decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D)
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    elif not workspace.is_cuda or workspace.device != q.device or workspace.dtype != torch.uint8 or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous uint8 device tensor")
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if stream is None:
        stream = torch.cuda.current_stream(q.device).cuda_stream
    with torch.cuda.device(q.device):
        entry = lib.fa_fwd_decode_kv8 if kv8 else lib.fa_fwd_decode
        st = entry(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr() if lse is not None else None,
                   B, Hq, Hkv, Nq, Nk, D, float(scale), qbs, qhs, kbs, khs, int(bool(is_causal)), _TORCH2FA[q.dtype],
                   workspace.data_ptr(), workspace.numel(), stream)
    if st != 0:
        raise FaError(st, lib.fa_last_error().decode())
    return out, lse


def flash_attention_backward(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    o: torch.Tensor,
    d_o: torch.Tensor,
    lse: torch.Tensor,
    is_causal: bool = False,
    scale: Optional[float] = None,
    stream: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(Q,K,V,O,dO,LSE) -> (dQ,dK,dV) in fp32, laid out like their inputs (binding table of
    /root/reference/kernels.metal:905-921; gradients are written, not accumulated). k / v may carry fewer heads than q and
    another sequence length (include/fa_mi355.h fa_bwd_ex, the counterpart of fa_fwd_ex): dK / dV have k's shape."""
    lib = load_library()
    if q.dim() != 4 or any(t.shape != q.shape for t in (o, d_o)) or k.dim() != 4 or v.shape != k.shape:
        raise ValueError("q, o, d_o must share one [B,Hq,N,D] shape and k, v one [B,Hkv,N,D] shape")
    if not all(t.is_cuda for t in (q, k, v, o, d_o, lse)):
        raise RuntimeError("flash_attention_backward needs device tensors: there is no CPU path")
    fp8 = q.dtype == torch.float8_e4m3fn
    if fp8:  # e4m3 Q, K, V with the bf16 O the forward wrote for them (and a bf16 dO)
        if any(t.dtype != q.dtype for t in (k, v)) or any(t.dtype != torch.bfloat16 for t in (o, d_o)):
            raise ValueError("backward with e4m3 q needs e4m3 k, v and bf16 o, d_o")
    elif q.dtype not in (torch.float16, torch.bfloat16) or any(t.dtype != q.dtype for t in (k, v, o, d_o)):
        raise ValueError("backward supports f16 / bf16 tensors of one dtype (or e4m3 q, k, v with bf16 o, d_o)")
    B, H, N, D = q.shape
    Bk, Hkv, Nk, Dk = k.shape
    if (Bk, Dk) != (B, D) or H % Hkv:
        raise ValueError(f"k/v shape {tuple(k.shape)} does not fit q {tuple(q.shape)} (same B, D; Hq % Hkv == 0)")
    bs, hs = _strides(q)
    kbs, khs = _strides(k)
    if any(_strides(t) != (bs, hs) for t in (o, d_o)) or _strides(v) != (kbs, khs):
        raise ValueError("q, o, d_o must share batch/head strides, and so must k, v")
    if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != B * H * N:
        raise ValueError("lse must be contiguous fp32 [B,H,N]")
    dq = torch.empty_strided((B, H, N, D), q.stride(), dtype=torch.float32, device=q.device)
    dk, dv = (torch.empty_strided((B, Hkv, Nk, D), k.stride(), dtype=torch.float32, device=q.device) for _ in range(2))
    ws = torch.empty(lib.fa_bwd_workspace_bytes_ex(_TORCH2FA[q.dtype], B, H, Hkv, N, Nk, D, bs, hs, kbs, khs), dtype=torch.uint8,
                     device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if stream is None:
        stream = torch.cuda.current_stream(q.device).cuda_stream
    with torch.cuda.device(q.device):
        st = lib.fa_bwd_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
                           dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, H, Hkv, N, Nk, D, float(scale), bs, hs,
                           kbs, khs, int(bool(is_causal)), _TORCH2FA[q.dtype], stream)
    if st != 0:
        raise FaError(st, lib.fa_last_error().decode())
    return dq, dk, dv
