"""torch.library custom op over the C-ABI.

    torch.ops.fa_mi355.attention_forward(q, k, v, is_causal, scale) -> (o, lse)

Lets the gfx950 kernel be called like any other torch operator (dispatcher, torch.compile graphs,
fake-tensor shape propagation) and compared in-process with scaled_dot_product_attention. The op is
registered for the CUDA/HIP device only -- there is deliberately no CPU implementation.

Autograd: the op carries a backward formula that saves (q, k, v, o, lse) and calls fa_bwd()
(csrc/fa_bwd_kernels.hip; the math of /root/reference/kernels.metal:905-1265, which consumes the
forward's LSE), through fa_bwd_ex: grouped-query heads (Hq % Hkv == 0) and Nq != Nk (causal: Nk >= Nq) are
differentiated too. Shapes fa_bwd_ex has no kernel for (head dims above 128 or not a multiple of 8, fp32 inputs)
raise instead of returning a silent zero gradient; so does a gradient flowing into the LSE output.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .ops import FaError, flash_attention_backward, flash_attention_forward, load_library

_LIB = torch.library.Library("fa_mi355", "DEF")
_LIB.define("attention_forward(Tensor q, Tensor k, Tensor v, bool is_causal=False, float scale=0.0) -> (Tensor, Tensor)")


def _impl(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, is_causal: bool = False, scale: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor]:
    o, lse = flash_attention_forward(q, k, v, is_causal=is_causal, scale=(scale if scale > 0 else None))
    return o, lse


def _meta(q, k, v, is_causal=False, scale=0.0):
    out_dtype = torch.bfloat16 if q.dtype == getattr(torch, "float8_e4m3fn", None) else q.dtype
    B, H, N, _ = q.shape
    return torch.empty_like(q, dtype=out_dtype), q.new_empty((B, H, N), dtype=torch.float32)


_LIB.impl("attention_forward", _impl, "CUDA")
_LIB.impl("attention_forward", _meta, "Meta")


def _setup_context(ctx, inputs, output):
    q, k, v, is_causal, scale = inputs
    o, lse = output
    ctx.save_for_backward(q, k, v, o, lse)
    ctx.is_causal, ctx.scale = bool(is_causal), float(scale)
    ctx.set_materialize_grads(False)


def _backward(ctx, grad_o, grad_lse):
    q, k, v, o, lse = ctx.saved_tensors
    if grad_lse is not None:
        raise NotImplementedError("fa_mi355::attention_forward: no gradient through the LSE output")
    if grad_o is None:
        return None, None, None, None, None
    B, H, N, D = q.shape
    gqa_ok = k.dim() == 4 and (k.shape[0], k.shape[3]) == (B, D) and H % k.shape[1] == 0 and not (ctx.is_causal and k.shape[2] < N)
    fa_dt = {torch.float16: 1, torch.bfloat16: 2, getattr(torch, "float8_e4m3fn", None): 3}
    if not gqa_ok or q.dtype not in fa_dt or not load_library().fa_bwd_supported(fa_dt[q.dtype], D):
        raise FaError(-2, f"no backward kernel for q {tuple(q.shape)} k {tuple(k.shape)} {q.dtype} "
                          "(fa_bwd_ex: f16 / bf16 / e4m3, Hq % Hkv == 0, causal needs Nk >= Nq, head_dim a multiple of 8 up to 128)")
    go = grad_o.to(o.dtype)  # (e4m3 inputs: O and its gradient are bf16)
    if go.stride() != q.stride():
        go = torch.empty_strided(q.shape, q.stride(), dtype=o.dtype, device=q.device).copy_(go)
    dq, dk, dv = flash_attention_backward(q, k, v, o, go, lse, is_causal=ctx.is_causal,
                                          scale=(ctx.scale if ctx.scale > 0 else None))
    return dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype), None, None


torch.library.register_autograd("fa_mi355::attention_forward", _backward, setup_context=_setup_context, lib=_LIB)


def attention_forward(q, k, v, is_causal: bool = False, scale: float = 0.0):
    return torch.ops.fa_mi355.attention_forward(q, k, v, is_causal, scale)
