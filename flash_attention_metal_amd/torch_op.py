"""torch.library custom op over the C-ABI.

    torch.ops.fa_mi355.attention_forward(q, k, v, is_causal, scale) -> (o, lse)

Lets the gfx950 kernel be called like any other torch operator (dispatcher, torch.compile graphs,
fake-tensor shape propagation) and compared in-process with scaled_dot_product_attention. The op is
registered for the CUDA/HIP device only -- there is deliberately no CPU implementation.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .ops import flash_attention_forward

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


def attention_forward(q, k, v, is_causal: bool = False, scale: float = 0.0):
    return torch.ops.fa_mi355.attention_forward(q, k, v, is_causal, scale)
