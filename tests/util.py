"""Shared helpers for the parity tests (test side only: imports the oracle)."""
import numpy as np

TORCH_DTYPE = {"f32": "float32", "f16": "float16", "bf16": "bfloat16", "fp8": "float8_e4m3fn"}


def make_qkv(oracle, B, H, N, D, dtype, seeds=(42, 43, 44), amp=1.0):
    """Independent Q,K,V ~ U(-1,1) (main.mm:24-30 generator, seeds 42/43/44), rounded RNE to dtype.

    Returns fp32 numpy arrays holding exactly-representable values of `dtype`."""
    out = []
    for s in seeds:
        x = oracle.init_random(B * H * N * D, s).reshape(B, H, N, D) * np.float32(amp)
        out.append(oracle.round_to(x, dtype))
    return out


def to_dev(x, dtype):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x)).to(getattr(torch, TORCH_DTYPE[dtype])).cuda()


def run_op(fa, q, k, v, dtype, causal, variant="auto", scale=None):
    import torch

    o, lse = fa.flash_attention_forward(to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype),
                                        is_causal=causal, variant=variant, scale=scale)
    torch.cuda.synchronize()
    return o.float().cpu().numpy(), lse.cpu().numpy()
