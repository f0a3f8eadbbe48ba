"""MI355X-native attention forward: (Q, K, V, is_causal) -> O (+ LSE).

Host-side mirror of the reference operator (the Metal binding table of
``flash_attention_v4_half_kernel``, /root/reference/kernels.metal:600-613, as
the host binds it at /root/reference/main.mm:821-852) over the C-ABI in
``include/fa_mi355.h``. The compute is hand-written HIP for gfx950 in
``csrc/``; torch is used for device memory and streams only. There is no CPU
fallback: importing works anywhere, calling the operator without the built
library or without a GPU raises.
"""
from ._lib import LibraryNotBuilt, lib_path, load_library, build_library  # noqa: F401
from .ops import (  # noqa: F401
    DTYPES,
    VARIANTS,
    FaError,
    ForwardPlan,
    algorithmic_bytes,
    algorithmic_flops,
    decode_workspace_bytes,
    flash_attention_backward,
    flash_attention_decode,
    flash_attention_forward,
    forward_kernel_name,
    supported,
)
from .shard import shard_heads  # noqa: F401

__version__ = "0.4.0"
