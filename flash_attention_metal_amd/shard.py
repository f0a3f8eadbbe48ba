"""(batch, head) sharding of the operator across the GPUs of a node.

Every (batch, head) slice is an independent attention problem -- the kernel's
only cross-slice coupling is the base offset (/root/reference/kernels.metal:622)
-- so the B*H slices are block-distributed over ranks with no exchange step and
no collective on the data path.
"""
from __future__ import annotations

from typing import Tuple


def shard_heads(n_slices: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Half-open range [lo, hi) of the flattened (batch*head) slices rank owns.

    Block distribution; the first ``n_slices % world_size`` ranks get one extra.
    Ranges are disjoint and cover [0, n_slices) exactly.
    """
    if world_size < 1 or not (0 <= rank < world_size) or n_slices < 0:
        raise ValueError(f"bad shard request n_slices={n_slices} world_size={world_size} rank={rank}")
    q, r = divmod(n_slices, world_size)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return lo, hi
