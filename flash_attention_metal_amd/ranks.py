"""One-process-per-GPU plumbing for the sharded operator: rank discovery, the timing barrier and
the max-over-ranks reduction. Control plane only -- the data path has no collective (every
(batch, head) slice is independent, /root/reference/kernels.metal:622).
Backend: "nccl" (= RCCL on ROCm) when the ranks own GPUs, "gloo" on CPU (tests)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from .shard import shard_heads


@dataclass
class RankInfo:
    rank: int
    local_rank: int
    world: int
    backend: Optional[str]  # None when world == 1


def init_ranks(use_gpu: bool) -> RankInfo:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return RankInfo(0, local_rank, 1, None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    backend = "nccl" if use_gpu else "gloo"
    kw = {}
    if use_gpu:
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return RankInfo(rank, local_rank, world, backend)


def barrier(info: RankInfo, device: Optional[torch.device] = None) -> None:
    """synchronize -> barrier -> synchronize (the bracket bench.py puts on both sides of the timed steps)."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if info.world > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(info: RankInfo, value: float, device: Optional[torch.device] = None) -> float:
    if info.world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(info: RankInfo, value: float, device: Optional[torch.device] = None) -> float:
    if info.world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(info: RankInfo, value: float, device: Optional[torch.device] = None) -> list:
    """Every rank's value, in rank order, on every rank (per-rank throughput for the imbalance fields of bench.py)."""
    if info.world == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.empty_like(t) for _ in range(info.world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def my_slices(info: RankInfo, slices_per_rank: int):
    """Weak scaling: the global problem has slices_per_rank * world (batch, head) slices; this rank's range."""
    return shard_heads(slices_per_rank * info.world, info.world, info.rank)


def aggregate_throughput(info: RankInfo, my_units: float, my_elapsed_s: float, device=None):
    """value = units all ranks processed / max over ranks of the elapsed time (bench.py contract)."""
    total = sum_over_ranks(info, my_units, device)
    worst = max_over_ranks(info, my_elapsed_s, device)
    return total / worst, worst


def finalize(info: RankInfo) -> None:
    if info.world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
