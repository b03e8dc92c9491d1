"""Batched-graph sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Crystals are independent graphs, so the data path needs no collective: every rank runs the whole network on its own
crystals.  The only exchange is one all-reduce (SUM) of the flat 10 MB gradient buffer per optimiser step -- issued
once at the gradient-accumulation boundary (reference accumulates 16 micro-batches, train/train.py:186-189), so with
xGMI's ~153 GB/s per link it is latency-bound (~0.1 ms) next to >= 10 ms of compute.  The mean over ranks is folded
into the fused Adam kernel's ``grad_scale``.  The reference itself has no distributed code.

BatchNorm statistics stay per rank (standard data-parallel semantics); ``broadcast_buffers`` keeps the running
statistics identical on every rank when a checkpoint is written.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract).
    Returns (rank, world_size, local_rank); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm; CARTNET_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs
            backend = os.environ.get("CARTNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, disjoint, exhaustive partition of ``n_items`` crystals over ranks (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def all_reduce_gradients(flat_grad: torch.Tensor) -> float:
    """SUM-all-reduce the flat gradient buffer in place; returns the scale (1/world) the optimiser should apply."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def broadcast_buffers(model: torch.nn.Module, src: int = 0) -> None:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for b in model.buffers():
        dist.broadcast(b, src=src)


def assert_replicas_in_sync(model: torch.nn.Module) -> None:
    """Data-parallel invariant: every rank applied the same averaged gradients, so the parameters must be identical on
    all ranks (BatchNorm running statistics are per rank and not compared).  One MAX and one MIN all-reduce of two
    checksums; raises on every rank if they differ."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params = [p.detach().double() for p in model.parameters()]
    dev = params[0].device
    chk = torch.stack([sum(p.sum() for p in params), sum((p * p).sum() for p in params)]).to(dev)
    hi, lo = chk.clone(), chk.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    if not torch.equal(hi, lo):
        raise RuntimeError(f"parameters diverged across ranks: checksums differ by {(hi - lo).tolist()}")


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
