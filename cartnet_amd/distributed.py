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

import datetime
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def _forced() -> bool:
    """CARTNET_DIST_FORCE=1: run every collective even in a world of ONE rank.  That is how the RCCL code path
    (communicator init, the 10 MB all-reduce, barrier, MAX/MIN reductions) is exercised on a box with a single card:
    ``backend="nccl"`` with world_size 1 goes through librccl exactly like a multi-rank job, only the ring is trivial."""
    return os.environ.get("CARTNET_DIST_FORCE", "") not in ("", "0")


def _active() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def init_from_env(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract).
    Returns (rank, world_size, local_rank); a no-op for single-process runs (unless CARTNET_DIST_FORCE is set)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _forced()) and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm; CARTNET_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs
            backend = os.environ.get("CARTNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # The rendezvous and every collective give up after CARTNET_DIST_TIMEOUT seconds and the process exits non-zero.
        # Default 1800 s = torch's own for RCCL: a training job has legitimate rank skew of minutes (rank 0 writing a
        # checkpoint or evaluating while the others wait in the next epoch's first all-reduce, uneven shard preprocessing, a
        # first-use build of the library); bench.py, where a rank that never arrives must not hold the box, sets 300.
        timeout = datetime.timedelta(seconds=float(os.environ.get("CARTNET_DIST_TIMEOUT", "1800")))
        if backend == "nccl":
            # the device BEFORE any collective, and named to the process group: RCCL then builds its communicator for this
            # device at once (a bad rank -> device map fails here, not at the first all-reduce) and never has to guess
            if local >= torch.cuda.device_count():
                raise RuntimeError(f"LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) are visible: RCCL needs "
                                   "one device per rank (CARTNET_DIST_BACKEND=gloo rehearses more ranks than devices)")
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, disjoint, exhaustive partition of ``n_items`` crystals over ranks (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def balanced_partition(weights: Sequence[int], parts: int) -> List[range]:
    """Contiguous, disjoint, exhaustive partition of ``len(weights)`` items into ``parts`` ranges whose weight sums
    are as equal as a prefix-sum cut allows (shards balanced by EDGE count -- the step time of a rank is
    proportional to its edges, and the job runs at the pace of the slowest rank).  Cut k sits where the running sum is
    nearest to k/parts of the total; every range is non-empty whenever there are at least ``parts`` items."""
    n = len(weights)
    if parts <= 0:
        raise ValueError("parts must be positive")
    pre = [0]
    for w in weights:
        pre.append(pre[-1] + max(int(w), 0) + 1)     # +1: zero-weight items still count, cuts stay strictly ordered
    total = pre[-1]
    cuts = [0]
    j = 0
    for k in range(1, parts):
        target = total * k / parts
        while j < n and pre[j + 1] <= target:
            j += 1
        c = j + 1 if j < n and (pre[j + 1] - target) < (target - pre[j]) else j     # nearer of the two neighbours
        lo = min(cuts[-1] + 1, n) if n >= parts else cuts[-1]                      # keep this part non-empty ...
        hi = n - (parts - k) if n >= parts else n                                  # ... and leave one for each later part
        cuts.append(max(lo, min(c, hi)))
    cuts.append(n)
    return [range(cuts[i], cuts[i + 1]) for i in range(parts)]


def rank_batches(weights: Sequence[int], batch_size: int, rank: int, world: int) -> List[range]:
    """The batches of ``rank`` for one epoch as index ranges into the (already permuted) crystal order.

    Every rank gets a contiguous edge-balanced slice (no crystal is dropped) and cuts it into the SAME number of
    batches -- ceil(n / (world * batch_size)) -- again balanced by edges, so that all ranks take the same number of
    optimiser steps (one all-reduce each) and every step carries about the same number of edges on every rank.
    ``batch_size`` is therefore nominal for world > 1: a rank holding small crystals packs a few more of them per
    step.  A rank with fewer crystals than steps (tiny data sets only) gets empty ranges at the end: it contributes a
    zero gradient to those steps."""
    n = len(weights)
    mine = balanced_partition(weights, world)[rank]
    n_batches = max(1, -(-n // (world * max(int(batch_size), 1))))
    w = [weights[i] for i in mine]
    return [range(mine.start + r.start, mine.start + r.stop) for r in balanced_partition(w, n_batches)]


def all_reduce_gradients(flat_grad: torch.Tensor) -> float:
    """SUM-all-reduce the flat gradient buffer in place; returns the scale (1/world) the optimiser should apply."""
    if not _active():
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


class GradSync:
    """The gradient all-reduce in BUCKETS, overlapped with backward (reference accumulation boundary:
    train/train.py:186-189).  ``cartnet_model_backward`` reports every bucket of the flat gradient buffer (head, layers
    L-1 .. 0, encoder: CartnetGradReadyFn in include/cartnet_hip.h) on a stream ordered behind the kernels that produced
    it; ``bucket(lo, hi)`` -- called with that stream current -- queues an asynchronous SUM all-reduce of
    ``flat_grad[lo:hi]`` (RCCL runs it on its own stream, behind the current one), and backward goes on underneath.
    ``finish()`` makes the CURRENT stream wait for all of them and returns the 1/world scale for the optimiser, like
    ``all_reduce_gradients``.  The summed values are the same as one flat all-reduce's (element-wise SUM over ranks;
    with more than two ranks the transport's summation order may differ between a bucket and the flat buffer in the last
    bit).  ``exposed_ms()``: GPU time the joining stream spent waiting, averaged over the finish() calls so far."""

    def __init__(self, flat_grad: torch.Tensor, measure: bool = False):
        self.flat = flat_grad
        self.works: list = []
        self.measure = measure and flat_grad.is_cuda
        self._events: list = []
        self.buckets_seen = 0

    @property
    def active(self) -> bool:
        return _active()

    def bucket(self, lo: int, hi: int) -> None:
        self.buckets_seen += 1
        if _active() and hi > lo:
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> float:
        if not _active():
            self.works.clear()
            return 1.0
        ev = None
        if self.measure:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self.works:
            w.wait()                  # stream-level for RCCL: the current stream waits, the host does not
        self.works.clear()
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        return 1.0 / dist.get_world_size()

    def exposed_ms(self) -> Optional[float]:
        if not self._events:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)


def ranks_seen(device) -> int:
    """How many ranks the communicator really carries: a SUM all-reduce of one 1 per rank."""
    if not _active():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def all_reduce_sum_(t: torch.Tensor) -> None:
    """In-place SUM all-reduce of a (device) tensor, enqueued in the current stream's order; no-op for one rank.
    The sync-BatchNorm exchange of cartnet_amd.model.CartNet (2D+1 doubles per BatchNorm and direction)."""
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def broadcast_buffers(model: torch.nn.Module, src: int = 0) -> None:
    if not _active():
        return
    for b in model.buffers():
        dist.broadcast(b, src=src)


def assert_replicas_in_sync(model: torch.nn.Module) -> None:
    """Data-parallel invariant: every rank applied the same averaged gradients, so the parameters must be identical on
    all ranks (BatchNorm running statistics are per rank and not compared).  One MAX and one MIN all-reduce of two
    checksums; raises on every rank if they differ."""
    if not _active():
        return
    params = [p.detach().double() for p in model.parameters()]
    dev = params[0].device
    chk = torch.stack([sum(p.sum() for p in params), sum((p * p).sum() for p in params)]).to(dev)
    hi, lo = chk.clone(), chk.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    if not torch.equal(hi, lo):
        raise RuntimeError(f"parameters diverged across ranks: checksums differ by {(hi - lo).tolist()}")


def min_max_over_ranks(value: float, device) -> tuple[float, float]:
    """(min, max) of a per-rank scalar over the ranks (one MIN and one MAX all-reduce)."""
    if not _active():
        return value, value
    lo = torch.tensor([value], dtype=torch.float64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if _active():
        dist.barrier()
