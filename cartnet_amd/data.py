"""Minimal ``Data`` / ``Batch`` containers with PyG's collation rules for the attributes CartNet reads.

The reference consumes ``torch_geometric.data.Batch`` objects produced by PyG's ``DataLoader``
(reference: train/train.py:167-171).  torch_geometric is not installed on the
target image, so this module restates the collation contract for exactly the attribute set on the hot path:

  * node-level tensors (``x``, ``non_H_mask``, ``pos``, ``y`` when per-atom) are concatenated on dim 0;
  * ``edge_index`` [2, E_g] is concatenated on dim 1 after adding the cumulative node offset of each graph;
  * edge-level tensors (``cart_dist``, ``cart_dir``) are concatenated on dim 0;
  * per-graph tensors (``temperature`` [1], ``cell`` [1,3,3], scalar ``y``) are concatenated on dim 0;
  * ``batch`` [N] (sorted graph id per node) and ``ptr`` [Bg+1] are added.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch

_NODE_KEYS = {"x", "non_H_mask", "pos"}
_EDGE_KEYS = {"cart_dist", "cart_dir"}
_GRAPH_KEYS = {"temperature", "cell", "natoms"}


class Data:
    """Attribute bag of tensors describing one crystal graph (or a collated batch of them)."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__.items() if not k.startswith("_")]

    def to(self, device, non_blocking: bool = False):
        """In-place move of every tensor attribute; returns self (PyG semantics, train/train.py:169)."""
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def clone(self):
        out = self.__class__()
        for k, v in self.__dict__.items():
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """A collated batch of graphs.  ``num_graphs`` is kept on the host so no device sync is needed."""

    @classmethod
    def from_data_list(cls, data_list: Sequence[Data]) -> "Batch":
        if len(data_list) == 0:
            raise ValueError("cannot collate an empty list of graphs")
        out = cls()
        keys = data_list[0].keys()
        n_nodes = [d.num_nodes for d in data_list]
        offsets = [0]
        for n in n_nodes:
            offsets.append(offsets[-1] + n)
        for k in keys:
            vals = [getattr(d, k) for d in data_list]
            if not torch.is_tensor(vals[0]):
                setattr(out, k, vals)
                continue
            if k == "edge_index":
                setattr(out, k, torch.cat([v + off for v, off in zip(vals, offsets[:-1])], dim=1))
            elif k == "y":
                # per-atom targets [M_g,3,3] or per-graph scalars ([] or [1]) -- both concatenate on dim 0
                vals = [v.reshape(1) if v.dim() == 0 else v for v in vals]
                setattr(out, k, torch.cat(vals, dim=0))
            else:
                vals = [v.reshape(1) if v.dim() == 0 else v for v in vals]
                setattr(out, k, torch.cat(vals, dim=0))
        out.batch = torch.repeat_interleave(
            torch.arange(len(data_list), dtype=torch.int64), torch.tensor(n_nodes, dtype=torch.int64)
        )
        out.ptr = torch.tensor(offsets, dtype=torch.int64)
        out.num_graphs = len(data_list)
        return out


class DataLoader:
    """Tiny single-process loader: shuffles graph indices with a seeded generator and collates ``batch_size``
    graphs at a time (stand-in for ``torch_geometric.loader.DataLoader``; reference: loader/loader.py:114-124).

    ``rank`` / ``world_size`` give the batched-graph sharding mode: every rank walks the same permutation and
    takes a contiguous, disjoint, edge-balanced slice of it (no crystal is dropped), so crystals are partitioned
    across GPUs with no data-path collective.
    """

    def __init__(self, dataset: Sequence[Data], batch_size: int, shuffle: bool = False, seed: int = 0,
                 rank: int = 0, world_size: int = 1, drop_last: bool = False, transform=None):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.seed = seed
        self.rank = rank
        self.world_size = world_size
        self.drop_last = drop_last
        self.transform = transform
        self.epoch = 0

    def _order(self) -> List[int]:
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            return torch.randperm(n, generator=g).tolist()
        return list(range(n))

    def _batches(self) -> List[List[int]]:
        """Crystal indices of this rank's batches for the current epoch.  One rank: consecutive chunks of
        ``batch_size`` (PyG's DataLoader).  Several ranks: cartnet_amd.distributed.rank_batches -- contiguous
        EDGE-balanced slices of the common permutation, nothing dropped, the same number of batches on every rank."""
        order = self._order()
        if self.world_size > 1:
            from .distributed import rank_batches
            weights = [int(self.dataset[j].edge_index.shape[1]) for j in order]
            return [[order[i] for i in r] for r in rank_batches(weights, self.batch_size, self.rank, self.world_size)]
        chunks = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        return chunks

    def __len__(self) -> int:
        return len(self._batches())

    def __iter__(self) -> Iterable[Batch]:
        batches = self._batches()
        self.epoch += 1
        for chunk in batches:
            if not chunk:                 # fewer crystals than steps on this rank: a step with a zero gradient
                yield None
                continue
            items = [self.dataset[j] for j in chunk]
            if self.transform is not None:
                items = [self.transform(d.clone()) for d in items]
            yield Batch.from_data_list(items)
