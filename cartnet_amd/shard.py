"""Packed crystal shards and device-side batching.

The reference keeps one pickled PyG ``Data`` per structure, ``torch.load``s it per sample in 5 worker processes
(dataset/datasetADP.py:41-42; loader/loader.py:114-124), augments / standardises on the CPU (:33-39,43-45,76-77) and
copies every batch over PCIe.  Here a dataset split is ONE flat file of CSR arrays that is uploaded to HBM once
(the whole ADP dataset -- ~2e5 crystals, ~6e8 edges, ~25 GB -- is a fraction of the 288 GB), and a batch is built by a
single kernel launch on the device (``cartnet_collate``), including the SO(3) augmentation.

File layout (little endian): 8-byte magic ``CNSHARD1``, a JSON header padded to a multiple of 64 bytes (its length
as uint64 right after the magic) listing ``{"name": [dtype, shape, offset]}``, then the arrays, each 64-byte aligned:

    atom_ptr, edge_ptr, y_ptr [G+1] int64 | z [N] int32 | pos [N,3] f32 | non_h_mask [N] u8 |
    edge_src, edge_tgt [E] int32 (atom index inside the crystal) | cart_dist [E] f32 | cart_dir [E,3] f32 |
    cell [G,9] f32 | temperature [G] f32 | y [Y, y_width] f32 (y_width 9: one ADP tensor per non-H atom)

``pos``, ``non_h_mask``, ``cell`` and ``temperature`` are optional (the Jarvis / MP graphs carry no mask or
temperature).
"""
from __future__ import annotations

import json
import struct
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import lib as _l
from .data import Batch, Data

MAGIC = b"CNSHARD1"
_ALIGN = 64
_OPTIONAL = ("pos", "non_h_mask", "cell", "temperature")


def pack(data_list: Sequence[Data]) -> Dict[str, np.ndarray]:
    """Flat CSR arrays of a list of crystals (the attribute set of cartnet_amd.data / SURVEY.md 8a)."""
    if len(data_list) == 0:
        raise ValueError("cannot pack an empty list of crystals")
    n = [int(d.x.shape[0]) for d in data_list]
    e = [int(d.edge_index.shape[1]) for d in data_list]
    per_atom = data_list[0].y.dim() == 3
    ys = [d.y.reshape(-1, 9) if per_atom else d.y.reshape(1, -1) for d in data_list]
    out = {
        "atom_ptr": np.concatenate([[0], np.cumsum(n)]).astype(np.int64),
        "edge_ptr": np.concatenate([[0], np.cumsum(e)]).astype(np.int64),
        "y_ptr": np.concatenate([[0], np.cumsum([y.shape[0] for y in ys])]).astype(np.int64),
        "z": torch.cat([d.x for d in data_list]).numpy().astype(np.int32),
        "edge_src": torch.cat([d.edge_index[0] for d in data_list]).numpy().astype(np.int32),
        "edge_tgt": torch.cat([d.edge_index[1] for d in data_list]).numpy().astype(np.int32),
        "cart_dist": torch.cat([d.cart_dist for d in data_list]).numpy().astype(np.float32),
        "cart_dir": torch.cat([d.cart_dir for d in data_list]).numpy().astype(np.float32).reshape(-1, 3),
        "y": torch.cat(ys).numpy().astype(np.float32),
    }
    d0 = data_list[0]
    if hasattr(d0, "pos"):
        out["pos"] = torch.cat([d.pos for d in data_list]).numpy().astype(np.float32).reshape(-1, 3)
    if hasattr(d0, "non_H_mask"):
        out["non_h_mask"] = torch.cat([d.non_H_mask for d in data_list]).numpy().astype(np.uint8)
    if hasattr(d0, "cell"):
        out["cell"] = torch.cat([d.cell.reshape(1, 9) for d in data_list]).numpy().astype(np.float32)
    if hasattr(d0, "temperature"):
        out["temperature"] = torch.cat([d.temperature.reshape(1) for d in data_list]).numpy().astype(np.float32)
    for i, d in enumerate(data_list):
        if e[i] and bool((d.edge_index[1][1:] < d.edge_index[1][:-1]).any()):
            raise ValueError(f"crystal {i}: edge_index[1] must be sorted ascending")
    return out


def pack_with_gpu_graph(geometries: Sequence[Data], radius: float = 5.0, device="cuda:0", chunk: int = 256,
                        max_neighbors: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Flat CSR arrays (as ``pack``) of crystals given WITHOUT edges -- ``x`` (atomic numbers), ``pos``, ``cell`` and the
    targets -- whose periodic radius graphs are built on the GPU, ``chunk`` crystals per launch pair
    (cartnet_amd.graph.radius_graph_pbc: the reference's dataset/utils.py:57-237 edge order, integers bit-exact), and
    rebased to the crystal.  20,283 crystals of 64-324 atoms (56 M edges): 0.55 s (tools/bench_config4.py)."""
    from .graph import radius_graph_pbc
    if len(geometries) == 0:
        raise ValueError("cannot pack an empty list of crystals")
    dev = torch.device(device)
    n = [int(d.x.shape[0]) for d in geometries]
    src_l, tgt_l, dist_l, dir_l, ecount = [], [], [], [], []
    for c0 in range(0, len(geometries), chunk):
        part = geometries[c0:c0 + chunk]
        pos = torch.cat([d.pos for d in part]).to(dev)
        cell = torch.cat([d.cell.reshape(1, 3, 3) for d in part]).to(dev)
        ptr = torch.tensor([0] + n[c0:c0 + chunk], dtype=torch.int64).cumsum(0).to(dev)
        ei, dist, dirs = radius_graph_pbc(pos, cell, ptr, radius, max_neighbors)
        gid = torch.repeat_interleave(torch.arange(len(part), device=dev), ptr[1:] - ptr[:-1])
        g_of_edge = gid[ei[1]]
        off = ptr[g_of_edge]
        src_l.append((ei[0] - off).to(torch.int32).cpu())
        tgt_l.append((ei[1] - off).to(torch.int32).cpu())
        dist_l.append(dist.cpu())
        dir_l.append(dirs.cpu())
        ecount.append(torch.bincount(g_of_edge, minlength=len(part)).cpu())
    e = torch.cat(ecount).numpy().astype(np.int64)
    d0 = geometries[0]
    per_atom = d0.y.dim() == 3
    ys = [d.y.reshape(-1, 9) if per_atom else d.y.reshape(1, -1) for d in geometries]
    out = {
        "atom_ptr": np.concatenate([[0], np.cumsum(n)]).astype(np.int64),
        "edge_ptr": np.concatenate([[0], np.cumsum(e)]).astype(np.int64),
        "y_ptr": np.concatenate([[0], np.cumsum([y.shape[0] for y in ys])]).astype(np.int64),
        "z": torch.cat([d.x for d in geometries]).numpy().astype(np.int32),
        "pos": torch.cat([d.pos for d in geometries]).numpy().astype(np.float32).reshape(-1, 3),
        "edge_src": torch.cat(src_l).numpy(), "edge_tgt": torch.cat(tgt_l).numpy(),
        "cart_dist": torch.cat(dist_l).numpy().astype(np.float32),
        "cart_dir": torch.cat(dir_l).numpy().astype(np.float32).reshape(-1, 3),
        "cell": torch.cat([d.cell.reshape(1, 9) for d in geometries]).numpy().astype(np.float32),
        "y": torch.cat(ys).numpy().astype(np.float32),
    }
    if hasattr(d0, "non_H_mask"):
        out["non_h_mask"] = torch.cat([d.non_H_mask for d in geometries]).numpy().astype(np.uint8)
    if hasattr(d0, "temperature"):
        out["temperature"] = torch.cat([d.temperature.reshape(1) for d in geometries]).numpy().astype(np.float32)
    return out


def write_shard(path: str, data_list: Sequence[Data]) -> None:
    arrays = pack(data_list)
    meta, off = {}, 0
    for k, a in arrays.items():
        meta[k] = [a.dtype.str, list(a.shape), off]
        off += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    header = json.dumps({"arrays": meta, "graphs": len(data_list)}).encode()
    header += b" " * (-(len(MAGIC) + 8 + len(header)) % _ALIGN)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        for k, a in arrays.items():
            f.write(np.ascontiguousarray(a).tobytes())
            f.write(b"\0" * (-a.nbytes % _ALIGN))


def read_shard(path: str) -> Dict[str, np.ndarray]:
    """Memory-maps the arrays of a shard file (no copy until they are uploaded)."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a CartNet shard")
        (hlen,) = struct.unpack("<Q", f.read(8))
        meta = json.loads(f.read(hlen).decode())
    base = 16 + hlen
    out = {}
    for k, (dt, shape, off) in meta["arrays"].items():
        out[k] = np.memmap(path, dtype=np.dtype(dt), mode="r", offset=base + off, shape=tuple(shape))
    return out


class DeviceShard:
    """A shard resident in HBM.  ``collate(sel)`` builds the batch of crystals ``sel`` with one kernel launch."""

    def __init__(self, arrays: Dict[str, np.ndarray], device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError("DeviceShard lives on the GPU; there is no CPU path (use Batch.from_data_list on the host)")
        need = ("atom_ptr", "edge_ptr", "y_ptr", "z", "edge_src", "edge_tgt", "cart_dist", "cart_dir", "y")
        missing = [k for k in need if k not in arrays]
        if missing:
            raise ValueError(f"shard lacks {missing}")
        self.device = dev
        # host copies of the offsets: batch sizes are known without a device round trip
        self.atom_ptr = np.asarray(arrays["atom_ptr"], dtype=np.int64)
        self.edge_ptr = np.asarray(arrays["edge_ptr"], dtype=np.int64)
        self.y_ptr = np.asarray(arrays["y_ptr"], dtype=np.int64)
        self.num_graphs = int(self.atom_ptr.shape[0] - 1)
        for name, p in (("atom_ptr", self.atom_ptr), ("edge_ptr", self.edge_ptr), ("y_ptr", self.y_ptr)):
            if p.shape[0] != self.num_graphs + 1 or p[0] != 0 or bool((np.diff(p) < 0).any()):
                raise ValueError(f"{name} is not a valid offset array")
        sizes = {"z": self.atom_ptr[-1], "pos": self.atom_ptr[-1], "non_h_mask": self.atom_ptr[-1],
                 "edge_src": self.edge_ptr[-1], "edge_tgt": self.edge_ptr[-1], "cart_dist": self.edge_ptr[-1],
                 "cart_dir": self.edge_ptr[-1], "cell": self.num_graphs, "temperature": self.num_graphs,
                 "y": self.y_ptr[-1]}
        self.t: Dict[str, torch.Tensor] = {}
        for k, a in arrays.items():
            if k in sizes and int(a.shape[0]) != int(sizes[k]):
                raise ValueError(f"{k}: {a.shape[0]} rows, expected {int(sizes[k])}")
            self.t[k] = torch.from_numpy(np.array(a)).to(dev)       # np.array: memmaps are read-only
        self.y_width = int(arrays["y"].shape[1]) if arrays["y"].ndim == 2 else 1
        self.per_atom_target = self.y_width == 9
        d = _l.Shard()
        for k in ("atom_ptr", "edge_ptr", "y_ptr", "z", "pos", "non_h_mask", "edge_src", "edge_tgt", "cart_dist",
                  "cart_dir", "cell", "temperature", "y"):
            setattr(d, k, self.t[k].data_ptr() if k in self.t else None)
        d.y_width = self.y_width
        self._desc = d
        self._lib = _l.load()

    @classmethod
    def from_file(cls, path: str, device="cuda:0") -> "DeviceShard":
        return cls(read_shard(path), device)

    @classmethod
    def from_data_list(cls, data_list: Sequence[Data], device="cuda:0") -> "DeviceShard":
        return cls(pack(data_list), device)

    def nbytes(self) -> int:
        return sum(v.numel() * v.element_size() for v in self.t.values())

    def collate(self, sel: Sequence[int], rot: Optional[torch.Tensor] = None, temp_mean: float = 0.0,
                temp_std: float = 1.0) -> Batch:
        """Batch of crystals ``sel`` (in that order) with PyG's collation rules (cartnet_amd/data.py), on the GPU.
        ``rot`` [B,3,3] fp32 (device): per-crystal augmentation rotation (dataset/datasetADP.py:33-39)."""
        sel_np = np.asarray(sel, dtype=np.int64).reshape(-1)
        B = int(sel_np.shape[0])
        if B == 0:
            raise ValueError("cannot collate an empty selection")
        if int(sel_np.min()) < 0 or int(sel_np.max()) >= self.num_graphs:
            raise IndexError("crystal index out of range")
        meta = np.empty(4 * B + 3, dtype=np.int64)                 # [sel | atom offsets | edge offsets | target offsets]
        meta[:B] = sel_np
        for j, p in enumerate((self.atom_ptr, self.edge_ptr, self.y_ptr)):
            seg = meta[B + j * (B + 1):B + (j + 1) * (B + 1)]
            seg[0] = 0
            np.cumsum(p[sel_np + 1] - p[sel_np], out=seg[1:])
        N, E, M = (int(meta[B + j * (B + 1) + B]) for j in range(3))
        dev = self.device
        meta_d = torch.from_numpy(meta).pin_memory().to(dev, non_blocking=True)
        if rot is not None:
            if not (rot.is_cuda and rot.dtype == torch.float32 and tuple(rot.shape) == (B, 3, 3)):
                raise ValueError("rot must be a CUDA fp32 tensor [B,3,3]")
            rot = rot.contiguous()
        b = Batch()
        b.x = torch.empty(N, dtype=torch.int64, device=dev)
        b.batch = torch.empty(N, dtype=torch.int64, device=dev)
        b.ptr = torch.empty(B + 1, dtype=torch.int64, device=dev)
        b.edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
        b.cart_dist = torch.empty(E, dtype=torch.float32, device=dev)
        b.cart_dir = torch.empty((E, 3), dtype=torch.float32, device=dev)
        b.y = torch.empty((M, 3, 3) if self.per_atom_target else ((M,) if self.y_width == 1 else (M, self.y_width)),
                          dtype=torch.float32, device=dev)
        if "pos" in self.t:
            b.pos = torch.empty((N, 3), dtype=torch.float32, device=dev)
        if "non_h_mask" in self.t:
            b.non_H_mask = torch.empty(N, dtype=torch.bool, device=dev)
        if "cell" in self.t:
            b.cell = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
        if "temperature" in self.t:
            b.temperature = torch.empty(B, dtype=torch.float32, device=dev)
        o = _l.Collated()
        o.x, o.batch, o.ptr = b.x.data_ptr(), b.batch.data_ptr(), b.ptr.data_ptr()
        o.edge_index, o.cart_dist, o.cart_dir = b.edge_index.data_ptr(), b.cart_dist.data_ptr(), b.cart_dir.data_ptr()
        o.y = b.y.data_ptr()
        o.pos = b.pos.data_ptr() if "pos" in self.t else None
        o.non_h_mask = b.non_H_mask.data_ptr() if "non_h_mask" in self.t else None
        o.cell = b.cell.data_ptr() if "cell" in self.t else None
        o.temperature = b.temperature.data_ptr() if "temperature" in self.t else None
        base = meta_d.data_ptr()
        _l.check(self._lib.cartnet_collate(_l.C.byref(self._desc), base, base + 8 * B, base + 8 * (2 * B + 1),
                                           base + 8 * (3 * B + 2), B, N, E, M,
                                           rot.data_ptr() if rot is not None else None, float(temp_mean),
                                           float(temp_std), _l.C.byref(o), _l.stream_ptr()), "cartnet_collate")
        b.num_graphs = B
        b._meta = meta_d                      # keeps the offsets alive until the kernel has run
        return b


def random_rotations(n: int, gen: torch.Generator, device) -> torch.Tensor:
    """[n,3,3] uniform rotations from unit quaternions, generated on the device (stands in for
    roma.utils.random_rotmat, dataset/datasetADP.py:34)."""
    q = torch.randn(n, 4, generator=gen, device=device, dtype=torch.float32)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(n, 3, 3).contiguous()


class ShardLoader:
    """Loader over a resident shard with the interface of cartnet_amd.data.DataLoader: seeded shuffling, ``rank`` /
    ``world_size`` crystal sharding (every rank walks the same permutation and takes a disjoint edge-balanced slice), optional
    SO(3) augmentation.  Batches are born on the GPU; the host only draws the permutation."""

    def __init__(self, shard: DeviceShard, batch_size: int, shuffle: bool = False, seed: int = 0, rank: int = 0,
                 world_size: int = 1, drop_last: bool = False, augment: bool = False, temp_mean: float = 0.0,
                 temp_std: float = 1.0, indices: Optional[Sequence[int]] = None):
        self.shard, self.batch_size, self.shuffle, self.seed = shard, int(batch_size), shuffle, seed
        self.rank, self.world_size, self.drop_last, self.augment = rank, world_size, drop_last, augment
        self.temp_mean, self.temp_std = temp_mean, temp_std
        self.indices = list(range(shard.num_graphs)) if indices is None else list(indices)
        self.epoch = 0
        self._gen = torch.Generator(device=shard.device).manual_seed(seed + 7919 * rank)

    def _batches(self) -> List[List[int]]:
        """Crystal ids of this rank's batches for the current epoch (same rule as cartnet_amd.data.DataLoader: one
        rank -> consecutive chunks of the permutation; several -> edge-balanced contiguous slices, nothing dropped,
        equally many batches per rank; edge counts come from the shard's host copy of ``edge_ptr``)."""
        n = len(self.indices)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = [self.indices[i] for i in torch.randperm(n, generator=g).tolist()]
        else:
            order = list(self.indices)
        if self.world_size > 1:
            from .distributed import rank_batches
            ep = self.shard.edge_ptr
            weights = [int(ep[j + 1] - ep[j]) for j in order]
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
            if not chunk:
                yield None
                continue
            rot = random_rotations(len(chunk), self._gen, self.shard.device) if self.augment else None
            yield self.shard.collate(chunk, rot, self.temp_mean, self.temp_std)
