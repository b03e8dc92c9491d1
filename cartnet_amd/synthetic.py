"""Synthetic ADP-shaped crystals and a CPU periodic radius graph.

There is no network on the build or GPU boxes, so the real ADP / Jarvis datasets are unavailable; benchmarks and
parity tests run on synthetic crystals whose size, density and attribute layout follow SURVEY.md §8(d):
per graph ``g`` (seed 1234+g) ``n`` atoms at uniform fractional coordinates in a sheared cubic cell scaled to
36.36 Å^3 per atom, so a 5 Å cutoff gives ~14.4 neighbours per atom (~2.8k edges at 194 atoms).

``radius_graph_pbc_single`` restates the *edge ordering and filtering* of the reference's graph builder
(reference: dataset/utils.py:57-237 ``radius_graph_pbc``; called by dataset/figshare_dataset.py:65-68):
edges are enumerated target-major (index1), then source (index2), then periodic image in
``cartesian_prod(arange(-r1..r1), arange(-r2..r2), arange(-r3..r3))`` order; pairs with d^2 > radius^2 or
d^2 <= 1e-4 are dropped; ``edge_index = stack(source, target)`` so row 1 is sorted ascending.
It is checked bit-exactly (integers) against the reference on the golden fixtures (tests/golden).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from .data import Batch, Data

TEMP_MEAN = 192.1785  # reference: dataset/datasetADP.py:17
TEMP_STD = 81.2135    # reference: dataset/datasetADP.py:18
VOLUME_PER_ATOM = 36.36


def neighbor_cap_mask(tgt: torch.Tensor, d2: torch.Tensor, n: int, max_neighbors: int,
                      tolerance: float = 0.01) -> torch.Tensor:
    """Boolean keep-mask of the reference's neighbour cap (dataset/utils.py:240-360, enforce_max_strictly False):
    a target with more than ``max_neighbors`` edges keeps those with d^2 <= (max_neighbors+1)-th smallest d^2 of its
    row + tolerance.  ``tgt`` sorted ascending, ``d2`` fp32 squared distances."""
    keep = torch.ones_like(tgt, dtype=torch.bool)
    deg = torch.bincount(tgt, minlength=n)
    start = torch.cumsum(deg, 0) - deg
    for i in torch.nonzero(deg > max_neighbors).flatten().tolist():
        row = d2[start[i]:start[i] + deg[i]]
        cutoff = torch.sort(row).values[max_neighbors] + tolerance        # fp32 add, as the reference's tensor + float
        keep[start[i]:start[i] + deg[i]] = torch.le(row, cutoff)
    return keep


def radius_graph_pbc_single(pos: torch.Tensor, cell: torch.Tensor, radius: float = 5.0, chunk: int = 64,
                            max_neighbors: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Periodic radius graph of ONE crystal on the CPU.

    pos [n,3] float32 Cartesian, cell [3,3] float32 (rows are lattice vectors).
    Returns (edge_index [2,E] int64 = (source, target), cart_dist [E] float32, cart_dir [E,3] float32) with
    ``cart_dir = (pos_target - (pos_source + offset)) / dist`` (reference: dataset/utils.py:196-198,
    dataset/figshare_dataset.py:67-68).
    """
    pos = pos.to(torch.float32)
    cell = cell.to(torch.float32).reshape(3, 3)
    n = pos.shape[0]
    # number of periodic repetitions per lattice direction (reference: dataset/utils.py:133-157)
    cross_a2a3 = torch.cross(cell[1], cell[2], dim=-1)
    cell_vol = torch.sum(cell[0] * cross_a2a3, dim=-1, keepdim=True)
    rep1 = torch.ceil(radius * torch.norm(cross_a2a3 / cell_vol, p=2, dim=-1))
    cross_a3a1 = torch.cross(cell[2], cell[0], dim=-1)
    rep2 = torch.ceil(radius * torch.norm(cross_a3a1 / cell_vol, p=2, dim=-1))
    cross_a1a2 = torch.cross(cell[0], cell[1], dim=-1)
    rep3 = torch.ceil(radius * torch.norm(cross_a1a2 / cell_vol, p=2, dim=-1))
    cells_per_dim = [torch.arange(-float(r), float(r) + 1, dtype=torch.float32) for r in (rep1, rep2, rep3)]
    unit_cell = torch.cartesian_prod(*cells_per_dim)          # [C,3], a1 slowest
    n_cells = unit_cell.shape[0]
    # Cartesian offset of every image: cell^T @ unit_cell^T (reference: dataset/utils.py:180-181)
    offsets = torch.bmm(cell.t().unsqueeze(0), unit_cell.t().unsqueeze(0))[0]   # [3,C]

    src_l: List[torch.Tensor] = []
    tgt_l: List[torch.Tensor] = []
    d2_l: List[torch.Tensor] = []
    dir_l: List[torch.Tensor] = []
    r2 = radius * radius
    idx2 = torch.arange(n, dtype=torch.int64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        p1 = pos[s:e].view(-1, 1, 3, 1)                        # targets
        p2 = pos.view(1, n, 3, 1) + offsets.view(1, 1, 3, n_cells)   # sources + image offset
        direction = p1 - p2                                    # [c,n,3,C]
        d2 = torch.sum(direction ** 2, dim=2)                  # [c,n,C]
        mask = torch.logical_and(torch.le(d2, r2), torch.gt(d2, 0.0001))
        t_idx, s_idx, c_idx = torch.nonzero(mask, as_tuple=True)   # row-major: target, source, image
        tgt_l.append(t_idx + s)
        src_l.append(idx2[s_idx])
        d2_l.append(d2[t_idx, s_idx, c_idx])
        dir_l.append(direction[t_idx, s_idx, :, c_idx])
    src = torch.cat(src_l)
    tgt = torch.cat(tgt_l)
    vec = torch.cat(dir_l)
    if max_neighbors is not None and max_neighbors > 0:                  # dataset/utils.py:216-233
        keep = neighbor_cap_mask(tgt, torch.cat(d2_l), n, max_neighbors)
        src, tgt, vec = src[keep], tgt[keep], vec[keep]
    edge_index = torch.stack((src, tgt))
    cart_dist = torch.norm(vec, p=2, dim=-1)
    cart_dir = torch.nn.functional.normalize(vec, p=2, dim=-1)
    return edge_index, cart_dist, cart_dir


def random_rotation(gen: torch.Generator) -> torch.Tensor:
    """Uniform random rotation matrix from a unit quaternion (stands in for roma.utils.random_rotmat,
    reference: dataset/datasetADP.py:34)."""
    q = torch.randn(4, generator=gen, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    R = torch.tensor([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ], dtype=torch.float64)
    return R.to(torch.float32)


def augment_data(data: Data, gen: torch.Generator) -> Data:
    """SO(3) augmentation of one graph (reference: dataset/datasetADP.py:33-39): y <- R^T y R,
    cart_dir <- cart_dir R, cell <- cell R."""
    R = random_rotation(gen)
    if data.y.dim() == 3:
        data.y = R.transpose(-1, -2) @ data.y @ R
    data.cart_dir = data.cart_dir @ R
    data.cell = data.cell @ R
    return data


def make_geometry(g: int, n_atoms: Optional[int] = 194, n_range=(64, 324), adp: bool = True,
                  base_seed: int = 1234) -> Data:
    """Everything of synthetic crystal ``g`` except its edges (SURVEY.md §8d recipe): atoms, cell, positions, targets.
    ``make_crystal`` adds the periodic radius graph on the host; tools/bench_config4.py builds it on the GPU instead
    (cartnet_amd.graph.radius_graph_pbc) for tens of thousands of crystals."""
    gen = torch.Generator().manual_seed(base_seed + g)
    if n_atoms is None:
        n = int(torch.randint(n_range[0], n_range[1] + 1, (1,), generator=gen).item())
    else:
        n = int(n_atoms)
    G = torch.randn(3, 3, generator=gen, dtype=torch.float32)
    cell = torch.eye(3) + 0.1 * G
    det = torch.det(cell).abs().item()
    scale = (n * VOLUME_PER_ATOM / det) ** (1.0 / 3.0)
    cell = (cell * scale).to(torch.float32)
    frac = torch.rand(n, 3, generator=gen, dtype=torch.float32)
    pos = frac @ cell
    # atomic numbers: P(H)=.45, C .35, N .07, O .10, remaining 3% uniform Z in [9,53]
    u = torch.rand(n, generator=gen)
    other = torch.randint(9, 54, (n,), generator=gen)
    z = torch.where(u < 0.45, torch.tensor(1), torch.where(u < 0.80, torch.tensor(6), torch.where(
        u < 0.87, torch.tensor(7), torch.where(u < 0.97, torch.tensor(8), other)))).to(torch.int64)
    temperature = ((90.0 + 210.0 * torch.rand(1, generator=gen)) - TEMP_MEAN) / TEMP_STD
    d = Data(x=z, pos=pos, cell=cell.unsqueeze(0), natoms=torch.tensor([n]))
    if adp:
        mask = z != 1
        m = int(mask.sum().item())
        A = torch.randn(m, 3, 3, generator=gen, dtype=torch.float32)
        d.y = 0.01 * A @ A.transpose(1, 2) + 0.005 * torch.eye(3)
        d.non_H_mask = mask
        d.temperature = temperature.to(torch.float32)
    else:
        d.y = torch.randn(1, generator=gen, dtype=torch.float32)
    return d


def make_crystal(g: int, n_atoms: Optional[int] = 194, radius: float = 5.0, n_range=(64, 324),
                 adp: bool = True, base_seed: int = 1234) -> Data:
    """One synthetic crystal graph with the attribute set CartNet reads (SURVEY.md §8a batch table)."""
    geo = make_geometry(g, n_atoms, n_range, adp, base_seed)
    edge_index, cart_dist, cart_dir = radius_graph_pbc_single(geo.pos, geo.cell[0], radius)
    d = Data(x=geo.x, pos=geo.pos, cell=geo.cell, edge_index=edge_index, cart_dist=cart_dist, cart_dir=cart_dir,
             natoms=geo.natoms)
    d.y = geo.y
    if adp:
        d.non_H_mask = geo.non_H_mask
        d.temperature = geo.temperature
    return d


def make_batch(n_graphs: int, n_atoms: Optional[int] = 194, first: int = 0, radius: float = 5.0,
               n_range=(64, 324), adp: bool = True, augment_seed: Optional[int] = None) -> Batch:
    """Collated batch of ``n_graphs`` synthetic crystals (graph ids first .. first+n_graphs-1)."""
    items = [make_crystal(first + g, n_atoms, radius, n_range, adp) for g in range(n_graphs)]
    if augment_seed is not None:
        gen = torch.Generator().manual_seed(augment_seed)
        items = [augment_data(d, gen) for d in items]
    return Batch.from_data_list(items)


def make_regular_batch(n_graphs: int, n_atoms: int, degree: int, seed: int = 0, adp: bool = True) -> Batch:
    """Cheap fixed-degree batch (every node has ``degree`` incoming edges from random nodes of its own graph,
    random unit directions, dist ~ U[1,5]) -- the shape used for the CPU numbers in BASELINE.md §2.  Used where
    building a periodic radius graph would dominate set-up time (large benchmark batches)."""
    gen = torch.Generator().manual_seed(seed)
    N = n_graphs * n_atoms
    tgt = torch.arange(N, dtype=torch.int64).repeat_interleave(degree)
    base = (tgt // n_atoms) * n_atoms
    src = base + torch.randint(0, n_atoms, (N * degree,), generator=gen)
    # within a target segment the reference orders edges by source (dataset/utils.py ordering)
    key = tgt * n_atoms + (src - base)
    order = torch.argsort(key, stable=True)
    src, tgt = src[order], tgt[order]
    E = N * degree
    v = torch.randn(E, 3, generator=gen)
    cart_dir = torch.nn.functional.normalize(v, dim=-1)
    cart_dist = 1.0 + 4.0 * torch.rand(E, generator=gen)
    u = torch.rand(N, generator=gen)
    other = torch.randint(9, 54, (N,), generator=gen)
    z = torch.where(u < 0.45, torch.tensor(1), torch.where(u < 0.80, torch.tensor(6), torch.where(
        u < 0.87, torch.tensor(7), torch.where(u < 0.97, torch.tensor(8), other)))).to(torch.int64)
    b = Batch(x=z, edge_index=torch.stack((src, tgt)), cart_dist=cart_dist, cart_dir=cart_dir)
    b.batch = torch.arange(n_graphs, dtype=torch.int64).repeat_interleave(n_atoms)
    b.ptr = torch.arange(0, N + 1, n_atoms, dtype=torch.int64)
    b.num_graphs = n_graphs
    b.cell = torch.eye(3).repeat(n_graphs, 1, 1) * 10.0
    if adp:
        mask = z != 1
        m = int(mask.sum().item())
        A = torch.randn(m, 3, 3, generator=gen)
        b.y = 0.01 * A @ A.transpose(1, 2) + 0.005 * torch.eye(3)
        b.non_H_mask = mask
        b.temperature = ((90.0 + 210.0 * torch.rand(n_graphs, generator=gen)) - TEMP_MEAN) / TEMP_STD
    else:
        b.y = torch.randn(n_graphs, generator=gen)
    return b
