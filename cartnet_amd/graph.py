"""Periodic radius graph on the GPU: the step right before the hot path.

``radius_graph_pbc(pos, cell, ptr, radius, max_neighbors)`` builds the edges of a whole batch of crystals on the device
in the reference's edge order (reference: dataset/utils.py:57-237 as called by dataset/figshare_dataset.py:65-68), so
the result can be fed straight to ``CartNet.forward`` / ``iComformer.forward``; integers match the reference bit for
bit, distances / directions to fp32 rounding (tests/test_gpu_radius_graph.py).  ``max_neighbors`` is the reference's
neighbour cap (dataset/utils.py:240-360; 25 for iComformer, off for CartNet -- main.py:141,176)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import lib as _l


DEGENERACY_TOLERANCE = 0.01      # dataset/utils.py:245, in squared-distance units


def radius_graph_pbc(pos: torch.Tensor, cell: torch.Tensor, ptr: torch.Tensor, radius: float = 5.0,
                     max_neighbors: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """pos [N,3] fp32, cell [Bg,3,3] fp32 (rows = lattice vectors), ptr [Bg+1] int64 atom offsets -- all on the GPU.
    ``max_neighbors`` None or <= 0: no cap (what figshare_dataset.py:18 maps -1 to).
    Returns (edge_index [2,E] int64 = (source, target), cart_dist [E], cart_dir [E,3])."""
    if not (pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 2 and pos.shape[1] == 3):
        raise ValueError("pos must be a CUDA fp32 tensor [N,3]")
    if not (cell.is_cuda and cell.dtype == torch.float32 and cell.dim() == 3 and tuple(cell.shape[1:]) == (3, 3)):
        raise ValueError("cell must be a CUDA fp32 tensor [Bg,3,3]")
    if not (ptr.is_cuda and ptr.dtype == torch.int64 and ptr.dim() == 1 and ptr.numel() == cell.shape[0] + 1):
        raise ValueError("ptr must be a CUDA int64 tensor [Bg+1]")
    lib = _l.load()
    dev = pos.device
    pos, cell, ptr = pos.contiguous(), cell.contiguous(), ptr.contiguous()
    N, Bg = int(pos.shape[0]), int(cell.shape[0])
    batch = torch.repeat_interleave(torch.arange(Bg, device=dev), ptr[1:] - ptr[:-1]).contiguous()
    if batch.numel() != N:
        raise ValueError("ptr does not cover all atoms")
    reps = torch.empty(Bg * 15, dtype=torch.int32, device=dev)      # int32 [Bg,3] repetitions + fp32 [Bg,12] reciprocal lattice
    deg = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)
    _l.check(lib.cartnet_radius_graph_count(pos.data_ptr(), cell.data_ptr(), ptr.data_ptr(), batch.data_ptr(), N, Bg,
                                            float(radius), reps.data_ptr(), deg.data_ptr(), _l.stream_ptr()),
             "cartnet_radius_graph_count")
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg[:N].long(), 0)
    E = int(rowptr[-1].item())                       # one sync: the edge count sizes the outputs
    edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
    dist = torch.empty(E, dtype=torch.float32, device=dev)
    dirs = torch.empty((E, 3), dtype=torch.float32, device=dev)
    cap = max_neighbors is not None and max_neighbors > 0 and E > 0
    dist_sq = torch.empty(E, dtype=torch.float32, device=dev) if cap else None
    _l.check(lib.cartnet_radius_graph_fill(pos.data_ptr(), cell.data_ptr(), ptr.data_ptr(), batch.data_ptr(),
                                           reps.data_ptr(), rowptr.data_ptr(), N, Bg, float(radius), E,
                                           edge_index.data_ptr(), dist.data_ptr(), dirs.data_ptr(),
                                           dist_sq.data_ptr() if cap else None, _l.stream_ptr()),
             "cartnet_radius_graph_fill")
    if not cap or int(deg.max().item()) <= max_neighbors:        # dataset/utils.py:283-290: nothing to drop
        return edge_index, dist, dirs
    cutoff = torch.empty(N, dtype=torch.float32, device=dev)
    _l.check(lib.cartnet_neighbor_cap_count(rowptr.data_ptr(), dist_sq.data_ptr(), N, int(max_neighbors),
                                            DEGENERACY_TOLERANCE, cutoff.data_ptr(), deg.data_ptr(), _l.stream_ptr()),
             "cartnet_neighbor_cap_count")
    rowptr_out = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    rowptr_out[1:] = torch.cumsum(deg[:N].long(), 0)
    E_out = int(rowptr_out[-1].item())
    ei_out = torch.empty((2, E_out), dtype=torch.int64, device=dev)
    dist_out = torch.empty(E_out, dtype=torch.float32, device=dev)
    dirs_out = torch.empty((E_out, 3), dtype=torch.float32, device=dev)
    _l.check(lib.cartnet_neighbor_cap_fill(rowptr.data_ptr(), rowptr_out.data_ptr(), cutoff.data_ptr(),
                                           dist_sq.data_ptr(), edge_index.data_ptr(), dist.data_ptr(), dirs.data_ptr(),
                                           N, E, E_out, ei_out.data_ptr(), dist_out.data_ptr(), dirs_out.data_ptr(),
                                           _l.stream_ptr()), "cartnet_neighbor_cap_fill")
    return ei_out, dist_out, dirs_out
