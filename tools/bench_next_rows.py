"""Measurement of the SURVEY.md §8(f) rows on either side of the hot path, at the benchmark batch (64 ADP-shaped
crystals x 194 atoms): one JSON line per row with the GPU time, the algorithmic work behind it, the bound it is priced
against and the host-CPU time of the same step (this build's own CPU code paths: the CPU radius graph of
cartnet_amd/synthetic.py and the host collation of cartnet_amd/data.py).

  (f)1  periodic radius graph + neighbour cap     cartnet_amd/graph.py      (reference: dataset/utils.py:57-360)
  (f)2  ADP evaluation metrics                    cartnet_amd/metrics.py    (reference: train/metrics.py:42-180)
  (f)3  packed shard -> device collate (+ SO(3))  cartnet_amd/shard.py      (reference: PyG Batch collation in the loader)

Run on the GPU box:  python tools/bench_next_rows.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch

from cartnet_amd import graph, metrics
from cartnet_amd.data import Batch
from cartnet_amd.shard import DeviceShard, random_rotations
from cartnet_amd.synthetic import make_crystal, radius_graph_pbc_single

dev = torch.device("cuda:0")
G, ATOMS = 64, 194
HBM_PEAK = 8.0e12
VALU_FP32_PEAK = 78.6e12      # 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2.4 GHz, unpacked fp32


_spin = torch.empty(64 << 20, device=dev)


def gpu_time(fn, iters=20, warm=3):
    for _ in range(200):       # ~50 ms of HBM traffic first: after host-side phases the chip sits in a low clock state
        _spin.add_(1.0)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


items = [make_crystal(100_000 + g, ATOMS) for g in range(G)]
host = Batch.from_data_list(items)
N, E = int(host.x.shape[0]), int(host.edge_index.shape[1])

# ---------------------------------------------------------------------------------------------- (f)1 radius graph
pos, cell, ptr = host.pos.to(dev), host.cell.to(dev), host.ptr.to(dev)
t_gpu = gpu_time(lambda: graph.radius_graph_pbc(pos, cell, ptr, 5.0))
t_cap = gpu_time(lambda: graph.radius_graph_pbc(pos, cell, ptr, 5.0, max_neighbors=12))
t0 = time.perf_counter()
for it in items[:8]:
    radius_graph_pbc_single(it.pos, it.cell[0], 5.0)
t_cpu = (time.perf_counter() - t0) / 8 * G
out_bytes = E * (16 + 4 + 12)                 # edge_index int64 x2, cart_dist, cart_dir
print(json.dumps({"row": "(f)1 periodic radius graph", "workload": f"{G} crystals x {ATOMS} atoms, r = 5 A, E = {E}",
                  "gpu_ms": round(1e3 * t_gpu, 3), "gpu_ms_with_cap_12": round(1e3 * t_cap, 3),
                  "crystals_per_s": round(G / t_gpu), "edges_per_s": round(E / t_gpu),
                  "bound": "VALU (n^2 pairs per crystal x the images inside the pair's image box, ~6 instead of 27 distance tests per pair; two passes: count, fill) + 2 host syncs "
                           "for the output sizes", "output_bytes": out_bytes,
                  "output_GBps": round(out_bytes / t_gpu / 1e9, 1),
                  "cpu_ms": round(1e3 * t_cpu, 1), "cpu": "cartnet_amd.synthetic.radius_graph_pbc_single, torch CPU "
                  f"({torch.get_num_threads()} threads), 8 crystals timed and scaled to {G}",
                  "speedup": round(t_cpu / t_gpu, 1),
                  "share_of_train_step": "0.55 ms of a 15.5 ms step if graphs were built on the fly"}), flush=True)

# ---------------------------------------------------------------------------------------------- (f)2 ADP metrics
M = int(host.y.shape[0])
gen = torch.Generator().manual_seed(0)
A = torch.randn(M, 3, 3, generator=gen)
pred = (host.y + 0.002 * (A @ A.transpose(1, 2))).to(dev)
true = host.y.to(dev)
t_all = gpu_time(lambda: metrics.adp_metrics(pred, true))
t_iou = gpu_time(lambda: metrics.compute_3D_IoU(pred, true))
t_small = gpu_time(lambda: metrics.adp_metrics(pred, true, True, True, False))
vox = 64 ** 3
# per voxel and ellipsoid 14 flops (csrc/metrics.hip: three fma for the z terms of p @ S, three products, two adds,
# compares and counters), two ellipsoids; the x / y terms are shared by a z column
flops = M * vox * 2 * 14
print(json.dumps({"row": "(f)2 ADP evaluation metrics", "workload": f"M = {M} non-H atoms, 64^3 voxels per atom",
                  "gpu_ms_all_three": round(1e3 * t_all, 3), "gpu_ms_iou_only": round(1e3 * t_iou, 3),
                  "gpu_ms_volume_and_similarity": round(1e3 * t_small, 3), "atoms_per_s": round(M / t_all),
                  "bound": "VALU fp32 (voxel classification in registers; the reference's [M, 262144] maps are never "
                           "materialised: 2 x 13 GB of fp32 at this M)",
                  "algorithmic_flops": flops, "achieved_TFLOPs": round(flops / t_iou / 1e12, 2),
                  "peak_TFLOPs": VALU_FP32_PEAK / 1e12, "frac": round(flops / t_iou / VALU_FP32_PEAK, 3),
                  "cpu": "not timed here: tools may not import the oracle (tests/test_gpu_metrics.py checks the kernel against "
                         "oracle/metrics_ref.py and the reference's golden outputs)"}),
      flush=True)

# ---------------------------------------------------------------------------------------------- (f)3 collate
shard = DeviceShard.from_data_list(items, device=dev)
sel = list(range(G))
rot = random_rotations(G, torch.Generator(device=dev).manual_seed(1), dev)
t_col = gpu_time(lambda: shard.collate(sel))
t_aug = gpu_time(lambda: shard.collate(sel, rot=rot))
Batch.from_data_list(items)                  # warm the allocator
t0 = time.perf_counter()
for _ in range(5):
    Batch.from_data_list(items)
t_host = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    Batch.from_data_list(items).to(dev)
torch.cuda.synchronize()
t_host_dev = (time.perf_counter() - t0) / 5
# bytes one collated batch occupies (read once from the shard, written once): x, batch, edge_index, dist, dir, mask, y
b = shard.collate(sel)
moved = sum(int(t.numel()) * t.element_size() for t in vars(b).values() if torch.is_tensor(t))
print(json.dumps({"row": "(f)3 packed shard -> device collate", "workload": f"{G} crystals, N = {N}, E = {E}",
                  "gpu_us": round(1e6 * t_col, 1), "gpu_us_with_so3_augmentation": round(1e6 * t_aug, 1),
                  "bound": "HBM (read the packed arrays, write the batch) -- at this size launch latency, not "
                           "bandwidth: one launch + the output allocations",
                  "batch_bytes": moved, "algorithmic_GBps": round(2 * moved / t_col / 1e9, 1),
                  "frac_of_hbm_peak": round(2 * moved / t_col / HBM_PEAK, 4),
                  "cpu_ms_host_collation": round(1e3 * t_host, 2),
                  "cpu_ms_host_collation_plus_upload": round(1e3 * t_host_dev, 2),
                  "cpu": "cartnet_amd.data.Batch.from_data_list (+ .to(device))",
                  "speedup_vs_host_path": round(t_host_dev / t_col, 1)}), flush=True)
