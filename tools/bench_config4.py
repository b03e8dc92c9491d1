"""One rank's share of BASELINE.json configs[3] on one MI355X: 162,270 / 8 = 20,284 ragged ADP-shaped crystals
(64..324 atoms), edges built by the GPU radius-graph builder (cartnet_amd.graph), resident in HBM as a
packed shard, every batch collated and SO(3)-augmented on the device, one training epoch per recipe:

    batch 64 x accumulation 1          (the bench workload's batching)
    batch 4 x accumulation 16          (the reference recipe, scripts/train_cartnet_adp.sh:4, train/train.py:183-189)
    [batch 64 as 16 groups of 4]       (--grouped: reference-recipe BatchNorm / loss semantics in one pass per step)

Two epochs per recipe: the first warms the allocator up, the second (>= 5 s) is the record.  Prints one JSON object; profiles/r02_config4_share.json is a copy of it.
usage: python tools/bench_config4.py [--crystals 20284] [--precision 0] [--grouped]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from cartnet_amd.config import cfg
from cartnet_amd.model import CartNet
from cartnet_amd.optim import FlatAdam
from cartnet_amd.shard import DeviceShard, ShardLoader, pack_with_gpu_graph
from cartnet_amd.synthetic import make_geometry
from cartnet_amd.train import train_epoch

ap = argparse.ArgumentParser()
ap.add_argument("--crystals", type=int, default=162270 // 8)
ap.add_argument("--precision", type=int, default=0)
ap.add_argument("--chunk", type=int, default=256, help="crystals per radius-graph launch")
ap.add_argument("--grouped", action="store_true", help="also run batch 64 with BatchNorm / loss per group of 4")
ap.add_argument("--only-batch64", action="store_true", help="skip the literal micro-batch recipe (full-epoch runs)")
args = ap.parse_args()
cfg.radius = 5.0
dev = torch.device("cuda:0")
n = args.crystals

t0 = time.perf_counter()
geo = []
for i in range(n):            # host-side geometry, single process (a progress line every 10k crystals keeps the run visible)
    geo.append(make_geometry(30000 + i, None))
    if (i + 1) % 10000 == 0:
        print(f"geometry {i + 1}/{n} {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
t_geo = time.perf_counter() - t0
# ---- edges on the GPU (cartnet_amd.shard.pack_with_gpu_graph: `chunk` crystals per launch pair, rebased to the crystal)
torch.cuda.synchronize()
t0 = time.perf_counter()
arrays = pack_with_gpu_graph(geo, 5.0, dev, args.chunk)
torch.cuda.synchronize()
t_graph = time.perf_counter() - t0
atom_ptr = arrays["atom_ptr"]
del geo
shard = DeviceShard(arrays, dev)
print(f"graph built in {t_graph:.1f} s, shard resident", file=sys.stderr, flush=True)
what = "the whole epoch of BASELINE configs[3] on ONE GPU" if n >= 162270 else "one rank's share of BASELINE configs[3]"
out = {"workload": f"{what}: {n} synthetic ADP crystals of 64..324 atoms "
                   f"({int(atom_ptr[-1])} atoms, {int(arrays['edge_ptr'][-1])} edges), SO(3) augmentation on, CartNet L=4 "
                   f"D=256 fp32 storage, gemm_precision={args.precision}, 1x MI355X",
       "host_geometry_seconds": round(t_geo, 2), "gpu_radius_graph_seconds": round(t_graph, 2),
       "shard_bytes_in_hbm": shard.nbytes(), "recipes": []}

recipes = [("batch 64 x accumulation 1", 64, 1, 0)]
if not args.only_batch64:
    recipes.append(("batch 4 x accumulation 16 (reference recipe)", 4, 16, 0))
if args.grouped:
    recipes.append(("batch 64 as 16 BatchNorm/loss groups of 4 (reference-recipe semantics, one pass)", 64, 1, 4))
for name, bs, accum, group in recipes:
    torch.manual_seed(0)
    model = CartNet(256, 64, 4).to(dev).train()
    model.gemm_precision = args.precision
    if group:
        model.bn_group_size = group
    opt = FlatAdam(model, lr=1e-3)
    # epoch 0 untimed (the caching allocator grows to the largest ragged batch, kernels and weight images warm up),
    # epoch 1 is the record
    loader = ShardLoader(shard, bs, shuffle=True, seed=0, augment=True)
    r0 = train_epoch(loader, model, opt, accum, None, device=dev)
    print(f"warm-up epoch done in {r0['seconds']:.1f} s", file=sys.stderr, flush=True)
    loader = ShardLoader(shard, bs, shuffle=True, seed=1, augment=True)
    r = train_epoch(loader, model, opt, accum, None, device=dev)
    out["recipes"].append({"recipe": name, "graphs": r["graphs"], "seconds": round(r["seconds"], 3),
                           "graphs_per_s": round(r["graphs"] / r["seconds"], 1), "train_mae": round(r["mae"], 5),
                           "first_epoch_seconds_untimed_warmup": round(r0["seconds"], 3),
                           "epoch_of_162270_on_8_ranks_seconds": round(r["seconds"] * 162270 / 8 / r["graphs"], 2)})
    print(json.dumps(out["recipes"][-1]), file=sys.stderr, flush=True)
print(json.dumps(out))
