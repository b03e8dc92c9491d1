"""Epoch-style throughput (BASELINE configs[3] on one GPU): variable-size ADP crystals (64..324 atoms) resident in HBM
as a packed shard, every batch collated + SO(3)-augmented on the GPU, train_epoch over all of them."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.config import cfg
from cartnet_amd.model import CartNet
from cartnet_amd.optim import FlatAdam
from cartnet_amd.shard import DeviceShard, ShardLoader
from cartnet_amd.synthetic import make_crystal
from cartnet_amd.train import train_epoch
cfg.radius = 5.0
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = time.perf_counter()
items = [make_crystal(20000 + i, None) for i in range(n)]      # n_atoms uniform in [64, 324]
print(f"built {n} crystals on the host in {time.perf_counter()-t0:.1f} s; atoms {sum(int(d.x.shape[0]) for d in items)}, "
      f"edges {sum(int(d.edge_index.shape[1]) for d in items)}", flush=True)
shard = DeviceShard.from_data_list(items, dev)
loader = ShardLoader(shard, bs, shuffle=True, seed=0, augment=True)
model = CartNet(256, 64, 4).to(dev).train()
model.gemm_precision = prec
opt = FlatAdam(model, lr=1e-3)
for ep in range(3):
    r = train_epoch(loader, model, opt, 1, None, device=dev)
    print(f"epoch {ep}: {r['graphs']} graphs in {r['seconds']:.3f} s = {r['graphs']/r['seconds']:.0f} graphs/s  mae {r['mae']:.4f}", flush=True)
