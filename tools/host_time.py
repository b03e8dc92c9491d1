"""Host-side cost of one training step at configs[2] shapes (cProfile of the Python / ctypes enqueue path; GPU box)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.train import compute_loss
from cartnet_amd.config import cfg
from cartnet_amd.data import Batch
from cartnet_amd.model import CartNet
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_crystal
cfg.radius = 5.0
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)
sizes = torch.randint(2, 21, (64,), generator=gen).tolist()
base = Batch.from_data_list([make_crystal(5000 + i, n, adp=False) for i, n in enumerate(sizes)]).to(dev)
model = CartNet(256, 64, 4, temperature=False, cholesky=False).to(dev).train()
model.gemm_precision = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = FlatAdam(model, lr=1e-3)
opt.direct_grads = True
if len(sys.argv) > 2:
    model.half_storage = bool(int(sys.argv[2]))
from cartnet_amd import train as ctrain
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs; b._cartnet_layout = None; b._cartnet_mask_index = None
    return b
T = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0}
def step(b):
    t0 = time.perf_counter(); pred, true = model(b)
    t1 = time.perf_counter(); loss = compute_loss(pred, true)[0]
    t2 = time.perf_counter(); ctrain.backward(loss)
    t3 = time.perf_counter(); opt.step(1.0); opt.zero_grad()
    t4 = time.perf_counter()
    T["fwd"] += t1 - t0; T["loss"] += t2 - t1; T["bwd"] += t3 - t2; T["opt"] += t4 - t3
bs = [fresh() for _ in range(45)]
for b in bs[:5]: step(b)
torch.cuda.synchronize()
for k in T: T[k] = 0.0
for b in bs[5:25]: step(b)
torch.cuda.synchronize()
print({k: round(1e3 * v / 20, 3) for k, v in T.items()})
pr = cProfile.Profile(); pr.enable()
for b in bs[25:45]: step(b)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
