import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.config import cfg
from cartnet_amd.model import CartNet
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_batch
cfg.radius = 5.0
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CartNet(dim_in=256, dim_rbf=64, num_layers=4).to(dev).train()
opt = FlatAdam(model, lr=1e-3)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = make_batch(G, 194, first=100000).to(dev)
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs; b._cartnet_layout = None; b._cartnet_mask_index = None
    return b
def step(b):
    pred, true = model(b)
    loss = (pred - true).abs().mean()
    loss.backward()
    opt.step(1.0); opt.zero_grad()
for prec in (0, 1):
    for ov in (True, False):
        model.gemm_precision = prec
        model.overlap_weight_gradients = ov
        bs = [fresh() for _ in range(13)]
        for b in bs[:3]: step(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in bs[3:]: step(b)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"precision {prec} overlap {ov}: {1e3*dt/10:.3f} ms/step  {G*10/dt:.1f} graphs/s", flush=True)
