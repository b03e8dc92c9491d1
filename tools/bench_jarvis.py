"""Step time on BASELINE configs[2] shapes: 64 crystals of 2..20 atoms, Scalar_head, no temperature."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.train import compute_loss
from cartnet_amd import train as ctrain
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
print("N", base.x.shape[0], "E", base.edge_index.shape[1], flush=True)
model = CartNet(256, 64, 4, temperature=False, cholesky=False).to(dev).train()
opt = FlatAdam(model, lr=1e-3)
opt.direct_grads = True
if "--no-overlap" in sys.argv:
    model.overlap_weight_gradients = False
print("weight-gradient stream:", model.overlap_weight_gradients, flush=True)
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs; b._cartnet_layout = None; b._cartnet_mask_index = None
    return b
def step(b):
    pred, true = model(b)
    loss = compute_loss(pred, true)[0]
    ctrain.backward(loss)
    opt.step(1.0); opt.zero_grad()
cases = ((0, False), (1, False), (2, False), (2, True))
if os.environ.get("JARVIS_ONLY"):          # e.g. JARVIS_ONLY=2,0: precision 2, fp32 storage only (tools/exp_small_batch_layer.sh)
    pr, hf = os.environ["JARVIS_ONLY"].split(",")
    cases = ((int(pr), hf == "1"),)
for prec, half in cases:
    model.gemm_precision = prec
    model.half_storage = half
    bs = [fresh() for _ in range(25)]
    for b in bs[:5]: step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in bs[5:]: step(b)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"precision {prec}{' + bf16 storage' if half else ''}: {1e3*dt/20:.3f} ms/step ({1e3*t_enq/20:.3f} ms host enqueue)  {64*20/dt:.0f} graphs/s", flush=True)
