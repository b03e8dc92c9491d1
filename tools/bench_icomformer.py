"""graphs/sec of the iComformer (default) / eComformer (4th argument "e") path (BASELINE configs[4]): forward + MAE + backward on ADP-shaped crystals."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.train import compute_loss
from cartnet_amd.config import cfg
from cartnet_amd.comformer import eComformer, iComformer
from cartnet_amd.synthetic import make_batch
cfg.radius = 5.0
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
kind = sys.argv[4] if len(sys.argv) > 4 else "i"
model = (eComformer if kind == "e" else iComformer)(256).to(dev).train()
model.gemm_precision = int(sys.argv[3]) if len(sys.argv) > 3 else 0
base = make_batch(G, 194, first=100000).to(dev)
print("N", base.x.shape[0], "E", base.edge_index.shape[1], flush=True)
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs
    return b
def step(b):
    pred, true = model(b)
    loss = compute_loss(pred, true)[0]
    loss.backward()
    for p in model.parameters(): p.grad = None
    return loss
bs = [fresh() for _ in range(2 + steps)]
for b in bs[:2]: step(b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in bs[2:]: l = step(b)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{kind}Comformer D=256 precision {model.gemm_precision} {G} crystals/step: {1e3*dt/steps:.2f} ms/step  {G*steps/dt:.1f} graphs/s  loss {l.item():.4f}  "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
