"""Where the host time of an iComformer training step goes (cProfile over 5 steps, GPU box)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd.config import cfg
from cartnet_amd.comformer import iComformer
from cartnet_amd.synthetic import make_batch
cfg.radius = 5.0
dev = torch.device("cuda:0")
model = iComformer(256).to(dev).train()
model.gemm_precision = int(sys.argv[1]) if len(sys.argv) > 1 else 1
base = make_batch(64, 194, first=100000).to(dev)
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs
    return b
def step(b):
    pred, true = model(b)
    (pred - true).abs().mean().backward()
    for p in model.parameters(): p.grad = None
bs = [fresh() for _ in range(8)]
for b in bs[:3]: step(b)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for b in bs[3:]: step(b)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
