"""Inference pass (eval mode, no_grad) of the benchmark batch, for timing / profiling: python tools/eval_forward.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.config import cfg
from cartnet_amd.model import CartNet
from cartnet_amd.synthetic import make_batch
cfg.radius = 5.0
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CartNet(256, 64, 4).to(dev).eval()
model.gemm_precision = int(os.environ.get("PREC", "0"))
base = make_batch(64, 194, first=100000).to(dev)
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs
    return b
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(3): model(fresh())
    bs = [fresh() for _ in range(n)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in bs: model(b)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"eval forward: {1e3 * dt / n:.3f} ms per batch of 64 ({64 * n / dt:.0f} graphs/s)")
