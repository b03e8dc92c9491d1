"""cProfile INSIDE iComformer's backward (it runs on autograd's device thread, which a profile of the main thread never
sees): where the host time of `_IComformerFunction.backward` goes.  GPU box."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd.config import cfg
from cartnet_amd import comformer
from cartnet_amd.comformer import iComformer
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_batch
cfg.radius = 5.0
dev = torch.device("cuda:0")
model = iComformer(256).to(dev).train()
model.gemm_precision = int(sys.argv[1]) if len(sys.argv) > 1 else 0
opt = FlatAdam(model, lr=1e-3)
base = make_batch(64, 194, first=100000).to(dev)
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs
    return b
pr = cProfile.Profile()
orig = comformer._IComformerFunction.backward
state = {"on": False}
def wrapped(ctx, *a):
    if state["on"]:
        pr.enable()
        try:
            return orig(ctx, *a)
        finally:
            pr.disable()
    return orig(ctx, *a)
comformer._IComformerFunction.backward = staticmethod(wrapped)
def step(b):
    pred, true = model(b)
    (pred - true).abs().mean().backward()
    opt.step(1.0); opt.zero_grad()
bs = [fresh() for _ in range(8)]
for b in bs[:3]: step(b)
torch.cuda.synchronize()
state["on"] = True
for b in bs[3:]: step(b)
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue())
