"""Print per-parameter gradient errors of the HIP path against a golden fixture (GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import golden_utils as gu
from test_gpu_model import _model
def _grad_err(g, ref, gmax):
    g, ref = g.detach().double().cpu(), ref.double()
    return (g - ref).abs().max().item() / gmax
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_adp"
z, hp, b, sd = gu.load(name)
m = _model(hp, sd).train()
bb = gu.clone_batch(b).to("cuda:0")
pred, true = m(bb)
loss = (pred - true).abs().mean()
loss.backward()
params = dict(m.named_parameters())
gmax = max(float(np.abs(z["grad64_" + k]).max()) for k in params)
print("loss", loss.item(), float(z["train_mae"]), "gmax", gmax)
for k, p in params.items():
    ref = torch.from_numpy(z["grad64_" + k])
    g = p.grad
    if g is None:
        print(f"{k:45s} NONE"); continue
    print(f"{k:45s} err={_grad_err(g, ref, gmax):.3e}  |g|={g.abs().max().item():.3e} |ref|={ref.abs().max().item():.3e}")
