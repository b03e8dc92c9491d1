import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd.model import make_state_dict
from cartnet_amd.synthetic import make_batch
from oracle import cartnet_ref as orc
batch = make_batch(4, 194, first=10_000)
sd = make_state_dict(256, 64, 4, seed=0)
params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "rbf" not in k else v.clone()) for k, v in sd.items()}
def step():
    for v in params.values():
        if v.requires_grad: v.grad = None
    pred = orc.cartnet_forward(params, batch, num_layers=4, training=True)
    (pred - batch.y).abs().mean().backward()
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    step(); step()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(nt, "threads:", round(4 / ts[len(ts)//2], 2), "graphs/s", flush=True)
