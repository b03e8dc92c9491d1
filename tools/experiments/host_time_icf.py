"""Host enqueue time of ONE iComformer training step on an idle queue (the bench line's host_enqueue_ms_per_step is
taken over 20 back-to-back steps and includes the time the host waits for room in the device queue when it runs ahead):
native C++ sequence (csrc/icomformer.hip) against the Python sequence.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd.comformer import iComformer
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_batch
from cartnet_amd.train import compute_loss
from cartnet_amd import train as ctrain
from cartnet_amd.config import cfg
cfg.radius = 5.0
dev = torch.device("cuda:0")
base = make_batch(64, 194, first=100_000).to(dev)
for native in (True, False):
    torch.manual_seed(0)
    m = iComformer(256).to(dev).train()
    m.native_sequence = native
    opt = FlatAdam(m, lr=1e-3)
    def fresh():
        b = base.clone(); b.num_graphs = base.num_graphs; return b
    def step(b):
        pred, true = m(b)
        loss = compute_loss(pred, true)[0]
        ctrain.backward(loss)
        opt.step(1.0); opt.zero_grad()
    bs = [fresh() for _ in range(12)]
    for b in bs[:4]: step(b)
    hs, ts = [], []
    for b in bs[4:]:
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(b); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(1e3 * (t1 - t0)); ts.append(1e3 * (t2 - t0))
    hs.sort(); ts.sort()
    print(f"{'native C++' if native else 'Python    '} sequence: host enqueue {hs[len(hs)//2]:.2f} ms per step (idle queue), "
          f"step {ts[len(ts)//2]:.2f} ms", flush=True)
    del m, opt
    torch.cuda.empty_cache()
