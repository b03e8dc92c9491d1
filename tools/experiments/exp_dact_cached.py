"""Experiment (GPU box): is the cost of the silu'(pre) epilogue (dpre product: 444 vs 371 us for the plain product) HBM
traffic or latency / VALU work?  Same launch with the `dact` operand's row stride set to 0, so that every row reads the
same 2 KB (L2 hits, no HBM traffic) but the epilogue executes the same loads and the same arithmetic."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
E, D = 177140, 256
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
gs, pre = rnd(E, 2 * D), rnd(E, 2 * D)
W = [rnd(D, D) * 0.05 for _ in range(2)]
img = ops.pack_b(W)
out2 = torch.empty(E, 2 * D, device=dev)


def sustained(fn, seconds=1.5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    return (time.perf_counter() - t0) / n * 1e6


def run(dact):
    ops.gemm([gs[:, :D], gs[:, D:]], W, [out2[:, :D], out2[:, D:]], b_kstrided=True, b_split=img, dact=dact)


print(f"plain                      {sustained(lambda: run(None)):7.1f} us")
print(f"dact from HBM              {sustained(lambda: run([pre[:, :D], pre[:, D:]])):7.1f} us")
ops._f32_2d = lambda *a, **k: None          # experiment only: let a stride-0 operand through the wrapper's checks
row = pre[:1].expand(E, 2 * D)
print(f"dact, every row the same   {sustained(lambda: run([row[:, :D], row[:, D:]])):7.1f} us")
