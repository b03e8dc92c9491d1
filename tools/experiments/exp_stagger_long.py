"""Experiment (round 3): is the un-hidden fixed cost of a GEMM launch a CHIP-WIDE burst of epilogue stores?  A very long
launch (10x the benchmark's rows, ~54 rounds of tiles: start-up and tail are negligible) of the plain two-group product,
with the library as built (env CARTNET lib A/B chosen by the caller)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops
dev = torch.device("cuda:0")
D = 256
g = torch.Generator().manual_seed(0)
W = [(torch.randn(D, D, generator=g) * 0.05).to(dev) for _ in range(2)]
img = ops.pack_b(W)
for mult in (1, 4, 10):
    E = 177140 * mult
    h = torch.randn(E, 2 * D, device=dev)
    out = torch.empty(E, 2 * D, device=dev)
    fn = lambda: ops.gemm([h[:, :D], h[:, D:]], W, [out[:, :D], out[:, D:]], b_kstrided=True, b_split=img, precision=0)
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 20
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    fl = 2.0 * E * D * D * 2
    print(f"rows x{mult:2d}: {us:9.1f} us per launch  ({us / mult:7.1f} us per 177k rows)  {fl / us / 1e6:6.1f} TF/s  {fl / us / 1e6 / 157.3:.3f} of peak", flush=True)
    del h, out
