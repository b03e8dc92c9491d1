"""How much of the DMA-fed fp32 kernels' distance to the matrix peak is the last, partly filled round of workgroups?
The plain two-group product [M, 256] x [256, 256] x 2 (cn_gemm_f32nn_kernel: 128 x 256 tiles, 2 workgroups per CU = 512 slots)
and the single-group one (cn_gemm_f32nn128_kernel: 128 x 128 tiles, 3 per CU = 768 slots) at row counts that fill whole rounds
and at the benchmark's E = 177,140 (5.41 / 3.6 rounds).  GPU box: python tools/experiments/exp_tile_rounds.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
D = 256
Ws = [(torch.randn(D, D, generator=g) * 0.05).to(dev) for _ in range(2)]
imgs = ops.pack_b(Ws)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M in (128 * 1280, 177140, 128 * 1536, 128 * 1408, 128 * 2560):
    x = torch.randn(M, 2 * D, generator=g).to(dev)
    out = torch.empty(M, 2 * D, device=dev)
    ms2 = timeit(lambda: ops.gemm([x[:, :D], x[:, D:]], Ws, [out[:, :D], out[:, D:]], b_kstrided=True, b_split=imgs))
    ms1 = timeit(lambda: ops.gemm(x[:, :D], Ws[0], out[:, :D], b_kstrided=True, b_split=imgs[:1]))
    t = (M + 127) // 128
    f2, f1 = 2.0 * M * D * D * 2, 2.0 * M * D * D
    print(f"M {M:7d}: two groups {2 * t:5d} tiles = {2 * t / 512:5.2f} rounds  {ms2 * 1e3:7.1f} us  {f2 / ms2 / 1e9 / 157.3:.3f} of peak |"
          f" one group (128-wide) {2 * t:5d} tiles = {2 * t / 768:5.2f} rounds  {ms1 * 1e3:7.1f} us  {f1 / ms1 / 1e9 / 157.3:.3f}", flush=True)
