"""Micro-benchmark of the fp32 MFMA GEMM variants at the benchmark's shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
E = int(os.environ.get("E", 177140)); D = 256
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)

def timeit(fn, flops, name, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:46s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s  ({100*flops/ms/1e9/157.3:.1f}% of fp32 MFMA peak)")

e = rnd(E, D); pre = rnd(E, 2*D); gs = rnd(E, 2*D)
W1g, W1a = rnd(D, 3*D)*0.05, rnd(D, 3*D)*0.05
W2g, W2a = rnd(D, D)*0.05, rnd(D, D)*0.05
N = E // 14
Pn = rnd(N, 4*D)
tgt = torch.sort(torch.randint(0, N, (E,), generator=g)).values.to(torch.int32).to(dev)
src = torch.randint(0, N, (E,), generator=g).to(torch.int32).to(dev)
out2 = torch.empty(E, 2*D, device=dev); out1 = torch.empty(E, D, device=dev)
tiles = ops.gemm_tiles_m(E)
cs = torch.empty(tiles*D, dtype=torch.float64, device=dev); cq = torch.empty_like(cs)
F2 = 2.0*E*D*D*2

timeit(lambda: ops.gemm([e, e], [W1g[:, 2*D:], W1a[:, 2*D:]], [out2[:, :D], out2[:, D:]]), F2, "NT x2 plain (K=256)")
timeit(lambda: ops.gemm([e, e], [W1g[:, 2*D:], W1a[:, 2*D:]], [out2[:, :D], out2[:, D:]],
                        gather_i=[Pn[:, :D], Pn[:, D:2*D]], gather_j=[Pn[:, 2*D:3*D], Pn[:, 3*D:]], tgt=tgt, src=src), F2, "NT x2 + gather epilogue (layer GEMM1)")
timeit(lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W2g, W2a], [out2[:, :D], out2[:, D:]], a_act=True,
                        colsum=[cs, None], colsq=[cq, None]), F2, "NT x2 a_act + stats (layer GEMM2)")
timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [out2[:, :D], out2[:, D:]], b_kstrided=True), F2, "NN x2 plain")
timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [out2[:, :D], out2[:, D:]], b_kstrided=True,
                        dact=[pre[:, :D], pre[:, D:]], colsum=[cs, cq]), F2, "NN x2 + dact + colsum (dpre)")
timeit(lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W1g[:, 2*D:], W1a[:, 2*D:]], out1, b_kstrided=True, segments=True,
                        resid=e), F2, "NN 2 segments + resid (de_in)")
S = 128
slabs = [torch.empty(S*D, D, device=dev) for _ in range(2)]
timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [pre[:, :D], pre[:, D:]], slabs, a_kstrided=True, b_kstrided=True,
                        b_act=True, splitk=S), F2, "TN x2 splitk=128 b_act (dW2)")
timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [e, e], slabs, a_kstrided=True, b_kstrided=True, splitk=S), F2, "TN x2 splitk=128 (dW1)")
