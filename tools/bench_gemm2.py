import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timeit(fn, flops, name, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:46s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s  ({100*flops/ms/1e9/157.3:.1f}%)", flush=True)
for M in (65536, 177140):
  for K in (256, 1024, 4096):
    A = torch.randn(M, K, device=dev); B = torch.randn(256, K, device=dev) * 0.05; Cc = torch.empty(M, 256, device=dev)
    timeit(lambda: ops.gemm(A, B, Cc), 2.0*M*256*K, f"NT M={M} N=256 K={K}")
    Bt = torch.randn(K, 256, device=dev) * 0.05
    timeit(lambda: ops.gemm(A, Bt, Cc, b_kstrided=True), 2.0*M*256*K, f"NN M={M} N=256 K={K}")
    del A
