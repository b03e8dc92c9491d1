import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops
dev = torch.device("cuda:0")
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
    print(f"{name:50s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s-equiv", flush=True)
def err(a, ref): return ((a.double()-ref).abs().max()/ref.abs().max()).item()
# accuracy on a small case vs fp64
M, K, N = 4096, 256, 256
A = rnd(M, K); Bt = rnd(K, N) * 0.1
ref = A.double() @ Bt.double()
for prec in (0, 1):
    C = torch.empty(M, N, device=dev)
    ops.gemm(A, Bt, C, b_kstrided=True, precision=prec)
    print("NN precision", prec, "rel err vs fp64:", err(C, ref))
ref2 = torch.nn.functional.silu(A.double()) @ Bt.double()
for prec in (0, 1):
    C = torch.empty(M, N, device=dev)
    ops.gemm(A, Bt, C, b_kstrided=True, a_act=True, precision=prec)
    print("NN+silu(A) precision", prec, "rel err vs fp64:", err(C, ref2))
E2 = 20000
dY, X = rnd(E2, 256), rnd(E2, 256)
ref3 = dY.double().t() @ X.double()
for prec in (0, 1):
    S = 16
    slabs = torch.empty(S * 256, 256, device=dev); out = torch.empty(256, 256, device=dev)
    ops.gemm(dY, X, slabs, a_kstrided=True, b_kstrided=True, splitk=S, precision=prec)
    ops.splitk_reduce(slabs, S, out)
    print("TN precision", prec, "rel err vs fp64:", err(out, ref3))
# speed at the layer shapes
E = 177140; D = 256
e = rnd(E, D); pre = rnd(E, 2*D); gs = rnd(E, 2*D)
W = [rnd(D, D) * 0.05 for _ in range(4)]
out2 = torch.empty(E, 2*D, device=dev)
F2 = 2.0*E*D*D*2
for prec in (0, 1):
    timeit(lambda: ops.gemm([e, e], [W[0], W[1]], [out2[:, :D], out2[:, D:]], b_kstrided=True, precision=prec), F2, f"NN x2 plain prec={prec}")
    timeit(lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W[2], W[3]], [out2[:, :D], out2[:, D:]], b_kstrided=True, a_act=True, precision=prec), F2, f"NN x2 silu(A) prec={prec}")
    S = 128
    slabs = [torch.empty(S*D, D, device=dev) for _ in range(2)]
    timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [e, e], slabs, a_kstrided=True, b_kstrided=True, splitk=S, precision=prec), F2, f"TN x2 splitk=128 prec={prec}")
