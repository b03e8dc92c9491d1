import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops, lib as _l
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
def timeit(fn, flops, name, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:56s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s-equiv", flush=True)
L = _l.load()
E = 177140; D = 256
e = rnd(E, D)
Wl = [rnd(D, D) * 0.05 for _ in range(2)]
Wt = [w.t().contiguous() for w in Wl]
im = ops.split_b([w.t() for w in Wl])
out2 = torch.empty(E, 2 * D, device=dev)
F2 = 2.0 * E * D * D * 2
for d in (0, 0):
    out2.zero_()
    timeit(lambda: ops.gemm([e, e], Wt, [out2[:, :D], out2[:, D:]], b_kstrided=True, precision=1, b_split=im), F2, f"NN x2 plain x3v2 diag={d}")
    ref = e[:4096].double() @ torch.cat(Wl, 0).double().t()
    print("   err", ((out2[:4096].double() - ref).abs().max() / ref.abs().max()).item(), "tail err",
          ((out2[-300:].double() - e[-300:].double() @ torch.cat(Wl, 0).double().t()).abs().max() / ref.abs().max()).item())
