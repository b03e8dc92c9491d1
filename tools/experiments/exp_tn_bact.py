"""The weight-gradient product dW = dY^T (silu?)(X) of a layer's second Linears alone on the chip (two groups of [256 x 256] over
E rows, 64 split-K slabs): plain operand (the kept silu(pre) of rounds 2-5) against SiLU applied in place in LDS
(cn_gemm_f32tn_kernel<true, 5>, round 6).   (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from cartnet_amd import ops
dev = torch.device("cuda:0")
E, D, S = 177140, 256, 64
g = torch.Generator().manual_seed(0)
dY = torch.randn(E, 2 * D, generator=g).to(dev); X = torch.randn(E, 2 * D, generator=g).to(dev)
Xa = torch.nn.functional.silu(X)
slabs = [torch.empty(S * D, D, device=dev) for _ in range(2)]
def timeit(fn, warm=30, iters=60):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
def run(b_act, Xop):
    ops.gemm([dY[:, :D], dY[:, D:]], [Xop[:, :D], Xop[:, D:]], slabs, a_kstrided=True, b_kstrided=True, b_act=b_act, splitk=S)
run(True, X); a = [s.clone() for s in slabs]
run(False, Xa); b = slabs
err = max(float((x.view(S, D, D).sum(0) - y.view(S, D, D).sum(0)).abs().max()) for x, y in zip(a, b))
scale = float(b[0].view(S, D, D).sum(0).abs().max())
print(f"in-place SiLU vs plain operand on torch's silu: max|d| = {err:.3e} of {scale:.3e}")
F = 2.0 * E * D * D * 2
for rep in range(2):
    t0 = timeit(lambda: run(False, Xa)); t1 = timeit(lambda: run(True, X))
    print(f"plain {t0:7.1f} us ({F/t0/1e6/157.3:.3f})   SiLU in place {t1:7.1f} us ({F/t1/1e6/157.3:.3f})", flush=True)
