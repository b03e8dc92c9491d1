"""Correctness (vs fp64) and speed of the second-generation bf16x3 kernels at the CartNet layer shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(*s, generator=g).to(dev)


def timeit(fn, flops, name, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:56s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s-equiv", flush=True)


def err(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


# ---------------- accuracy
M, K, N = 1000, 256, 256          # ragged last row tile
A = rnd(M, K)
W = rnd(N, K) * 0.1               # weight [out, in]; forward operand B = W^T
ref = A.double() @ W.double().t()
Bt = W.t().contiguous()
img = ops.split_b([W.t()])[0]
pimg = ops.pack_b([W.t()])[0]
for name, kw in (("fp32", dict(precision=0)), ("fp32v2", dict(precision=0, b_split=pimg)), ("x3 v1", dict(precision=1)),
                 ("x3 v2", dict(precision=1, b_split=img))):
    C = torch.zeros(M, N, device=dev)
    ops.gemm(A, Bt, C, b_kstrided=True, **kw)
    print(f"NN {name:6s} rel err vs fp64: {err(C, ref):.3e}")
ref2 = torch.nn.functional.silu(A.double()) @ W.double().t()
C = torch.zeros(M, N, device=dev)
ops.gemm(A, Bt, C, b_kstrided=True, a_act=True, precision=1, b_split=img)
print(f"NN silu(A) x3 v2 rel err: {err(C, ref2):.3e}")
C = torch.zeros(M, N, device=dev)
ops.gemm(A, Bt, C, b_kstrided=True, a_act=True, precision=0, b_split=pimg)
print(f"NN silu(A) fp32v2 rel err: {err(C, ref2):.3e}")
# two K-segments, N = 512 (two column tiles), backward-style operand (B = W as [K=out, N=in])
W2 = [rnd(256, 512) * 0.1 for _ in range(2)]
A2 = [rnd(M, 256) for _ in range(2)]
ref3 = sum(a.double() @ w.double() for a, w in zip(A2, W2))
imgs = ops.split_b(W2)
C = torch.zeros(M, 512, device=dev)
ops.gemm(A2, W2, C, b_kstrided=True, segments=True, precision=1, b_split=imgs)
print(f"NN 2 segments N=512 x3 v2 rel err: {err(C, ref3):.3e}")
bias = rnd(512)
resid = rnd(M, 512)
C = torch.zeros(M, 512, device=dev)
ops.gemm(A2, W2, C, b_kstrided=True, segments=True, precision=1, b_split=imgs, bias=bias, resid=resid)
print(f"   + bias + resid rel err: {err(C, ref3 + bias.double() + resid.double()):.3e}")

E2 = 20000 + 7
dY, X = rnd(E2, 256), rnd(E2, 256)
ref4 = dY.double().t() @ X.double()
ref5 = dY.double().t() @ torch.nn.functional.silu(X.double())
for prec in (0, 1):
    S = 16
    slabs = torch.empty(S * 256, 256, device=dev)
    out = torch.empty(256, 256, device=dev)
    ops.gemm(dY, X, slabs, a_kstrided=True, b_kstrided=True, splitk=S, precision=prec)
    ops.splitk_reduce(slabs, S, out)
    print(f"TN precision {prec} rel err vs fp64: {err(out, ref4):.3e}")
    ops.gemm(dY, X, slabs, a_kstrided=True, b_kstrided=True, b_act=True, splitk=S, precision=prec)
    ops.splitk_reduce(slabs, S, out)
    print(f"TN silu(B) precision {prec} rel err vs fp64: {err(out, ref5):.3e}")

# ---------------- speed at the layer shapes
E = 177140
D = 256
e = rnd(E, D)
pre = rnd(E, 2 * D)
gs = rnd(E, 2 * D)
Wl = [rnd(D, D) * 0.05 for _ in range(4)]
Wt = [w.t().contiguous() for w in Wl]
im = ops.split_b([w.t() for w in Wl])
pim = ops.pack_b([w.t() for w in Wl])
out2 = torch.empty(E, 2 * D, device=dev)
Nn = 12416
Pn = rnd(Nn, 4 * D)
tgt = torch.sort(torch.randint(0, Nn, (E,), generator=g)).values.to(torch.int32).to(dev)
src = torch.randint(0, Nn, (E,), generator=g).to(torch.int32).to(dev)
F2 = 2.0 * E * D * D * 2
for name, kw0, kw1 in (("fp32", dict(precision=0), dict(precision=0)),
                       ("fp32v2", dict(precision=0, b_split=pim[:2]), dict(precision=0, b_split=pim[2:])),
                       ("x3v1", dict(precision=1), dict(precision=1)),
                       ("x3v2", dict(precision=1, b_split=im[:2]), dict(precision=1, b_split=im[2:]))):
    timeit(lambda: ops.gemm([e, e], Wt[:2], [out2[:, :D], out2[:, D:]], b_kstrided=True, **kw0), F2, f"NN x2 plain {name}")
    timeit(lambda: ops.gemm([e, e], Wt[:2], [out2[:, :D], out2[:, D:]], b_kstrided=True,
                            gather_i=[Pn[:, :D], Pn[:, D:2 * D]], gather_j=[Pn[:, 2 * D:3 * D], Pn[:, 3 * D:]],
                            tgt=tgt, src=src, **kw0), F2, f"NN x2 gather {name}")
    timeit(lambda: ops.gemm([pre[:, :D], pre[:, D:]], Wt[2:], [out2[:, :D], out2[:, D:]], b_kstrided=True, a_act=True,
                            **kw1), F2, f"NN x2 silu(A) {name}")
for prec in (0, 1):
    S = 128
    slabs = [torch.empty(S * D, D, device=dev) for _ in range(2)]
    timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [e, e], slabs, a_kstrided=True, b_kstrided=True, splitk=S,
                            precision=prec), F2, f"TN x2 splitk=128 prec={prec}")
    timeit(lambda: ops.gemm([gs[:, :D], gs[:, D:]], [pre[:, :D], pre[:, D:]], slabs, a_kstrided=True, b_kstrided=True,
                            b_act=True, splitk=S, precision=prec), F2, f"TN x2 silu(B) splitk=128 prec={prec}")
