"""Experiment (GPU box): atom-sized activation x weight products (M = 12,416) with and without their weight image --
with the image the launch takes the 128-wide DMA-fed kernel, without it the register-staged narrow-tile kernels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
M, D = 12416, 256
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
X4, resid, x = rnd(M, 4 * D), rnd(M, D), rnd(M, 2 * D)
W4 = [rnd(D, D) * 0.05 for _ in range(4)]
fold = torch.cat(ops.pack_b(W4))
Wa = rnd(2 * D, D) * 0.05
img_a = ops.pack_b([Wa])
Wb = rnd(D, 2 * D) * 0.05
img_b = ops.pack_b([Wb])
out, out2 = torch.empty(M, D, device=dev), torch.empty(M, 2 * D, device=dev)


def sustained(fn, seconds=1.0):
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


segs = [X4[:, i * D:(i + 1) * D] for i in range(4)]
xin = rnd(M, D)
Wn = [rnd(D, D) * 0.05 for _ in range(4)]
img_n = ops.pack_b(Wn)
Pn = torch.empty(M, 4 * D, device=dev)
bn2 = [rnd(D), rnd(D), None, None]
Wh = rnd(D, D // 2) * 0.05
cases = {
    "node terms forward (4 groups, K=N=256), image": lambda: ops.gemm([xin] * 4, Wn, [Pn[:, i * D:(i + 1) * D] for i in range(4)], b_kstrided=True, b_split=img_n, bias=bn2),
    "dX of node terms (4 segments, K=1024), image": lambda: ops.gemm(segs, W4, out, b_kstrided=True, segments=True, resid=resid, b_split_folded=fold),
    "dX of node terms (4 segments, K=1024), no image": lambda: ops.gemm(segs, W4, out, b_kstrided=True, segments=True, resid=resid),
    "atom encoder (K=512, silu in/out), image": lambda: ops.gemm(x, Wa, out, b_kstrided=True, a_act=True, out_act=True, b_split=img_a),
    "atom encoder (K=512, silu in/out), no image": lambda: ops.gemm(x, Wa, out, b_kstrided=True, a_act=True, out_act=True),
    "atom encoder backward (N=512), image": lambda: ops.gemm(out, Wb, out2, b_kstrided=True, dact=x, b_split=img_b),
    "atom encoder backward (N=512), no image": lambda: ops.gemm(out, Wb, out2, b_kstrided=True, dact=x),
}
print("CARTNET_Q =", os.environ.get("CARTNET_Q"), "(experimental builds only)")
for name, fn in cases.items():
    print(f"  {name:52s} {sustained(fn):7.1f} us")
