"""Experiment (GPU box): the 128-wide DMA-fed fp32 kernel (three workgroups per CU) against the 256-wide one on the
layer GEMM variants, sustained.  Run with TILE_POLICY=256 and =128 (unset / 0: the library's own choice per variant; CartnetGemmArgs.tile_policy)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
E, D = 177140, 256
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
e, pre, gs = rnd(E, D), rnd(E, 2 * D), rnd(E, 2 * D)
W = [rnd(D, D) * 0.05 for _ in range(2)]
img = ops.pack_b(W)
N = E // 14
Pn = rnd(N, 4 * D)
tgt = torch.sort(torch.randint(0, N, (E,), generator=g)).values.to(torch.int32).to(dev)
src = torch.randint(0, N, (E,), generator=g).to(torch.int32).to(dev)
out2, out1 = torch.empty(E, 2 * D, device=dev), torch.empty(E, D, device=dev)
tiles = ops.gemm_tiles_m(E)
TP = int(os.environ.get('TILE_POLICY', '0'))
cs = torch.empty(tiles * D, dtype=torch.float64, device=dev); cq = torch.empty_like(cs)
F2 = 2.0 * E * D * D * 2


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


cases = {
    "plain": lambda: ops.gemm([gs[:, :D], gs[:, D:]], W, [out2[:, :D], out2[:, D:]], b_kstrided=True, tile_policy=TP, b_split=img),
    "gather (GEMM1)": lambda: ops.gemm([e, e], W, [out2[:, :D], out2[:, D:]], b_kstrided=True, tile_policy=TP, b_split=img,
                                       gather_i=[Pn[:, :D], Pn[:, D:2 * D]], gather_j=[Pn[:, 2 * D:3 * D], Pn[:, 3 * D:]],
                                       tgt=tgt, src=src),
    "silu(A)+stats (GEMM2)": lambda: ops.gemm([pre[:, :D], pre[:, D:]], W, [out2[:, :D], out2[:, D:]], b_kstrided=True, tile_policy=TP,
                                              b_split=img, a_act=True, colsum=[cs, None], colsq=[cq, None]),
    "dact (dpre)": lambda: ops.gemm([gs[:, :D], gs[:, D:]], W, [out2[:, :D], out2[:, D:]], b_kstrided=True, tile_policy=TP, b_split=img,
                                    dact=[pre[:, :D], pre[:, D:]]),
}
Wk = [rnd(2 * D, D) * 0.05 for _ in range(2)]          # dE: two K-segments (adjacent column blocks of dpre) folded into K = 512
folded = torch.cat(ops.pack_b([Wk[0][:D].contiguous(), Wk[0][D:].contiguous()]))
cases["2 K-segments + resid (dE)"] = lambda: ops.gemm([pre[:, :D], pre[:, D:]], [Wk[0][:D], Wk[0][D:]], out1, b_kstrided=True, tile_policy=TP,
                                                     segments=True, resid=e, b_split_folded=folded)
print("TILE_POLICY =", TP)
for name, fn in cases.items():
    t = sustained(fn)
    print(f"  {name:24s} {t:7.1f} us  {F2 / t / 1e6:6.1f} TF/s ({F2 / t / 1e6 / 157.3:.3f})")
