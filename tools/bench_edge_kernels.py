"""Times the HBM-bound per-edge kernels of one CartNet layer at the benchmark shape (64 crystals x 194 atoms) in
isolation and prints achieved GB/s against their algorithmic bytes.  Run on the GPU box: python tools/bench_edge_kernels.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops
from cartnet_amd.synthetic import make_batch

dev = torch.device("cuda:0")
D = 256
b = make_batch(64, 194, first=100_000).to(dev)
N, E = int(b.x.shape[0]), int(b.edge_index.shape[1])
lay = ops.GraphLayout(b.edge_index, N, b.ptr.to(dev) if hasattr(b, "ptr") else None)
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
# a second big tensor touched between timed launches would defeat MALL reuse; here every iteration re-reads the same
# 362 MB which exceeds the 256 MB MALL, so the figures are HBM figures
dpre = rnd(E, 2 * D)
dPn = torch.empty(N, 4 * D, device=dev)
t = timeit(lambda: ops.segment_sum(dpre, lay.rowptr, None, dPn[:, :2 * D]))
print(f"segment_sum tgt  W=512: {t:7.1f} us  {E * 2 * D * 4 / t / 1e3:7.1f} GB/s")
t = timeit(lambda: ops.segment_sum(dpre, lay.colptr, lay.perm, dPn[:, 2 * D:]))
print(f"segment_sum src  W=512: {t:7.1f} us  {E * 2 * D * 4 / t / 1e3:7.1f} GB/s")
def pair():
    ops.segment_sum(dpre, lay.rowptr, None, dPn[:, :2 * D])
    ops.segment_sum(dpre, lay.colptr, lay.perm, dPn[:, 2 * D:])
t = timeit(pair)
print(f"segment_sum tgt+src pair: {t:7.1f} us  {2 * E * 2 * D * 4 / t / 1e3:7.1f} GB/s")
# gate kernels (forward after a GEMM-like ascending write of gs; backward statistics then apply)
gs, e_in, env = rnd(E, 2 * D), rnd(E, D), torch.rand(E, generator=g).to(dev)
e_out, aggr = torch.empty(E, D, device=dev), torch.empty(N, D, device=dev)
npart = ops.gate_nparts(N)
mr = torch.cat([torch.zeros(D), torch.ones(D)]).to(dev)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
p1, p2, p3, p4 = (torch.zeros(npart * D, dtype=torch.float64, device=dev) for _ in range(4))
src_gs = rnd(E, 2 * D)
def fwd_after_write():
    gs.copy_(src_gs)          # stands in for the GEMM that has just written gs in ascending row order
    ops.gate_scatter_fwd(gs, e_in, env, lay, mr, gamma, beta, e_out, aggr, p1, p2)
def write_only():
    gs.copy_(src_gs)
tw = timeit(write_only)
t = timeit(fwd_after_write) - tw
print(f"gate_scatter_fwd after ascending write of gs: {t:7.1f} us  {E * D * 16 / t / 1e3:7.1f} GB/s")
t = timeit(lambda: ops.gate_scatter_fwd(gs, e_in, env, lay, mr, gamma, beta, e_out, aggr, p1, p2))
print(f"gate_scatter_fwd alone:                       {t:7.1f} us  {E * D * 16 / t / 1e3:7.1f} GB/s")
de_out, daggr, sums = rnd(E, D), rnd(N, D), torch.zeros(2 * D, device=dev)
def bwd_pair():
    ops.gate_scatter_bwd_stats(gs, de_out, daggr, env, lay, mr, gamma, beta, p1, p2)
    ops.gate_scatter_bwd_apply(gs, de_out, daggr, env, lay, mr, gamma, beta, sums, True, p3, p4)
t = timeit(bwd_pair)
print(f"gate_scatter_bwd stats+apply:                 {t:7.1f} us  {E * D * 32 / t / 1e3:7.1f} GB/s")
print(f"N={N} E={E}")
