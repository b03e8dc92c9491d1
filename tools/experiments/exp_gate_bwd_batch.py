"""gate_scatter_bwd apply pass alone on the chip: edges per trip (variants built with -DCN_GATE_BWD_BATCH=2 / 8 against the
product's 4; CARTNET_LIB selects).  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cartnet_amd import ops
from cartnet_amd import lib
dev = torch.device("cuda:0")
E, N, D = 177140, 12416, 256
g = torch.Generator().manual_seed(0)
gs0 = torch.randn(E, 2 * D, generator=g).to(dev); de = torch.randn(E, D, generator=g).to(dev); daggr = torch.randn(N, D, generator=g).to(dev)
env = torch.rand(E, generator=g).to(dev)
deg = torch.full((N,), E // N, dtype=torch.int64); deg[: E - int(deg.sum())] += 1
rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).to(torch.int32).to(dev)
mr = torch.cat([gs0[:, :D].mean(0), torch.rsqrt(gs0[:, :D].var(0, unbiased=False) + 1e-5)]).contiguous()
gamma, beta = torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
sums = torch.randn(2 * D, generator=g).to(dev)
npart = ops.gate_nparts(N)
p3, p4 = (torch.zeros(npart * D, dtype=torch.float64, device=dev) for _ in range(2))
gs = gs0.clone()
tgt = torch.repeat_interleave(torch.arange(N), deg)
src = torch.randint(0, N, (E,), generator=g)
lay = ops.GraphLayout(torch.stack([src, tgt]).to(dev), N, None)
def run():
    ops.gate_scatter_bwd_apply(gs, de, daggr, env, lay, mr, gamma, beta, sums, True, p3, p4)
try:
    run()
except Exception as e:
    print("call failed:", e); raise
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): run()
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    t = 1e3 * e0.elapsed_time(e1) / 100
    print(f"{os.environ.get('CARTNET_LIB','product')}: gate apply {t:.1f} us  {(E*D*4*(2+1+2))/t/1e6:.2f} TB/s", flush=True)
print("checksum", float(gs.double().abs().sum()), float(p3.sum()), float(p4.sum()))
