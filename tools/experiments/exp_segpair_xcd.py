"""cartnet_segment_sum_pair alone on the chip on crystal-structured graphs (64 crystals x 194 atoms, ~14.3 edges per atom, both
atoms of an edge in one crystal): time per launch.  Run under tools/experiments/pmc_one_kernel.sh for the fetched bytes; variants
built with -DCN_SEG_XCD_NODES=0 (plain dealing) / 128 / 256 / 512 (CARTNET_LIB selects).  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from cartnet_amd import ops
dev = torch.device("cuda:0")
B, A, D = 64, 194, 256
N = B * A
g = torch.Generator().manual_seed(0)
deg = torch.randint(10, 19, (N,), generator=g)
tgt = torch.repeat_interleave(torch.arange(N), deg)
E = int(tgt.numel())
src = (tgt // A) * A + torch.randint(0, A, (E,), generator=g)
gptr = (torch.arange(B + 1, dtype=torch.int64) * A).to(dev)
lay = ops.GraphLayout(torch.stack([src, tgt]).to(dev), N, gptr)
lay.validate()
rows = torch.randn(E, 2 * D, generator=g).to(dev)
out = torch.empty(N, 4 * D, device=dev)
def run(): ops.segment_sum_pair(rows, lay, out[:, :2 * D], out[:, 2 * D:])
run(); torch.cuda.synchronize()
ref_t = torch.zeros(N, 2 * D, dtype=torch.float64, device=dev).index_add_(0, tgt.to(dev), rows.double())
ref_s = torch.zeros(N, 2 * D, dtype=torch.float64, device=dev).index_add_(0, src.to(dev), rows.double())
err = max(float((out[:, :2 * D].double() - ref_t).abs().max()), float((out[:, 2 * D:].double() - ref_s).abs().max()))
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): run()
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    t = 1e3 * e0.elapsed_time(e1) / 100
    print(f"{os.path.basename(os.environ.get('CARTNET_LIB', 'product'))}: E={E} segment_sum_pair {t:.1f} us  max|err| {err:.2e}", flush=True)
