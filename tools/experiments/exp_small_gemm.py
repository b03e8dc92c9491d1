"""Round 4: where the time of a configs[2]-sized precision-2 product goes.  Launch durations are read from a rocprofv3
kernel trace of THIS script (python calls cost more than these kernels run): tools/experiments/exp_small_gemm.sh.  Every case is
RUNS launches of one kernel; the post-processing (this file with the CSV as argument) takes the median of each run.
Cases: duration against K (slope = one K-step, intercept = launch + pipeline head + epilogue) for atom-sized (M = 736)
and edge-sized (M = 9,970) activation x weight products, bare, with the gather epilogue, with gather + column sums."""
import sys
RUNS = 60
CASES = [("bare", M, N, K) for M, N in ((736, 256), (736, 768), (9970, 256), (9970, 512)) for K in (16, 64, 128, 256, 512, 1024)]
CASES += [(kind, 9970, 256, K) for K in (16, 256, 1024) for kind in ("gather", "gather+sums")]
if len(sys.argv) > 1:
    import csv, statistics
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
    d = [(e - s) / 1e3 for s, e, n in rows if "cn_gemm_x3nn_kernel" in n]
    assert len(d) == RUNS * len(CASES), (len(d), RUNS * len(CASES))
    for i, c in enumerate(CASES):
        print(f"{c[0]:12s} M {c[1]:5d} N {c[2]:4d} K {c[3]:5d}: {statistics.median(d[i * RUNS + 10:(i + 1) * RUNS]):6.2f} us")
    raise SystemExit
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cartnet_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
nn = 736
for kind, M, N, K in CASES:
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, N, generator=g) * 0.05).to(dev)
    img = ops.split_b([W])
    C = torch.empty(M, N, device=dev)
    kw = {}
    if kind != "bare":
        tgt = torch.randint(0, nn, (M,), generator=g).sort().values.int().to(dev)
        src = torch.randint(0, nn, (M,), generator=g).int().to(dev)
        kw = dict(gather_i=[torch.randn(nn, N, generator=g).to(dev)], gather_j=[torch.randn(nn, N, generator=g).to(dev)], tgt=tgt, src=src)
        if kind == "gather+sums":
            tiles = ops.gemm_tiles_m(M)
            kw.update(colsum=[torch.zeros(tiles * N, dtype=torch.float64, device=dev)], colsq=[torch.zeros(tiles * N, dtype=torch.float64, device=dev)])
    torch.cuda.synchronize()
    for _ in range(RUNS):
        ops.gemm([A], [W], [C], b_kstrided=True, b_split=img, precision=2, **kw)
    torch.cuda.synchronize()
