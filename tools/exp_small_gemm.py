"""Round 4: where the time of a configs[2]-sized precision-2 product goes -- launch duration against K (slope = one K-step,
intercept = everything else: launch, head of the pipeline, epilogue) for the atom-sized (M = 736) and edge-sized
(M = 9,970) activation x weight products, bare and with the edge epilogue (gather + BatchNorm column sums)."""
import sys, torch
sys.path.insert(0, ".")
from cartnet_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
def timed(f, n=300):
    for _ in range(50): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for M, N in ((736, 256), (736, 768), (9970, 256), (9970, 512)):
    row = []
    for K in (16, 64, 128, 256, 512, 1024):
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(K, N, generator=g) * 0.05).to(dev)
        img = ops.split_b([W])
        C = torch.empty(M, N, device=dev)
        row.append(timed(lambda: ops.gemm([A], [W], [C], b_kstrided=True, b_split=img, precision=2)))
    print(f"bare   M {M:5d} N {N}: " + "  ".join(f"{t:6.2f}" for t in row) + "  us at K = 16, 64, 128, 256, 512, 1024", flush=True)
M, N = 9970, 256
tiles = ops.gemm_tiles_m(M)
nn = 736
tgt = torch.randint(0, nn, (M,), generator=g).sort().values.int().to(dev)
src = torch.randint(0, nn, (M,), generator=g).int().to(dev)
gi = torch.randn(nn, N, generator=g).to(dev); gj = torch.randn(nn, N, generator=g).to(dev)
for K in (16, 256, 1024):
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, N, generator=g) * 0.05).to(dev)
    img = ops.split_b([W])
    C = torch.empty(M, N, device=dev)
    cs = torch.zeros(tiles * N, dtype=torch.float64, device=dev); cq = torch.zeros_like(cs)
    t0 = timed(lambda: ops.gemm([A], [W], [C], b_kstrided=True, b_split=img, precision=2))
    t1 = timed(lambda: ops.gemm([A], [W], [C], b_kstrided=True, b_split=img, precision=2, gather_i=[gi], gather_j=[gj], tgt=tgt, src=src))
    t2 = timed(lambda: ops.gemm([A], [W], [C], b_kstrided=True, b_split=img, precision=2, gather_i=[gi], gather_j=[gj], tgt=tgt, src=src,
                                colsum=[cs], colsq=[cq]))
    print(f"K {K}: bare {t0:.2f}  + gather {t1:.2f}  + gather + sums {t2:.2f} us", flush=True)
