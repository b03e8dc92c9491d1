"""Experiment (round 3): main-loop rate of ONE workgroup per CU (256 tiles = one per CU) against TWO (512 tiles), fp32
DMA-fed kernel, from the slope of launch time over K -- is a lone workgroup (2 waves per SIMD) able to keep the matrix
pipe busy?  Ideal: 32 MFMAs x 64 cycles per wave per K-step; 2 (4) waves per SIMD -> 4096 (8192) cycles per K-step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from cartnet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for tiles in (256, 512, 2048):
    M = tiles * 128
    res = []
    for K in (256, 1024, 4096):
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(K, 256, generator=g) * 0.02).to(dev)
        img = ops.pack_b([W])
        out = torch.empty(M, 256, device=dev)
        fn = lambda: ops.gemm(A, W, out, b_kstrided=True, b_split=img, precision=0)
        for _ in range(30): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        res.append((K, 1e3 * e0.elapsed_time(e1) / 50))
    (k0, t0), (k1, t1), (k2, t2) = res
    slope = (t2 - t1) / ((k2 - k1) / 16)                     # us per K-step
    per_cu = tiles / 256.0
    ideal = per_cu * 32 * 64 * 4 / 2 / 2.4e3                  # tiles per CU x 2 waves per SIMD... = per_cu * 1.707 us
    print(f"tiles {tiles:5d}: " + "  ".join(f"K={k}: {t:8.1f} us" for k, t in res) +
          f"   slope {slope:6.3f} us per K-step (ideal at 2.4 GHz: {per_cu * 1.7067:6.3f}) -> main-loop rate {per_cu * 1.7067 / slope:5.3f} of peak", flush=True)
