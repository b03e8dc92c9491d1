"""Experiment (round 3): cycles per K-step of ONE workgroup per CU (256 tiles, nothing else resident) in the fp32
activation x weight kernel, from the clock stamps (-DCN_CLOCK_STAMP), for diagnostic builds with parts of the K-step
compiled out (-DCN_EXP_NO_DMA / NO_ASTORE / NO_ALOAD / NO_BARRIER / NO_FRAGS: wrong results, timing only).  The matrix
pipe needs 4,096 cycles per K-step for the workgroup's two waves per SIMD."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
from cartnet_amd import ops, lib as _lib
dev = torch.device("cuda:0")
L = _lib.load()
fn = L.cartnet_debug_clock_f32
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
g = torch.Generator().manual_seed(0)
for tiles in (256, 512):
    M, K, N = tiles * 128, 1024, 256
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, N, generator=g) * 0.05).to(dev)
    img = ops.pack_b([W])
    out = torch.empty(M, N, device=dev)
    run = lambda: ops.gemm(A, W, out, b_kstrided=True, b_split=img, precision=0)
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(20): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    buf = np.zeros(2 * 4096, dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t, w = buf[0::2][:tiles].astype(np.float64), buf[1::2][:tiles].astype(np.float64)
    ok = w > 0
    print(f"tiles {tiles}: {1e3 * e0.elapsed_time(e1) / 20:7.1f} us per launch; main loop {np.median(t[ok]) / (K // 16):7.0f} cycles per K-step "
          f"at {np.median(t[ok] / w[ok]) * 0.1:.3f} GHz", flush=True)
