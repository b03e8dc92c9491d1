"""Experiment (round 3): the shader clock INSIDE the main loop of the fp32 and the bf16x3 activation x weight kernels.

Needs a diagnostic library (CARTNET_HIPCC_EXTRA=-DCN_CLOCK_STAMP python -m cartnet_amd.build --force): every workgroup
stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its main loop into a buffer of its own; the in-kernel
clock is their ratio x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6).  Each case runs back to back for
~2.5 s first so that the chip is at the clock it HOLDS under that load, then the median over workgroups is read.
Also prints the launch time and, from K = 1024 / 4096, the main loop's rate in MFMA cycles per K-step.
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
from cartnet_amd import ops, lib as _lib

dev = torch.device("cuda:0")
L = _lib.load()
g = torch.Generator().manual_seed(0)
M = 177140
N = 512


def read(name):
    fn = getattr(L, "cartnet_debug_clock_" + name)
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    buf = np.zeros(2 * 4096, dtype=np.uint64)
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    t, w = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
    ok = w > 0
    return np.median(t[ok] / w[ok]) * 0.1, np.median(t[ok]), np.median(w[ok]) * 10.0     # GHz, cycles, ns


import os as _os
_shape16 = True      # precision 1 runs on the 16x16x32 shape (gemm_x3s.h); the 32x32x16 kernel serves precision 2 only
_only_x3 = bool(_os.environ.get("EXP_ONLY_X3"))
for prec, name in ((0, "f32"), (1, "x3s" if _shape16 else "x3")):
    if _only_x3 and prec == 0:
        continue
    for K in ((1024,) if _only_x3 else (256, 1024, 4096)):
        for zero in ((False,) if _only_x3 else (False, True)):
            A = (torch.zeros(M, K) if zero else torch.randn(M, K, generator=g)).to(dev)
            W = (torch.zeros(K, N) if zero else torch.randn(K, N, generator=g) * 0.05).to(dev)
            img = ops.split_b([W]) if prec else ops.pack_b([W])
            out = torch.empty(M, N, device=dev)
            fn = lambda: ops.gemm(A, W, out, b_kstrided=True, b_split=img, precision=prec)
            fn(); torch.cuda.synchronize()
            t0 = time.time()
            while time.time() - t0 < 2.5:
                for _ in range(20): fn()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20 if K < 4096 else 5
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            ghz, cyc, ns = read(name)
            steps = K // 16
            per_step = cyc / max(steps - 1, 1)
            # two workgroups share a CU: per K-step a SIMD owes 4 waves x (32 MFMAs x 64 cycles / 2 waves... ) see below
            ideal = 4096.0 if prec == 0 else 3072.0          # matrix-pipe cycles per K-step of BOTH workgroups of a CU
            print(f"{name} K={K:5d} {'zeros ' if zero else 'random'}: {us:8.1f} us per launch, main loop {ns/1e3:8.1f} us "
                  f"= {cyc:10.0f} cycles at {ghz:5.3f} GHz; {per_step:7.0f} cycles per K-step "
                  f"(matrix pipe needs {ideal:.0f} for the CU's two workgroups -> {ideal / per_step:5.3f})", flush=True)
            del A, W, out
