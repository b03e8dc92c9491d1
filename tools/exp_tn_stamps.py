"""In-kernel stamps of the fp32 weight-gradient kernel (csrc/gemm_f32.h built with -DCN_TN_STAMP): per workgroup the cycles of
its K-loop, per wave the cycles spent in the counted s_waitcnt and in the barrier.
Usage: CARTNET_LIB=cartnet_amd/libcartnet_hip_tnstamp.so python tools/exp_tn_stamps.py   (GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from cartnet_amd import ops, lib as _l

dev = torch.device("cuda:0")
E, D, S = 177140, 256, 64
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dev)
gs = rnd(E, 2 * D); act = rnd(E, 2 * D)
slabs = [torch.empty(S * D, D, device=dev) for _ in range(2)]
def run():
    ops.gemm([gs[:, :D], gs[:, D:]], [act[:, :D], act[:, D:]], slabs, a_kstrided=True, b_kstrided=True, splitk=S)
for _ in range(300): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"dW x2 (dY^T X, split-K {S}): {1e3 * e0.elapsed_time(e1) / 50:.1f} us per launch (with the stamps in)")
L = _l.load()
wg = np.zeros(1024 * 4, dtype=np.uint64); wv = np.zeros(1024 * 8 * 2, dtype=np.uint64)
assert L.cartnet_debug_tn_stamps(wg.ctypes.data_as(C.c_void_p), wv.ctypes.data_as(C.c_void_p)) == 0
wg = wg.reshape(1024, 4).astype(np.float64); wv = wv.reshape(1024, 8, 2).astype(np.float64)
m = wg[:, 2] > 100
steps = wg[m, 2]
print(f"workgroups {int(m.sum())}, K-steps each {np.median(steps):.0f}; loop {np.median(wg[m, 0] / steps):.0f} cycles per K-step "
      f"(matrix pipe: 4096), clock {np.median(wg[m, 0] / (wg[m, 1] * 10.0)):.3f} GHz, loop {np.median(wg[m, 1]) * 0.01:.1f} us")
for w in range(8):
    print(f"  wave {w}: wait {np.median(wv[m, w, 0] / steps):7.1f}   barrier {np.median(wv[m, w, 1] / steps):7.1f}  cycles per K-step")
