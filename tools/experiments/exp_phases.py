"""Experiment (round 3): the life of every workgroup of one fp32 activation x weight launch, from stamps inside the kernel.

Needs a diagnostic library: gemm_f32.o compiled with -DCN_PHASE_STAMP.  Every workgroup records the 100 MHz time at
entry, main-loop start, main-loop end, last store issued, all stores acknowledged, and the hardware CU it ran on
(HW_REG_HW_ID / HW_REG_XCC_ID).  One warm launch of the plain two-group layer product is read back: phase lengths
(median over workgroups), and per CU the timeline of the workgroups it ran -- how much of the launch each CU had 0, 1
or 2 workgroups inside their main loops, and the gap between a workgroup leaving and its successor's first stamp."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
from cartnet_amd import ops, lib as _lib

dev = torch.device("cuda:0")
L = _lib.load()
PREC = int(os.environ.get("EXP_PRECISION", "0"))          # 1: the bf16x3 kernel (gemm_x3s.o built with -DCN_PHASE_STAMP)
QUAD = os.environ.get("CARTNET_Q", "0") != "0"            # round 4: the four-workgroups-per-CU kernel (gemm_f32q.o stamped)
fn = L.cartnet_debug_phase_x3s if PREC else (L.cartnet_debug_phase_f32q if QUAD else L.cartnet_debug_phase_f32)
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
g = torch.Generator().manual_seed(0)
E, D = 177140, 256
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = torch.randn(E, K, generator=g).to(dev)
Ws = [(torch.randn(K, D, generator=g) * 0.05).to(dev) for _ in range(2)]
imgs = ops.split_b(Ws) if PREC else ops.pack_b(Ws)
out = torch.empty(E, 2 * D, device=dev)
run = lambda: ops.gemm([A, A], Ws, [out[:, :D], out[:, D:]], b_kstrided=True, b_split=imgs, precision=PREC)
for _ in range(200):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"K = {K}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per launch (with the stamps in)")
buf = np.zeros(8192 * 8, dtype=np.uint64)
assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(8192, 8)
n = (4 if QUAD else 2) * ((E + 127) // 128)
s = s[:n].astype(np.int64)
t0 = s[:, 0].min()
ent, l0, l1, si, sd = [(s[:, i] - t0) * 0.01 for i in range(5)]        # us since the first workgroup's entry
hw = s[:, 5]
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 15)     # xcc, se, cu
med = lambda x: float(np.median(x))
print(f"workgroups {n}, launch span by the stamps {sd.max():.1f} us, distinct CUs {len(set(cu.tolist()))}")
print(f"median per workgroup: prologue {med(l0 - ent):.2f} us, main loop {med(l1 - l0):.2f}, epilogue to last store issued "
      f"{med(si - l1):.2f}, stores acknowledged after {med(sd - si):.2f} more; whole life {med(sd - ent):.2f}")
b0, b1 = (s[:, 6] - t0) * 0.01, (s[:, 7] - t0) * 0.01
print(f"inside the epilogue (wave 0): first 32x32 block stored after {med(b0 - l1):.2f} us, second after {med(b1 - b0):.2f} more, "
      f"third + fourth after {med(si - b1):.2f} more")
for q in (10, 50, 90):
    print(f"  p{q}: prologue {np.percentile(l0 - ent, q):.2f}  loop {np.percentile(l1 - l0, q):.2f}  epilogue "
          f"{np.percentile(si - l1, q):.2f}  ack {np.percentile(sd - si, q):.2f}")
# per CU: occupancy of main loops over time, successor gaps
span = sd.max()
res = 0.05
T = int(span / res) + 2
occ_hist = np.zeros(6)
alive_hist = np.zeros(6)
gaps = []
for c in set(cu.tolist()):
    idx = np.where(cu == c)[0]
    loops = np.zeros(T, dtype=np.int32)
    alive = np.zeros(T, dtype=np.int32)
    for i in idx:
        loops[int(l0[i] / res):int(l1[i] / res)] += 1
        alive[int(ent[i] / res):int(sd[i] / res)] += 1
    for k in range(6):
        occ_hist[k] += (loops[:int(span / res)] == k).sum()
        alive_hist[k] += (alive[:int(span / res)] == k).sum()
    # successor gap: sort by entry; a workgroup's entry minus the latest exit before it among the CU's workgroups
    order = idx[np.argsort(ent[idx])]
    exits = np.sort(sd[idx])
    for i in order[(4 if QUAD else 2):]:                       # the first two fill the empty CU
        prev = exits[exits <= ent[i] + 1e-9]
        if len(prev):
            gaps.append(ent[i] - prev.max())
occ_hist /= occ_hist.sum()
alive_hist /= alive_hist.sum()
print("share of CU-time with k workgroups INSIDE THEIR MAIN LOOP: " + "  ".join(f"k={k}: {occ_hist[k]:.3f}" for k in range(6)))
print("share of CU-time with k workgroups RESIDENT (entry .. stores acknowledged): " + "  ".join(f"k={k}: {alive_hist[k]:.3f}" for k in range(6)))
gaps = np.array(gaps)
print(f"gap between a workgroup's last acknowledged store and the next entry on the same CU: median {np.median(gaps):.2f} us, "
      f"p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f} ({len(gaps)} successions)")
# one CU's timeline as an example
c = sorted(set(cu.tolist()))[7]
idx = np.where(cu == c)[0]
print(f"CU {c}: (entry, loop start, loop end, last store, acknowledged) us")
for i in idx[np.argsort(ent[idx])]:
    print(f"   wg {i:5d}: {ent[i]:7.2f} {l0[i]:7.2f} {l1[i]:7.2f} {si[i]:7.2f} {sd[i]:7.2f}")

if QUAD and hasattr(L, "cartnet_debug_phase2_f32q"):
    f2 = L.cartnet_debug_phase2_f32q
    f2.argtypes = [ctypes.c_void_p]
    f2.restype = ctypes.c_int
    b2 = np.zeros(8192 * 4, dtype=np.uint64)
    assert f2(b2.ctypes.data_as(ctypes.c_void_p)) == 0
    q = b2.reshape(8192, 4)[:n].astype(np.float64) / (K // 16)
    print("wave 0, shader cycles per K-step (median / p10 / p90 over workgroups): "
          + "  ".join(f"{nm} {np.median(q[:, i]):.0f} / {np.percentile(q[:, i], 10):.0f} / {np.percentile(q[:, i], 90):.0f}"
                      for i, nm in enumerate(("whole loop", "wait for the DMA", "barrier"))))
