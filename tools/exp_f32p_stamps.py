"""In-kernel stamps of the persistent fp32 GEMM kernel (csrc/gemm_f32p.h built with -DCN_P_STAMP): per workgroup and tile the
shader clock, the 100 MHz clock, and the cycles wave 0 spent in the counted s_waitcnt and in the barrier of its K-steps.
Usage: CARTNET_LIB=cartnet_amd/libcartnet_hip_pstamp.so python tools/exp_f32p_stamps.py [form]   (GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from cartnet_amd import ops, lib as _l

dev = torch.device("cuda:0")
E, D = 177140, 256
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dev)
gs = rnd(E, 2 * D); pre = rnd(E, 2 * D)
W2g, W2a = rnd(D, D, sc=0.05), rnd(D, D, sc=0.05)
T = lambda w: w.t().contiguous()
W2gT, W2aT = T(W2g), T(W2a)
img = ops.pack_b([W2gT, W2aT])
out = torch.empty(E, 2 * D, device=dev)
form = sys.argv[1] if len(sys.argv) > 1 else "plain"
def run():
    if form == "plain":
        ops.gemm([gs[:, :D], gs[:, D:]], [W2gT, W2aT], [out[:, :D], out[:, D:]], b_kstrided=True, b_split=img, tile_policy=3)
    else:
        ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], [out[:, :D], out[:, D:]], b_kstrided=True, b_split=img, a_act=True,
                 tile_policy=3)
for _ in range(300): run()          # clocks settled
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"{form}: {1e3 * e0.elapsed_time(e1) / 50:.1f} us per launch (with the stamps in)")
L = _l.load()
buf = np.zeros(256 * 16 * 4, dtype=np.uint64)
rc = L.cartnet_debug_p_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
s = buf.reshape(256, 16, 4).astype(np.int64)
ntiles = np.array([int((s[w, :, 0] != 0).sum()) - 1 for w in range(256)])      # last record = end of the drain
print("tiles per workgroup:", np.bincount(ntiles))
rows = []
for w in range(256):
    n = ntiles[w]
    if n < 3: continue
    t = s[w, :n + 1]
    cyc = np.diff(t[:n, 0]); wall = np.diff(t[:n, 2]) * 10.0          # ns
    sw = np.diff(t[:n, 1]); sb = np.diff(t[:n, 3])
    rows.append((cyc.mean(), (cyc / wall).mean(), sw.mean(), sb.mean(), (t[n, 0] - t[n - 1, 0]), (t[n, 2] - t[0, 2]) * 0.01,
                 t[0, 2], t[n, 2]))
r = np.array(rows, dtype=np.float64)
print(f"per tile (16 K-steps), median over workgroups: {np.median(r[:, 0]):.0f} cycles = {np.median(r[:, 0]) / 16:.0f} per K-step "
      f"(matrix pipe: 4096); clock {np.median(r[:, 1]):.3f} GHz")
print(f"  of which in s_waitcnt {np.median(r[:, 2]):.0f} cycles per tile, in the barrier {np.median(r[:, 3]):.0f} (wave 0)")
print(f"  drain after the last tile: {np.median(r[:, 4]):.0f} cycles; first tile end -> drain end {np.median(r[:, 5]):.1f} us (p10 {np.percentile(r[:, 5], 10):.1f}, p90 {np.percentile(r[:, 5], 90):.1f})")
t0 = r[:, 6].min()
print(f"  first tile ends at {np.median((r[:, 6] - t0) * 0.01):.1f} us after the earliest one (p90 {np.percentile((r[:, 6] - t0) * 0.01, 90):.1f}); "
      f"last workgroup done at {(r[:, 7].max() - t0) * 0.01:.1f} us, median {np.median((r[:, 7] - t0) * 0.01):.1f}")
wb = np.zeros(256 * 8 * 2, dtype=np.uint64)
assert L.cartnet_debug_p_waves(wb.ctypes.data_as(C.c_void_p)) == 0
wv = wb.reshape(256, 8, 2).astype(np.float64)
steps = ntiles[:, None] * 16.0
print("per wave and K-step, median over workgroups (cycles in the counted wait / in the barrier):")
for w in range(8):
    print(f"  wave {w}: wait {np.median(wv[:, w, 0] / steps[:, 0]):7.1f}   barrier {np.median(wv[:, w, 1] / steps[:, 0]):7.1f}")
