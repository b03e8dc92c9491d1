"""Experiment (GPU box): do CU-masked HIP streams (hipExtStreamCreateWithCUMask) partition the chip, so that an
MFMA-bound GEMM on one CU subset and an HBM-bound edge kernel on the other really overlap?

Prints, for several masks: the layer GEMM's time on the masked stream alone, an HBM-bound kernel's time alone, and the
wall time of both issued together (GEMM on mask A, edge kernels on mask B) against their sum.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch

from cartnet_amd import ops
from cartnet_amd.synthetic import make_batch

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) enabled."""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


D = 256
b = make_batch(64, 194, first=100_000).to(dev)
N, E = int(b.x.shape[0]), int(b.edge_index.shape[1])
lay = ops.GraphLayout(b.edge_index, N, b.ptr.to(dev))
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
gs, pre = rnd(E, 2 * D), rnd(E, 2 * D)
W2g, W2a = rnd(D, D) * 0.05, rnd(D, D) * 0.05
out2 = torch.empty(E, 2 * D, device=dev)
dpre = rnd(E, 2 * D)
dPn = torch.empty(N, 4 * D, device=dev)


def gemm():
    ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [out2[:, :D], out2[:, D:]], b_kstrided=True)


def hbm():
    ops.segment_sum(dpre, lay.rowptr, None, dPn[:, :2 * D])
    ops.segment_sum(dpre, lay.colptr, lay.perm, dPn[:, 2 * D:])


def time_on(stream, fn, iters=10):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def together(sa, fa, ia, sb, fb, ib):
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        for _ in range(ia):
            fa()
    with torch.cuda.stream(sb):
        for _ in range(ib):
            fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


full = torch.cuda.Stream(device=dev)
full2 = torch.cuda.Stream(device=dev)
print(f"E={E} N={N}")
tg, th = time_on(full, gemm), time_on(full, hbm)
print(f"full chip: gemm {tg:.1f} us   hbm pair {th:.1f} us")
w = together(full, gemm, 10, full2, hbm, 10)
print(f"full+full together 10+10: wall {w:.0f} us  vs sum {10 * (tg + th):.0f}  vs max {10 * max(tg, th):.0f}")

layouts = {
    "first-k": lambda k: list(range(k)),
    "interleave(i*256/k)": lambda k: [int(i * 256 / k) for i in range(k)],
    "per-32-word low bits": lambda k: [w * 32 + j for w in range(8) for j in range(k // 8)],
}
for name, fn in layouts.items():
    for k in (192, 160, 128, 64):
        try:
            bits = fn(k)
            sa = masked_stream(bits)
            rest = sorted(set(range(256)) - set(bits))
            sb = masked_stream(rest)
        except Exception as exc:
            print(name, k, "failed:", exc)
            continue
        ta, tb = time_on(sa, gemm), time_on(sb, hbm)
        tb_full = time_on(sa, hbm)
        # balanced iteration counts so that both sides take about equally long
        ia = 10
        ib = max(1, int(round(ia * ta / tb)))
        w = together(sa, gemm, ia, sb, hbm, ib)
        print(f"{name:24s} k={k:3d}: gemm on k CUs {ta:7.1f} us (x{ta / tg:.2f}, ideal x{256 / k:.2f})   hbm on 256-k CUs "
              f"{tb:7.1f} us (x{tb / th:.2f})  hbm on k CUs {tb_full:7.1f}   together {ia}+{ib}: wall {w:.0f} us vs sum "
              f"{ia * ta + ib * tb:.0f} vs max {max(ia * ta, ib * tb):.0f}")
