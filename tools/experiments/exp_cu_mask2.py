"""Experiment 2 (GPU box): where does the full-chip fp32 GEMM lose efficiency?  Same launch on CU-masked streams
(hipExtStreamCreateWithCUMask; mask bit i -> XCC i % 8, then SE (i/8) % 4, CU (i/8) / 4, so bits [0, 32 m) = the first m
CUs of every shader engine) alone, and two launches on complementary halves at once.  DMA-fed fp32 kernel (packed
weights), bf16x3 kernel, weight-gradient kernel."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch

from cartnet_amd import ops

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


D = 256
E = 177140
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
gs, pre, e = rnd(E, 2 * D), rnd(E, 2 * D), rnd(E, D)
W2g, W2a = rnd(D, D) * 0.05, rnd(D, D) * 0.05
outA, outB = torch.empty(E, 2 * D, device=dev), torch.empty(E, 2 * D, device=dev)
S = 128
slabsA = [torch.empty(S * D, D, device=dev) for _ in range(2)]
slabsB = [torch.empty(S * D, D, device=dev) for _ in range(2)]


def variants(prec):
    img = (ops.pack_b if prec == 0 else ops.split_b)([W2g, W2a])

    def nn(out):
        return lambda: ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [out[:, :D], out[:, D:]], b_kstrided=True,
                                b_split=img, precision=prec)

    def tn(slabs):
        return lambda: ops.gemm([gs[:, :D], gs[:, D:]], [e, e], slabs, a_kstrided=True, b_kstrided=True, splitk=S,
                                precision=prec)
    return {"nn": (nn(outA), nn(outB)), "tn": (tn(slabsA), tn(slabsB))}


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


def together(sa, fa, sb, fb, iters=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        for _ in range(iters):
            fa()
    with torch.cuda.stream(sb):
        for _ in range(iters):
            fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / iters


full, full2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
streams = {}
for m in (7, 6, 5, 4, 3, 2):
    streams[m] = (masked_stream(range(32 * m)), masked_stream(range(32 * m, 256)))
F = 2.0 * E * D * D * 2
for prec in (0, 1):
    V = variants(prec)
    for name, (fa, fb) in V.items():
        t_full = time_on(full, fa)
        tt = together(full, fa, full2, fb)
        print(f"prec {prec} {name}: full chip {t_full:7.1f} us ({F / t_full / 1e6:6.1f} TF/s); two launches on two unmasked "
              f"streams: {tt:7.1f} us per pair ({2 * F / tt / 1e6:6.1f} TF/s)")
        for m, (sa, sb) in streams.items():
            ta, tb = time_on(sa, fa), time_on(sb, fb)
            tt = together(sa, fa, sb, fb)
            print(f"    m={m}: A on {32 * m:3d} CUs {ta:7.1f} us (x{ta / t_full:.2f}, ideal x{8 / m:.2f}; "
                  f"{F / ta / 1e6 / (m / 8):6.1f} TF/s chip-equivalent)   B on {256 - 32 * m:3d} CUs {tb:7.1f} us   "
                  f"A||B one each: {tt:7.1f} us (max {max(ta, tb):.1f}, sum {ta + tb:.1f})")
