"""Experiment 3 (GPU box): is the full-chip fp32 GEMM power/clock-limited?  (a) random vs zero operands (zeros draw
less power: MI355X_MICROARCH.md 'DVFS give-back'); (b) rocm-smi sclk / power sampled during 3 s of back-to-back launches
on the full chip and on CU-masked halves."""
import ctypes as C
import os
import subprocess
import sys
import threading
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
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value, device=dev)


D, E = 256, 177140
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
W2g, W2a = rnd(D, D) * 0.05, rnd(D, D) * 0.05
out = torch.empty(E, 2 * D, device=dev)
F = 2.0 * E * D * D * 2


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        import json
        d = json.loads(o)
        c = next(iter(d.values()))
        keep = {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower()}
        return keep
    except Exception as exc:
        return {"error": str(exc)}


def sustained(stream, fn, seconds, label):
    samples = []
    stop = threading.Event()

    def poll():
        time.sleep(0.8)
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.5)
    th = threading.Thread(target=poll)
    th.start()
    n = 0
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        while time.perf_counter() - t0 < seconds:
            for _ in range(50):
                fn()
            stream.synchronize()
            n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    print(f"{label}: {dt / n * 1e6:7.1f} us per launch ({F * n / dt / 1e12:6.1f} TF/s) over {dt:.1f} s; smi: {samples[-2:] if samples else None}", flush=True)


full = torch.cuda.Stream(device=dev)
half = masked_stream(range(128))
for prec in (0, 1):
    img = (ops.pack_b if prec == 0 else ops.split_b)([W2g, W2a])
    for fill in ("random", "zeros"):
        gs = rnd(E, 2 * D) if fill == "random" else torch.zeros(E, 2 * D, device=dev)
        fn = lambda: ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [out[:, :D], out[:, D:]], b_kstrided=True,
                              b_split=img, precision=prec)
        sustained(full, fn, 3.0, f"prec {prec} {fill:6s} full chip ")
        sustained(half, fn, 3.0, f"prec {prec} {fill:6s} 128 CUs   ")
print("idle:", smi())
