"""Round 4: the shader clock the chip HOLDS inside the main loop of the plain two-group layer product, for the shipped
256-wide kernel (CARTNET_Q=0) and the four-workgroups-per-CU kernel (CARTNET_Q=1).  Needs a library with gemm_f32.o and
gemm_f32q.o built with -DCN_CLOCK_STAMP (tools/build_variant.sh clock "-DCN_CLOCK_STAMP" gemm_f32.hip gemm_f32q.hip).
Every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop; after 2.5 s of
back-to-back launches the median ratio is the in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
from cartnet_amd import ops, lib as _lib, telemetry as tele

dev = torch.device("cuda:0")
L = _lib.load()
quad = os.environ.get("CARTNET_Q", "0") != "0"
fn = L.cartnet_debug_clock_f32q if quad else L.cartnet_debug_clock_f32
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
g = torch.Generator().manual_seed(0)
E, D = 177140, 256
for zero in (False, True):
    A = (torch.zeros(E, 2 * D) if zero else torch.randn(E, 2 * D, generator=g)).to(dev)
    Ws = [((torch.zeros(D, D) if zero else torch.randn(D, D, generator=g) * 0.05)).to(dev) for _ in range(2)]
    img = ops.pack_b(Ws)
    out = torch.empty(E, 2 * D, device=dev)
    run = lambda: ops.gemm([A[:, :D], A[:, D:]], Ws, [out[:, :D], out[:, D:]], b_kstrided=True, b_split=img, precision=0)
    run(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 2.5:
        for _ in range(20): run()
        torch.cuda.synchronize()
    s = tele.Sampler(0, period=0.05).start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    t = s.stop()
    us = 1e3 * e0.elapsed_time(e1) / 200
    buf = np.zeros(2 * 4096, dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    c, w = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
    ok = w > 0
    print(f"{'quad' if quad else 'wide'} {'zeros ' if zero else 'random'}: {us:7.1f} us per launch; in-kernel clock "
          f"{np.median(c[ok] / w[ok]) * 0.1:5.3f} GHz (main loop {np.median(w[ok]) * 0.01:6.2f} us = {np.median(c[ok]):8.0f} cycles); "
          f"sysfs sclk {t['sclk_mhz']['mean']:.0f} MHz, {t['power_w']['mean']:.0f} W", flush=True)
