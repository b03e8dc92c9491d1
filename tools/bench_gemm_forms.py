"""Every edge-sized GEMM form one training step of the benchmark batch launches (fp32 MFMA or bf16x3), timed ALONE on the
chip and warm: 60 warm launches, then 100 between two events.  Forms and operands as csrc/model.hip issues them
(weights as DMA images).  Usage: python tools/bench_gemm_forms.py [precision]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
E = int(os.environ.get("E", 177140)); D = 256; N = E // 14
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dev)
PEAK = 157.3 if prec == 0 else 2500.0 / 6
make = ops.pack_b if prec == 0 else ops.split_b

def timeit(name, fn, flops, warm=60, iters=100):
    if only and not any(o in name for o in only):
        return
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print(f"{name:58s} {us:8.1f} us  {flops/us/1e6:7.1f} TF/s  {flops/us/1e6/PEAK:6.3f} of peak", flush=True)

e = rnd(E, D); pre = rnd(E, 2*D); gs = rnd(E, 2*D); act = torch.empty(E, 2*D, device=dev)
out2 = torch.empty(E, 2*D, device=dev); out1 = torch.empty(E, D, device=dev)
W0g, W0a = rnd(D, 3*D, sc=0.05), rnd(D, 3*D, sc=0.05)
W2g, W2a = rnd(D, D, sc=0.05), rnd(D, D, sc=0.05)
b2g, b2a = rnd(D), rnd(D)
Pn = rnd(N, 4*D)
tgt = torch.sort(torch.randint(0, N, (E,), generator=g)).values.to(torch.int32).to(dev)
src = torch.randint(0, N, (E,), generator=g).to(torch.int32).to(dev)
tiles = ops.gemm_tiles_m(E)
cs = torch.empty(tiles*2*D, dtype=torch.float64, device=dev); cq = torch.empty_like(cs)
F = 2.0*E*D*D*2

# forward operands are W^T ([in, out]); the model transposes the weights once per step
T = lambda w: w.t().contiguous()
W2gT, W2aT = T(W2g), T(W2a)
W0geT, W0aeT = T(W0g[:, 2*D:]), T(W0a[:, 2*D:])
img_gs = make([W2gT, W2aT]); img_pre = make([W0geT, W0aeT])
timeit("nn256 plain x2 (calibration)", lambda: ops.gemm([gs[:, :D], gs[:, D:]], [W2gT, W2aT], [out2[:, :D], out2[:, D:]],
       b_kstrided=True, b_split=img_gs, precision=prec), F)
timeit("GEMM1: x2 + node-term gather (nn128 at fp32)", lambda: ops.gemm([e, e], [W0geT, W0aeT], [out2[:, :D], out2[:, D:]],
       b_kstrided=True, b_split=img_pre, precision=prec, gather_i=[Pn[:, :D], Pn[:, D:2*D]], gather_j=[Pn[:, 2*D:3*D], Pn[:, 3*D:]],
       tgt=tgt, src=src), F)
timeit("GEMM2: x2 silu(A) + bias + fp64 stats", lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], [out2[:, :D], out2[:, D:]],
       b_kstrided=True, b_split=img_gs, precision=prec, a_act=True, bias=[b2g, b2a], colsum=[cs[:tiles*D], None], colsq=[cq[:tiles*D], None]), F)
timeit("GEMM2: same + silu(A) written (a_act_out)", lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], [out2[:, :D], out2[:, D:]],
       b_kstrided=True, b_split=img_gs, precision=prec, a_act=True, bias=[b2g, b2a], colsum=[cs[:tiles*D], None], colsq=[cq[:tiles*D], None],
       a_act_out=[act[:, :D], act[:, D:]]), F)
timeit("GEMM2: x2 silu(A) + bias, no stats", lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], [out2[:, :D], out2[:, D:]],
       b_kstrided=True, b_split=img_gs, precision=prec, a_act=True, bias=[b2g, b2a]), F)
# backward operands are W itself ([out, in] read as [K = out, N = in])
img_dpre = make([W2g, W2a])
timeit("dpre: x2 * silu'(pre)", lambda: ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [out2[:, :D], out2[:, D:]],
       b_kstrided=True, b_split=img_dpre, precision=prec, dact=[pre[:, :D], pre[:, D:]]), F)
img_de = torch.cat(make([W0g[:, 2*D:], W0a[:, 2*D:]]))
timeit("dE: folded K=512 + residual", lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W0g[:, 2*D:], W0a[:, 2*D:]], out1,
       b_kstrided=True, segments=True, resid=e, b_split_folded=img_de, precision=prec), F)
timeit("dE: same, no residual", lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W0g[:, 2*D:], W0a[:, 2*D:]], out1,
       b_kstrided=True, segments=True, b_split_folded=img_de, precision=prec), F)
# round 5: the two epilogue fusions next to the passes they replace (fp32 only)
if prec == 0:
    rowptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(tgt.long(), minlength=N), 0).to(torch.int32)
    dPn = torch.empty(N, 4*D, device=dev)
    timeit("pass: segment_sum by target [E, 2D]", lambda: ops.segment_sum(out2, rowptr, None, dPn[:, :2*D]), F)
    # by source through a CSC permutation of a random graph with the same degrees + both in one launch
    order = torch.argsort(src.long(), stable=True).to(torch.int32)
    colptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    colptr[1:] = torch.cumsum(torch.bincount(src.long(), minlength=N), 0).to(torch.int32)
    def two():
        ops.segment_sum(out2, rowptr, None, dPn[:, :2*D]); ops.segment_sum(out2, colptr, order, dPn[:, 2*D:])
    timeit("pass: segment_sum by target, then by source (2 launches)", two, F)
    class LP: pass
    lp = LP(); lp.E, lp.N, lp.rowptr, lp.colptr, lp.perm = E, N, rowptr, colptr, order
    timeit("pass: segment_sum_pair (1 launch)", lambda: ops.segment_sum_pair(out2, lp, dPn[:, :2*D], dPn[:, 2*D:]), F)
    env = torch.rand(E, generator=g).to(dev); mr = torch.cat([rnd(D, sc=0.1), 1.0 + torch.rand(D, generator=g).to(dev)]).contiguous()
    gam, bet = rnd(D), rnd(D)
    ca = torch.empty(tiles*D, dtype=torch.float64, device=dev); cb = torch.empty_like(ca)
    timeit("dE: folded K=512 + residual + gate stats (gst_*)", lambda: ops.gemm([pre[:, :D], pre[:, D:]], [W0g[:, 2*D:], W0a[:, 2*D:]], out1,
           b_kstrided=True, segments=True, resid=e, b_split_folded=img_de, precision=prec, colsum=ca, colsq=cb,
           gate_stats=(gs[:, :D], env, mr, gam, bet)), F)
    class L: pass
    lay = L(); lay.E, lay.N, lay.rowptr = E, N, rowptr
    daggr = rnd(N, D); npg = ops.gate_nparts(N)
    pa = torch.empty(npg*D, dtype=torch.float64, device=dev); pb = torch.empty_like(pa)
    timeit("pass: gate_scatter_bwd_stats", lambda: ops.gate_scatter_bwd_stats(gs, e, daggr, env, lay, mr, gam, bet, pa, pb), F)
    aggr = torch.empty(N, D, device=dev); eo = torch.empty(E, D, device=dev); bcb = torch.empty(N, 2*D, device=dev)
    timeit("pass: gate_scatter_fwd", lambda: ops.gate_scatter_fwd(gs, e, env, lay, mr, gam, bet, eo, aggr, pa, pb), F)
    timeit("pass: gate_scatter_fwd + bc", lambda: ops.gate_scatter_fwd(gs, e, env, lay, mr, gam, bet, eo, aggr, pa, pb, bc=bcb), F)
# encoder
We2 = rnd(D, 2*D, sc=0.05); be2 = rnd(D)
We2T = T(We2); img_e2 = make([We2T]); img_e2b = make([We2])
e0pre = torch.empty(E, D, device=dev)
timeit("enc GEMM2: K=512 N=256 silu(A) + silu out + cpre + act_out", lambda: ops.gemm(pre, We2T, out1, b_kstrided=True, b_split=img_e2,
       precision=prec, a_act=True, out_act=True, bias=be2, cpre=e0pre, a_act_out=act), F)
timeit("enc dhe: K=256 N=512 * silu' + colsum", lambda: ops.gemm(e, We2, out2, b_kstrided=True, b_split=img_e2b, precision=prec,
       dact=pre, colsum=cs), F)
feat = rnd(E, 80); We0T = rnd(80, 2*D, sc=0.05); img_e0 = make([We0T]); be0 = rnd(2*D)
timeit("enc GEMM1: K=80 N=512 + bias", lambda: ops.gemm(feat, We0T, out2, b_kstrided=True, b_split=img_e0, precision=prec, bias=be0),
       2.0*E*80*2*D)
# weight gradients (reduction over the E rows, split-K as the model picks it: 512 workgroups)
S = 128
slabs = [torch.empty(S*D, D, device=dev) for _ in range(2)]
timeit("dW: x2 dY^T X, split-K 128", lambda: ops.gemm([gs[:, :D], gs[:, D:]], [act[:, :D], act[:, D:]], slabs, a_kstrided=True,
       b_kstrided=True, splitk=S, precision=prec), F)
timeit("dW1e: x2 dY^T e (X ld = D)", lambda: ops.gemm([gs[:, :D], gs[:, D:]], [e, e], slabs, a_kstrided=True,
       b_kstrided=True, splitk=S, precision=prec), F)
slab1 = torch.empty(S*D, 2*D, device=dev)
timeit("enc dW2: dY^T X, M=256 N=512", lambda: ops.gemm(e, pre, slab1, a_kstrided=True, b_kstrided=True, splitk=S, precision=prec), F)
slab0 = torch.empty(256*2*D, 80, device=dev)
timeit("enc dW0: dhe^T feat, M=512 N=80, split-K 256", lambda: ops.gemm(pre, feat, slab0, a_kstrided=True, b_kstrided=True, splitk=256,
       precision=prec), 2.0*E*80*2*D)
# fixed cost per tile: the same output [E, 512] from K = 16 .. 256 (one K-step .. sixteen)
for K_ in (16, 32, 80, 128, 256):
    fk = rnd(E, K_); WkT = rnd(K_, 2*D, sc=0.05); imgk = make([WkT])
    timeit(f"thin: K={K_} N=512, no bias", lambda: ops.gemm(fk, WkT, out2, b_kstrided=True, b_split=imgk, precision=prec), 2.0*E*K_*2*D)
