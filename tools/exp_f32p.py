"""The persistent fp32 GEMM kernel (csrc/gemm_f32p.h, CartnetGemmArgs.tile_policy = 3) next to the shipped kernels
(tile_policy = 256) on the edge-sized forms of one training step: results compared bit for bit, then both timed alone on the
chip (60 warm launches, 100 between two events), A B A B.   Usage: python tools/exp_f32p.py [E] [forms,comma,separated]
(GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops

dev = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 177140
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
D = 256
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dev)

def timeit(fn, warm=60, iters=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters

pre = rnd(E, 2 * D); gs = rnd(E, 2 * D)
W2g, W2a = rnd(D, D, sc=0.05), rnd(D, D, sc=0.05)
b2g, b2a = rnd(D), rnd(D)
T = lambda w: w.t().contiguous()
W2gT, W2aT = T(W2g), T(W2a)
img_gs = ops.pack_b([W2gT, W2aT]); img_dpre = ops.pack_b([W2g, W2a])
Wd = rnd(D, 2 * D, sc=0.05); img_dhe = ops.pack_b([Wd])
We2T = rnd(2 * D, D, sc=0.05); img_e2 = ops.pack_b([We2T])
img_fold = torch.cat(ops.pack_b([W2g, W2a]))
NA = max(2, E // 14)               # atoms: ~14 edges each, targets sorted (CSR order) as in the model
Pn = rnd(NA, 4 * D)
tgt = torch.sort(torch.randint(0, NA, (E,), generator=g)).values.to(torch.int32).to(dev)
src = (tgt.cpu() // 194 * 194 + torch.randint(0, 194, (E,), generator=g)).clamp(max=NA - 1).to(torch.int32).to(dev)   # same crystal
tiles = ops.gemm_tiles_m(E)
env = torch.rand(E, generator=g).to(dev)
mean_rstd = torch.cat([gs[:, D:].mean(0), torch.rsqrt(gs[:, D:].var(0, unbiased=False) + 1e-5)]).contiguous()
F = 2.0 * E * D * D * 2

def outputs():
    return dict(out=torch.full((E, 2 * D), float("nan"), device=dev), act=torch.full((E, 2 * D), float("nan"), device=dev),
                cs=torch.full((tiles * D,), float("nan"), dtype=torch.float64, device=dev),
                cq=torch.full((tiles * D,), float("nan"), dtype=torch.float64, device=dev),
                cs2=torch.full((tiles * 2 * D,), float("nan"), dtype=torch.float64, device=dev))

def form(name, o, pol):
    out = [o["out"][:, :D], o["out"][:, D:]]
    if name == "plain":
        ops.gemm([gs[:, :D], gs[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, tile_policy=pol)
    elif name == "bias":
        ops.gemm([gs[:, :D], gs[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, bias=[b2g, b2a], tile_policy=pol)
    elif name == "act":
        ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, a_act=True, bias=[b2g, b2a],
                 tile_policy=pol)
    elif name == "stats":
        ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, a_act=True, bias=[b2g, b2a],
                 colsum=[o["cs"], None], colsq=[o["cq"], None], tile_policy=pol)
    elif name == "stats_actout":
        ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, a_act=True, bias=[b2g, b2a],
                 colsum=[o["cs"], None], colsq=[o["cq"], None], a_act_out=[o["act"][:, :D], o["act"][:, D:]], tile_policy=pol)
    elif name == "dpre":
        ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], out, b_kstrided=True, b_split=img_dpre, dact=[pre[:, :D], pre[:, D:]],
                 tile_policy=pol)
    elif name == "dhe":       # the encoder's dhe form: * silu'(pre) + bias gradient (fp32 column sums), one group of N = 512
        ops.gemm(gs[:, :D], Wd, o["out"], b_kstrided=True, b_split=img_dhe, dact=pre, colsum=o["cs2"], tile_policy=pol)
    elif name == "act_actout":      # iComformer's second Linears: silu(A), bias, silu(A) written, no statistics
        ops.gemm([pre[:, :D], pre[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, a_act=True, bias=[b2g, b2a],
                 a_act_out=[o["act"][:, :D], o["act"][:, D:]], tile_policy=pol)
    elif name == "rbf352":          # iComformer's RBF branch: pre kept, softplus(pre) out (dact_kind = 1), two groups here
        ops.gemm([gs[:, :D], gs[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, bias=[b2g, b2a],
                 cpre=[o["act"][:, :D], o["act"][:, D:]], out_act=True, dact_kind=1, tile_policy=pol)
    elif name == "enc2":            # the edge encoder's second Linear: K = 512, N = 256, silu(A), pre kept, silu out, silu(A) written
        ops.gemm(pre, We2T, o["out"][:, :D], b_kstrided=True, b_split=img_e2, a_act=True, out_act=True, bias=b2g,
                 cpre=o["out"][:, D:], a_act_out=o["act"], tile_policy=pol)
    elif name == "k512resid":       # two folded K-segments + residual (iComformer's d(rows) products; CartNet's dE without statistics)
        ops.gemm([pre[:, :D], pre[:, D:]], [W2g, W2a], o["out"][:, :D], b_kstrided=True, segments=True, resid=gs[:, :D],
                 b_split_folded=img_fold, tile_policy=pol)
    elif name == "de_gst":          # CartNet's dE: K = 512 + residual + the gate statistics of the layer below (shipped = the 128-wide kernel)
        ops.gemm([pre[:, :D], pre[:, D:]], [W2g, W2a], o["out"][:, :D], b_kstrided=True, segments=True, resid=gs[:, :D],
                 b_split=img_dpre, b_split_folded=img_fold, colsum=o["cs"], colsq=o["cq"],
                 gate_stats=(gs[:, D:], env, mean_rstd, b2g, b2a), tile_policy=128 if pol == 256 else pol)
    elif name == "gather":          # the layer's first product: bias + node terms by target / source atom
        ops.gemm([gs[:, :D], gs[:, D:]], [W2gT, W2aT], out, b_kstrided=True, b_split=img_gs, bias=[b2g, b2a],
                 gather_i=[Pn[:, :D], Pn[:, D:2 * D]], gather_j=[Pn[:, 2 * D:3 * D], Pn[:, 3 * D:]], tgt=tgt, src=src,
                 tile_policy=pol)
    else:
        raise SystemExit(f"unknown form {name}")

forms = ["plain", "bias", "act", "stats", "stats_actout", "dpre", "dhe", "act_actout", "rbf352", "enc2", "k512resid", "gather", "de_gst"]
for name in forms:
    if only and name not in only:
        continue
    ref, new = outputs(), outputs()
    form(name, ref, 256)
    form(name, new, 3)
    torch.cuda.synchronize()
    msg = []
    for k in ("out", "act", "cs", "cq", "cs2"):
        a, b = ref[k], new[k]
        if torch.isnan(a).all():
            if not torch.isnan(b).all():
                msg.append(f"{k}: WRITTEN by the new kernel only")
            continue
        same = torch.equal(a.nan_to_num(nan=12345.0), b.nan_to_num(nan=12345.0))      # (unwritten parts stay NaN in both)
        nan_new = int(torch.isnan(b).sum()) - int(torch.isnan(a).sum())
        err = float((a - b).abs().nan_to_num(nan=float("inf")).max())
        msg.append(f"{k}: {'bitwise equal' if same else f'DIFFERENT max|d|={err:.3e} nan_new={nan_new}'}")
    print(f"== {name}: " + "; ".join(msg), flush=True)
    o = outputs()
    for rep in range(2):
        t_old = timeit(lambda: form(name, o, 256))
        t_new = timeit(lambda: form(name, o, 3))
        print(f"   shipped {t_old:7.1f} us ({F/t_old/1e6/157.3:.3f})   persistent {t_new:7.1f} us ({F/t_new/1e6/157.3:.3f})", flush=True)
