"""the forward of ONE CartNet layer at BASELINE configs[2] sizes (64 crystals of 2..20 atoms: N = 736,
E = 9,970; D = 256, precision 2) as ONE cooperative launch (csrc/coop_layer.hip) -- checked against the fp64 oracle's
cartnet_layer, timed alone on the chip (events around back-to-back launches), next to today's chain of that layer in the
model's own forward (kernel trace: tools/exp_small_batch_layer.sh).   Usage: python tools/exp_small_batch_layer.py   (GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import lib as _l
from cartnet_amd.data import Batch
from cartnet_amd.model import make_state_dict
from cartnet_amd.synthetic import make_crystal
from oracle import cartnet_ref as orc

dev = torch.device("cuda:0")
D = 256
gen = torch.Generator().manual_seed(7)
sizes = torch.randint(2, 21, (64,), generator=gen).tolist()
batch = Batch.from_data_list([make_crystal(5000 + i, n, adp=False) for i, n in enumerate(sizes)])
N, E = int(batch.x.shape[0]), int(batch.edge_index.shape[1])
print("N", N, "E", E, flush=True)
sd = make_state_dict(D, 64, 4, seed=5)
g = torch.Generator().manual_seed(11)
x = torch.randn(N, D, generator=g)
e = torch.randn(E, D, generator=g)
src, tgt = batch.edge_index[0], batch.edge_index[1]
deg = torch.bincount(tgt, minlength=N)
assert int(deg.min()) >= 1, "the prototype needs every atom to have an incoming edge"
rowptr = torch.zeros(N + 1, dtype=torch.int64); rowptr[1:] = torch.cumsum(deg, 0)
l = 1
# fp64 oracle of the layer (training-mode BatchNorm, envelope on); operands rounded to bf16 as precision 2 does
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ref_x, ref_e = orc.cartnet_layer(sd64, l, x.double(), e.double(), batch.edge_index, batch.cart_dist.double(), 5.0, True, True, {})
Wg, Wa = sd[f"layers.{l}.MLP_gate.0.weight"], sd[f"layers.{l}.MLP_aggr.0.weight"]
bf = lambda t: t.to(torch.bfloat16).contiguous().to(dev)
wn = bf(torch.cat([Wg[:, :D], Wa[:, :D], Wg[:, D:2 * D], Wa[:, D:2 * D]]))
w1e = bf(torch.cat([Wg[:, 2 * D:], Wa[:, 2 * D:]]))
w2 = bf(torch.cat([sd[f"layers.{l}.MLP_gate.2.weight"], sd[f"layers.{l}.MLP_aggr.2.weight"]]))
b1 = torch.cat([sd[f"layers.{l}.MLP_gate.0.bias"], sd[f"layers.{l}.MLP_aggr.0.bias"]]).to(dev)
b2 = torch.cat([sd[f"layers.{l}.MLP_gate.2.bias"], sd[f"layers.{l}.MLP_aggr.2.bias"]]).to(dev)
f = lambda k: sd[f"layers.{l}.{k}"].float().to(dev)
bn1w, bn1b, bn2w, bn2b = f("norm.weight"), f("norm.bias"), f("norm2.weight"), f("norm2.bias")
env = (0.5 * (torch.cos(torch.pi * batch.cart_dist / 5.0) + 1.0) * (batch.cart_dist < 5.0)).float().to(dev)
xd, ed = x.to(dev), e.to(dev)
tgt32, src32, rp32 = tgt.to(torch.int32).to(dev), src.to(torch.int32).to(dev), rowptr.to(torch.int32).to(dev)
L = _l.load()
Pn = torch.empty(N, 4 * D, device=dev); pre = torch.empty(E, 2 * D, device=dev); gs = torch.empty(E, 2 * D, device=dev)
e_out = torch.full((E, D), float("nan"), device=dev); aggr = torch.empty(N, D, device=dev)
x_out = torch.full((N, D), float("nan"), device=dev)
work = torch.empty(L.cartnet_coop_layer_workspace_floats(N, E), device=dev)
bar = torch.zeros(3 * 8 * 32, dtype=torch.int32, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
epoch = [0]
def run():
    st = torch.cuda.current_stream().cuda_stream
    rc = L.cartnet_coop_layer_fwd(xd.data_ptr(), ed.data_ptr(), tgt32.data_ptr(), src32.data_ptr(), rp32.data_ptr(), env.data_ptr(),
                                  wn.data_ptr(), w1e.data_ptr(), w2.data_ptr(), b1.data_ptr(), b2.data_ptr(), bn1w.data_ptr(),
                                  bn1b.data_ptr(), bn2w.data_ptr(), bn2b.data_ptr(), N, E, 1e-5, Pn.data_ptr(), pre.data_ptr(),
                                  gs.data_ptr(), e_out.data_ptr(), aggr.data_ptr(), x_out.data_ptr(), work.data_ptr(),
                                  bar.data_ptr(), epoch[0], status.data_ptr(), C.c_void_p(st))
    assert rc == 0, L.cartnet_last_error()
    epoch[0] += 1
run()
torch.cuda.synchronize()
assert int(status.item()) == 0, "a grid barrier gave up"
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
print(f"against the fp64 oracle (bf16 operands: the model's precision-2 tests allow 3e-2): x_out {rel(x_out, ref_x):.2e}, e_out {rel(e_out, ref_e):.2e}")
assert rel(x_out, ref_x) < 3e-2 and rel(e_out, ref_e) < 3e-2
if "--check-only" in sys.argv:
    raise SystemExit(0)
for _ in range(200): run()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(1e3 * e0.elapsed_time(e1) / 500)
assert int(status.item()) == 0
print("one cooperative launch per layer forward: " + " / ".join(f"{t:.1f}" for t in ts) + " us (events around 500 back-to-back launches)")
# phase stamps of the last launch (100 MHz): entry, [end of phase k, after barrier k] x 4, end
import numpy as np
tiles_e = (E + 63) // 64
off = tiles_e * 2 * D + tiles_e * 65 * D + 32 * 2 * D
st = work[off:off + 256 * 32].cpu().numpy().view(np.uint64).reshape(256, 16)[:, :10].astype(np.int64)
t0 = st[:, 0].min()
names = ["P1 node terms", "barrier 1", "P2 edge MLPs", "barrier 2", "P3 gate + run sums", "barrier 3", "P4 atom sums", "barrier 4", "P5 node update"]
print("phases, us (median over workgroups that do work in the phase / max; barrier = from a workgroup's arrival to its release):")
for k, nm in enumerate(names):
    d = (st[:, k + 1] - st[:, k]) * 0.01
    print(f"  {nm:22s} median {np.median(d):6.1f}   max {d.max():6.1f}   (all workgroups past it at {((st[:, k + 1].max() - t0) * 0.01):6.1f} us)")
