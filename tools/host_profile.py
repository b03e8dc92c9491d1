"""Host time of the C-side sequencing of one configs[2] training step, by RUN(...) statement of csrc/model.hip.
Needs a diagnostic library:  tools/build_variant.sh hostprof "-DCN_HOST_PROFILE" model.hip
and CARTNET_LIB=cartnet_amd/libcartnet_hip_hostprof.so.  Every step is synchronised first (idle queue)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cartnet_amd import lib as _lib, train as ctrain
from cartnet_amd.train import compute_loss
from cartnet_amd.config import cfg
from cartnet_amd.data import Batch
from cartnet_amd.model import CartNet
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_crystal
cfg.radius = 5.0
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)
sizes = torch.randint(2, 21, (64,), generator=gen).tolist()
base = Batch.from_data_list([make_crystal(5000 + i, n, adp=False) for i, n in enumerate(sizes)]).to(dev)
model = CartNet(256, 64, 4, temperature=False, cholesky=False).to(dev).train()
model.gemm_precision, model.half_storage = 2, True
opt = FlatAdam(model, lr=1e-3)
opt.direct_grads = True
L = _lib.load()
def fresh():
    b = base.clone(); b.num_graphs = base.num_graphs; b._cartnet_layout = None; b._cartnet_mask_index = None
    return b
T = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0}
def step(b):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pred, true = model(b)
    t1 = time.perf_counter(); loss = compute_loss(pred, true)[0]
    t2 = time.perf_counter(); ctrain.backward(loss)
    t3 = time.perf_counter(); opt.step(1.0); opt.zero_grad()
    t4 = time.perf_counter()
    T["fwd"] += t1 - t0; T["loss"] += t2 - t1; T["bwd"] += t3 - t2; T["opt"] += t4 - t3
CT = {}
def wrap(name):
    orig = getattr(L, name)
    def f(*a):
        t = time.perf_counter(); r = orig(*a); CT[name] = CT.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(L, name, f)
for nm in ("cartnet_model_forward", "cartnet_model_backward", "cartnet_workspace_bytes", "cartnet_loss_fwd", "cartnet_loss_bwd",
           "cartnet_adam_step"):
    wrap(nm)
bs = [fresh() for _ in range(60)]
for b in bs[:10]: step(b)
for k in T: T[k] = 0.0
fn = L.cartnet_debug_host_profile
fn.argtypes = [ctypes.c_int32]; fn.restype = ctypes.c_int
torch.cuda.synchronize()
fn(10)
CT.clear()
for b in bs[10:60]: step(b)
torch.cuda.synchronize()
print({k: round(1e3 * v / 50, 3) for k, v in T.items()}, "ms per step (python-side, idle queue)")
print({k: round(1e6 * v / 50, 1) for k, v in CT.items()}, "us per step inside the C entry points")
fn(50)
