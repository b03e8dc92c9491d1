"""The RCCL code path on ONE card.  The GPU box has a single MI355X, so a multi-rank RCCL job cannot run there; what
can run is everything RCCL-specific except the inter-GPU transport: ``backend="nccl"`` (= librccl on ROCm) with a
world of one rank, with the single-process early returns of cartnet_amd.distributed switched off
(CARTNET_DIST_FORCE=1).  Communicator creation, the all-reduce of the flat 10 MB gradient buffer on the device, the
barrier, the MAX / MIN reductions of the replica check and Adam with ``grad_scale`` all execute through librccl
exactly as they do on 8 GPUs.  Runs in a child process so the forced environment never leaks into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from cartnet_amd import distributed as cdist
from cartnet_amd.config import cfg
from cartnet_amd.model import CartNet
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_batch

rank, world, local = cdist.init_from_env()                 # CARTNET_DIST_FORCE=1 -> init_process_group("nccl"), world 1
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
cfg.radius = 5.0
torch.manual_seed(0)
dev = torch.device("cuda", local)
model = CartNet(256, 64, 4).to(dev).train()
opt = FlatAdam(model, lr=1e-3)
assert opt.flat_grad.numel() == 2498438                    # the 9.99 MB buffer of SURVEY.md 2b
b = make_batch(4, 194, first=31000).to(dev)
pred, true = model(b)
(pred - true).abs().mean().backward()
g_before = opt.flat_grad.clone()
p_before = opt.flat_param.clone()
scale = cdist.all_reduce_gradients(opt.flat_grad)          # RCCL all-reduce (SUM) of the flat gradient, in place
torch.cuda.synchronize()
assert scale == 1.0
same = bool(torch.equal(opt.flat_grad, g_before))          # SUM over one rank is the identity, bit for bit
# the same gradient through the BUCKETED path (CartnetGradReadyFn -> distributed.GradSync): six asynchronous RCCL all-reduces
# queued from inside cartnet_model_backward on the weight-gradient stream, joined by finish(); bit for bit the flat result
opt.zero_grad()
sync = cdist.GradSync(opt.flat_grad, measure=True)
seen = []
_ob = sync.bucket
def _rec(lo, hi):
    seen.append((int(lo), int(hi)))
    _ob(lo, hi)
sync.bucket = _rec
model.grad_sync = sync
b2 = make_batch(4, 194, first=31000).to(dev)
pred2, true2 = model(b2)
(pred2 - true2).abs().mean().backward()
model.grad_sync = None
n_works = len(sync.works)
scale_b = sync.finish()
torch.cuda.synchronize()
bucketed = {"same": bool(torch.equal(opt.flat_grad, g_before)), "scale": scale_b, "works": n_works,
            "order_ok": seen == [tuple(x) for x in model.grad_bucket_order()],
            "covers": sorted(seen)[0][0] == 0 and sum(h - l for l, h in seen) == opt.flat_grad.numel() and
                      all(a[1] == b[0] for a, b in zip(sorted(seen), sorted(seen)[1:])),
            "exposed_ms": sync.exposed_ms()}
opt.zero_grad()
opt.flat_grad.copy_(g_before)
opt.step(scale)
cdist.barrier()
t = cdist.max_over_ranks(3.25, dev)
cdist.assert_replicas_in_sync(model)                       # MAX and MIN all-reduce of the parameter checksums
cdist.broadcast_buffers(model)
moved = float((opt.flat_param - p_before).abs().max().item())
# sync-BatchNorm: 16 + 16 RCCL all-reduces of 2D+1 doubles inside forward / backward; over one rank the sums are unchanged,
# so prediction and gradient must equal the per-rank-statistics run to rounding (the finalisers take another path)
preds, grads = [], []
for sync in (False, True):
    model.sync_batchnorm = sync
    opt.zero_grad()
    bb = make_batch(4, 194, first=31000).to(dev)
    pred, true = model(bb)
    (pred - true).abs().mean().backward()
    preds.append(pred.detach().clone()); grads.append(opt.flat_grad.clone())
# the same exchange under the bf16 mode with bf16 storage (the statistics are taken before the output is rounded)
model.gemm_precision, model.half_storage, model.sync_batchnorm = 2, True, True
opt.zero_grad()
bb = make_batch(4, 194, first=31000).to(dev)
pred, true = model(bb)
(pred - true).abs().mean().backward()
half_ok = bool(torch.isfinite(pred).all() and torch.isfinite(opt.flat_grad).all() and
               (pred - preds[0]).abs().max().item() < 3e-2 * preds[0].abs().max().item())
model.gemm_precision, model.half_storage, model.sync_batchnorm = 0, False, False
# train_epoch with the reference recipe carried as BatchNorm groups (main.py --fused_accumulation: micro-batches of 4 inside
# a batch of 8), three optimiser steps, every gradient all-reduce through librccl (train/train.py:186-189)
from cartnet_amd.data import DataLoader
from cartnet_amd.synthetic import make_crystal
from cartnet_amd.train import train_epoch
cfg.loss = "MAE"
model.bn_group_size = 4
items = [make_crystal(33000 + i, 20 + (i % 5)) for i in range(24)]
loader = DataLoader(items, 8, shuffle=True, seed=1)
calls = []                                                 # (elements all-reduced, buckets) per optimiser step
_GS = cdist.GradSync
class _Counting(_GS):
    def bucket(self, lo, hi):
        self._n = getattr(self, "_n", 0) + (hi - lo)
        self._k = getattr(self, "_k", 0) + 1
        super().bucket(lo, hi)
    def finish(self):
        calls.append((self._n, self._k))
        self._n = self._k = 0
        return super().finish()
cdist.GradSync = _Counting
p0 = opt.flat_param.clone()
stats = train_epoch(loader, model, opt, batch_accumulation=1, device=dev)
cdist.GradSync = _GS
cdist.assert_replicas_in_sync(model)
model.bn_group_size = 0
fused = {"steps": len(calls), "per_step": sorted(set(calls)), "graphs": stats["graphs"], "finite": bool(torch.isfinite(opt.flat_param).all()),
         "moved": float((opt.flat_param - p0).abs().max().item()), "mae": stats["mae"]}
sync_pred = float((preds[0] - preds[1]).abs().max().item() / preds[0].abs().max().item())
sync_grad = float((grads[0] - grads[1]).abs().max().item() / grads[0].abs().max().item())
print(json.dumps({"backend": dist.get_backend(), "same": same, "max": t, "moved": moved,
                  "grad_norm": float(g_before.norm().item()), "sync_pred": sync_pred, "sync_grad": sync_grad,
                  "half_sync_ok": half_ok, "fused": fused, "bucketed": bucketed}), flush=True)
dist.destroy_process_group()
'''


def test_rccl_world_of_one_runs_every_collective_of_the_training_step(tmp_path):
    env = dict(os.environ, CARTNET_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CARTNET_DIST_BACKEND", None)
    script = tmp_path / "rccl_child.py"
    script.write_text(_CHILD)
    out = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["same"] is True and d["max"] == 3.25
    assert d["grad_norm"] > 0 and 0 < d["moved"] <= 1.1e-3            # one Adam step at lr 1e-3 moves each weight by <= lr
    assert d["sync_pred"] <= 1e-6 and d["sync_grad"] <= 1e-5 and d["half_sync_ok"] is True
    f = d["fused"]                     # three optimiser steps of train_epoch, each all-reducing the whole buffer in 6 buckets
    assert f["steps"] == 3 and f["graphs"] == 24 and f["finite"] and 0 < f["moved"] <= 3.3e-3 and f["mae"] == f["mae"]
    assert f["per_step"] == [[2498438, 6]]
    bk = d["bucketed"]                 # head, layers 3..0, encoder: six RCCL all-reduces queued from inside backward
    assert bk["same"] is True and bk["scale"] == 1.0 and bk["works"] == 6 and bk["order_ok"] and bk["covers"]
    assert bk["exposed_ms"] is not None and bk["exposed_ms"] >= 0.0


def test_bench_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with no outer launcher and WORLD_SIZE unset: bench.py spawns the two ranks itself
    (a child torch.distributed.run, before any GPU call; here they share the card over gloo), relays rank 0's single
    JSON line and exits with the child's status.  A world that disagrees with --gpus is a hard error."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CARTNET_DIST_FORCE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "2",
           "--graphs", "8"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "graph-sharded dp2" and d["value"] > 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         cwd=tmp_path, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)
