"""RCCL with MORE than one rank.  ``backend="nccl"`` is librccl on ROCm; it needs one device per rank,
so the RCCL case of this file runs wherever at least two GPUs are visible -- the driver's 8-GPU node, a 2-GPU lease -- and
SKIPS cleanly on the one-card box.  The same child script always runs with two ranks sharing the card over gloo, so the
script itself (and everything in it that is not the transport) is exercised on every box.

What the two ranks do, in one process group:
  1. one step with the gradient all-reduce in BUCKETS under backward (six asynchronous all-reduces queued from inside
     cartnet_model_backward on the weight-gradient stream: include/cartnet_hip.h CartnetGradReadyFn, distributed.GradSync)
     against the flat all-reduce after backward: bit for bit (a + b has one order);
  2. sync-BatchNorm: 2 + 4 crystals on the two ranks against one process on all 6 (predictions, gradient, running stats);
  3. ``train_epoch`` over a ragged set where one rank runs an EMPTY step (it must issue the same collectives, in the
     same order, with a zero gradient), every rank the same number of optimiser steps, ``assert_replicas_in_sync``;
  4. the replica check itself: a rank whose parameters were nudged is reported on every rank.
The reference has no distributed code at all (scripts/train_cartnet_adp.sh:3-14 starts N independent
processes); the accumulation boundary being synchronised is train/train.py:186-189."""
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
from cartnet_amd.data import Batch, DataLoader
from cartnet_amd.model import CartNet, make_state_dict
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_crystal
from cartnet_amd.train import train_epoch

rank, world, local = cdist.init_from_env()
want = sys.argv[3]
if world > 1:
    assert dist.get_backend() == want, (dist.get_backend(), want)
cfg.radius = 5.0
cfg.loss = "MAE"
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
out = {"world": world, "rank": rank, "device": str(dev), "ranks_seen": cdist.ranks_seen(dev)}


def model(L=2, seed=43):
    m = CartNet(64, 16, L)
    m.load_state_dict(make_state_dict(64, 16, L, seed=seed))
    return m.to(dev).train()


# ---- 1. bucketed all-reduce under backward == flat all-reduce behind it
m = model(3, 47)
opt = FlatAdam(m, lr=1e-3)
items = [make_crystal(9900 + 7 * rank + i, 10 + 3 * i + rank) for i in range(3)]      # every rank its own crystals
for mode in ("flat", "bucketed"):
    opt.zero_grad()
    b = Batch.from_data_list(items).to(dev)
    sync = cdist.GradSync(opt.flat_grad, measure=True) if (mode == "bucketed" and world > 1) else None
    m.grad_sync = sync
    pred, true = m(b)
    (pred - true).abs().mean().backward()
    m.grad_sync = None
    scale = sync.finish() if sync is not None else cdist.all_reduce_gradients(opt.flat_grad)
    torch.cuda.synchronize()
    out[mode] = opt.flat_grad.cpu().clone()
    out[mode + "_scale"] = scale
    if sync is not None:
        out["buckets"] = sync.buckets_seen
        out["exposed_ms"] = sync.exposed_ms()
del m, opt

# ---- 2. sync-BatchNorm: shards of 2 and 4 crystals == one process on the 6
items = [make_crystal(9700 + i, 9 + 4 * i) for i in range(6)]
m = model(2, 43)
m.sync_batchnorm = world > 1
opt = FlatAdam(m, lr=1e-3)
mine = items[:2] if (world == 2 and rank == 0) else (items[2:] if world == 2 else items)
b = Batch.from_data_list(mine).to(dev)
pred, true = m(b)
(pred - true).abs().sum().backward()                                   # a SUM: the union loss is the sum of the shard losses
cdist.all_reduce_gradients(opt.flat_grad)
out["sync"] = {"grad": opt.flat_grad.cpu().clone(), "pred": pred.detach().cpu(),
               "bufs": {k: v.cpu() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}}
del m, opt

# ---- 3. train_epoch with an empty step on one rank, bucketed all-reduce at every boundary
gen = torch.Generator().manual_seed(77)
sizes = torch.randint(3, 41, (5,), generator=gen).tolist()
items = [make_crystal(12000 + i, n) for i, n in enumerate(sizes)]
m = model(2, 43)
m.bn_group_size = 1          # every crystal its own BatchNorm group and loss term: the sum does not depend on the cut


class Recording(FlatAdam):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.seen = []

    def step(self, grad_scale=1.0):
        self.seen.append((self.flat_grad.clone(), grad_scale))
        super().step(grad_scale)


opt = Recording(m, lr=1e-3)
loader = DataLoader(items, 1, rank=rank, world_size=world)
batches = loader._batches()
stats = train_epoch(loader, m, opt, batch_accumulation=len(loader), device=dev)      # ONE optimiser step: the union's gradient
out["epoch"] = {"g": opt.seen[0][0].cpu(), "scale": opt.seen[0][1], "n_iter": len(loader), "graphs": stats["graphs"],
                "empty_steps": sum(1 for bb in batches if not bb), "crystals": sorted(j for bb in batches for j in bb)}
m2 = model(2, 43)
m2.bn_group_size = 1
opt2 = Recording(m2, lr=1e-3)
train_epoch(DataLoader(items, 1, rank=rank, world_size=world), m2, opt2, batch_accumulation=1, device=dev)
cdist.assert_replicas_in_sync(m2)
out["epoch"]["steps2"] = len(opt2.seen)
out["epoch"]["p"] = opt2.flat_param.detach().cpu()

# ---- 4. the replica check reports a diverged rank on EVERY rank
caught = False
if world > 1:
    if rank == 1:
        with torch.no_grad():
            opt2.flat_param[0] += 1.0
    try:
        cdist.assert_replicas_in_sync(m2)
    except RuntimeError as exc:
        caught = "diverged" in str(exc)
out["divergence_caught"] = caught
torch.save(out, os.path.join(sys.argv[2], f"m_{want}_w{world}_r{rank}.pt"))
if world > 1:
    cdist.barrier()
    dist.destroy_process_group()
'''


def _run(tmp_path, backend, port):
    import torch
    script = tmp_path / "multi_child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", CARTNET_DIST_BACKEND=backend)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CARTNET_DIST_FORCE"):
        env.pop(k, None)
    args = [ROOT, str(tmp_path), backend]
    one = subprocess.run([sys.executable, str(script)] + args, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)] + args,
                         env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    ref = torch.load(tmp_path / f"m_{backend}_w1_r0.pt")
    r = [torch.load(tmp_path / f"m_{backend}_w2_r{i}.pt") for i in range(2)]
    assert [x["ranks_seen"] for x in r] == [2, 2] and ref["ranks_seen"] == 1
    if backend == "nccl":
        assert r[0]["device"] != r[1]["device"]                       # one device per rank
    # 1. bucketed == flat, bit for bit, the same sum on both ranks, 3 layers + head + encoder = 5 buckets
    for x in r:
        assert x["buckets"] == 5 and x["flat_scale"] == x["bucketed_scale"] == 0.5
        assert torch.equal(x["bucketed"], x["flat"]) and torch.equal(x["bucketed"], r[0]["bucketed"])
        assert x["flat"].abs().max().item() > 0 and x["exposed_ms"] is not None and x["exposed_ms"] >= 0.0
    # 2. sync-BatchNorm
    s, s0, s1 = ref["sync"], r[0]["sync"], r[1]["sync"]
    pred2 = torch.cat([s0["pred"], s1["pred"]])
    assert (pred2 - s["pred"]).abs().max().item() <= 1e-5 * s["pred"].abs().max().item()
    assert torch.equal(s0["grad"], s1["grad"])
    assert (s0["grad"] - s["grad"]).abs().max().item() <= 3e-5 * s["grad"].abs().max().item()
    for k, v in s["bufs"].items():
        for x in (s0, s1):
            assert torch.equal(x["bufs"][k], v) if v.dtype == torch.int64 else torch.allclose(x["bufs"][k], v, rtol=1e-5, atol=1e-7), k
    # 3. the epoch: nothing dropped or doubled, one rank ran an empty step, same step count, same parameters, union gradient
    e, e0, e1 = ref["epoch"], r[0]["epoch"], r[1]["epoch"]
    assert e["crystals"] == list(range(5)) and sorted(e0["crystals"] + e1["crystals"]) == list(range(5))
    assert e0["graphs"] + e1["graphs"] == 5 and e0["n_iter"] == e1["n_iter"] == e0["steps2"] == e1["steps2"] == 3
    assert e0["empty_steps"] + e1["empty_steps"] == 1
    assert e0["scale"] == e1["scale"] == 0.5 and e["scale"] == 1.0
    assert torch.equal(e0["g"], e1["g"]) and torch.equal(e0["p"], e1["p"])
    assert (e0["g"] - e["g"]).abs().max().item() <= 3e-5 * e["g"].abs().max().item()
    # 4. divergence is seen by both ranks
    assert r[0]["divergence_caught"] is True and r[1]["divergence_caught"] is True


def test_two_ranks_sharing_the_card_over_gloo(tmp_path):
    _run(tmp_path, "gloo", 29571)


def test_two_ranks_over_rccl(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:                 # (counting devices does not initialise the GPU on this image)
        pytest.skip("RCCL needs one device per rank: fewer than two GPUs are visible")
    _run(tmp_path, "nccl", 29572)


def test_bench_two_ranks_over_rccl(tmp_path):
    """The driver's N = 2 line on real devices: ranks_seen == 2, per-rank step times, the exposed all-reduce time."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than two GPUs are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CARTNET_DIST_FORCE",
                                                            "CARTNET_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--preroll-steps",
           "10", "--sustain-seconds", "0", "--no-x3-pass", "--no-recipe-pass", "--no-calibration"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "nccl"
    assert 0 < d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.001
    assert d["allreduce_exposed_ms_per_step"] is not None
