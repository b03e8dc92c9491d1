"""Two data-parallel ranks of the main.py-equivalent on the GPU path (one process per rank under torch.distributed.run,
crystals sharded by rank, one gradient all-reduce per optimiser step, FlatAdam on every rank).  The GPU box has one
card, so both ranks share it and the exchange goes over gloo (CARTNET_DIST_BACKEND); RCCL only replaces the transport.
main.py ends with ``assert_replicas_in_sync``: the parameters of the two ranks must be identical."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_train_in_lockstep(tmp_path):
    env = dict(os.environ, CARTNET_DIST_BACKEND="gloo", CARTNET_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "main.py"), "--synthetic", "40", "--atoms", "20",
           "40", "--dim_in", "64", "--num_layers", "2", "--epochs", "2", "--batch", "4", "--batch_accumulation", "2",
           "--name", "ddp", "--augment", "--lr", "2e-3"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    epochs = [l for l in lines if "epoch" in l]
    assert len(epochs) == 2 and all(e["train_mae"] == e["train_mae"] for e in epochs)      # finite (not NaN)
    assert any("test" in l for l in lines)
    assert os.path.exists(tmp_path / "results" / "ddp" / "0" / "ckpt" / "best.ckpt")


def test_bench_contract_with_two_ranks(tmp_path):
    """bench.py under the driver's N > 1 launch line (here 2 ranks sharing the card over gloo): rank 0 prints ONE JSON
    line with the whole-job rate, n_gpus = 2, weak scaling, the roofline object, and cpu_baseline: null with a note (N = 1 only)."""
    env = dict(os.environ, CARTNET_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--graphs", "8", "--share-gpu"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["unit"] == "graphs/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    # round 6: the N > 1 line says that it has no CPU baseline, and why
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_note"]
    assert d["config"]["parallelism"] == "graph-sharded dp2"
    # round 5: what a multi-rank line says about itself
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert 0 < d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.001
    assert d["cold"]["steps"] == 3 and d["cold"]["warmup"] == 2 and d["cold"]["ms_per_step"] > 0
    assert d["untimed_steps_total"] == max(d["preroll_steps"], 5) + d["warmup"] + d["spinup"]


def test_divergence_is_detected():
    """The checksum comparison itself, single process: identical replicas pass (world size 1 is a no-op)."""
    import torch
    from cartnet_amd import distributed as cdist
    cdist.assert_replicas_in_sync(torch.nn.Linear(4, 4))


_GROUPS_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from cartnet_amd import distributed as cdist
from cartnet_amd.config import cfg
from cartnet_amd.data import Batch
from cartnet_amd.model import CartNet, make_state_dict
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_crystal
from cartnet_amd.train import grouped_loss

rank, world, local = cdist.init_from_env()
cfg.radius = 5.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
items = [make_crystal(9500 + i, 12 + 3 * i) for i in range(8)]
m = CartNet(64, 16, 2)
m.load_state_dict(make_state_dict(64, 16, 2, seed=41))
m = m.to(dev).train()
m.bn_group_size = 2
opt = FlatAdam(m, lr=1e-3)
mine = items[4 * rank:4 * rank + 4] if world == 2 else items           # two groups of 2 per rank / four groups in one
b = Batch.from_data_list(mine).to(dev)
pred, true = m(b)
grouped_loss(pred, true, b, 2)[0].backward()
cdist.all_reduce_gradients(opt.flat_grad)                              # SUM over ranks = the accumulation of all 4 micro-batches
torch.save(opt.flat_grad.cpu(), os.path.join(sys.argv[2], f"g_w{world}_r{rank}.pt"))
if world > 1:
    dist.destroy_process_group()
'''


def test_groups_make_data_parallel_runs_equal_to_the_accumulation_recipe(tmp_path):
    """With BatchNorm groups a micro-batch never sees crystals outside itself, so sharding the micro-batches over ranks
    changes nothing: the SUM-all-reduced gradient of 2 ranks x 2 groups equals the gradient of one process carrying all
    4 groups -- the reference's accumulation over 4 micro-batches (train/train.py:183-189) -- without any sync-BatchNorm."""
    import torch
    script = tmp_path / "groups_child.py"
    script.write_text(_GROUPS_CHILD)
    env = dict(os.environ, CARTNET_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29536", str(script), ROOT, str(tmp_path)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    g1 = torch.load(tmp_path / "g_w1_r0.pt")
    g2a, g2b = torch.load(tmp_path / "g_w2_r0.pt"), torch.load(tmp_path / "g_w2_r1.pt")
    assert torch.equal(g2a, g2b)
    assert (g2a - g1).abs().max().item() <= 2e-5 * g1.abs().max().item()


_SYNC_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from cartnet_amd import distributed as cdist
from cartnet_amd.config import cfg
from cartnet_amd.data import Batch
from cartnet_amd.model import CartNet, make_state_dict
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_crystal

rank, world, local = cdist.init_from_env()
cfg.radius = 5.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
items = [make_crystal(9700 + i, 9 + 4 * i) for i in range(6)]          # 9..29 atoms: the two shards differ in size
m = CartNet(64, 16, 2)
m.load_state_dict(make_state_dict(64, 16, 2, seed=43))
m = m.to(dev).train()
m.sync_batchnorm = world > 1
opt = FlatAdam(m, lr=1e-3)
mine = items[:2] if (world == 2 and rank == 0) else (items[2:] if world == 2 else items)   # 2 + 4 crystals
b = Batch.from_data_list(mine).to(dev)
pred, true = m(b)
(pred - true).abs().sum().backward()                                   # a SUM: the union loss is the sum of the shard losses
cdist.all_reduce_gradients(opt.flat_grad)
bufs = {k: v.cpu() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
torch.save({"grad": opt.flat_grad.cpu(), "pred": pred.detach().cpu(), "bufs": bufs},
           os.path.join(sys.argv[2], f"s_w{world}_r{rank}.pt"))
if world > 1:
    dist.destroy_process_group()
'''


def test_sync_batchnorm_makes_two_shards_equal_to_the_union_batch(tmp_path):
    """CartNet.sync_batchnorm: two ranks with 2 and 4 crystals, every BatchNorm's sums exchanged in
    forward and backward -> predictions, the SUM-all-reduced gradient and the running statistics equal those of one
    process on all 6 crystals (without it the shards' statistics differ and so does everything downstream)."""
    import torch
    script = tmp_path / "sync_child.py"
    script.write_text(_SYNC_CHILD)
    env = dict(os.environ, CARTNET_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(script), ROOT, str(tmp_path)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    ref = torch.load(tmp_path / "s_w1_r0.pt")
    r0, r1 = torch.load(tmp_path / "s_w2_r0.pt"), torch.load(tmp_path / "s_w2_r1.pt")
    pred2 = torch.cat([r0["pred"], r1["pred"]])
    assert pred2.shape == ref["pred"].shape
    assert (pred2 - ref["pred"]).abs().max().item() <= 1e-5 * ref["pred"].abs().max().item()
    assert torch.equal(r0["grad"], r1["grad"])
    assert (r0["grad"] - ref["grad"]).abs().max().item() <= 3e-5 * ref["grad"].abs().max().item()
    for k, v in ref["bufs"].items():
        for r in (r0, r1):
            if v.dtype == torch.int64:
                assert torch.equal(r["bufs"][k], v), k
            else:
                assert torch.allclose(r["bufs"][k], v, rtol=1e-5, atol=1e-7), k


_BUCKET_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from cartnet_amd import distributed as cdist
from cartnet_amd.config import cfg
from cartnet_amd.data import Batch
from cartnet_amd.model import CartNet, make_state_dict
from cartnet_amd.optim import FlatAdam
from cartnet_amd.synthetic import make_crystal

rank, world, local = cdist.init_from_env()
cfg.radius = 5.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
m = CartNet(64, 16, 3)
m.load_state_dict(make_state_dict(64, 16, 3, seed=47))
m = m.to(dev).train()
opt = FlatAdam(m, lr=1e-3)
items = [make_crystal(9900 + 7 * rank + i, 10 + 3 * i + rank) for i in range(3)]      # every rank its own crystals
out = {}
for mode in ("flat", "bucketed"):
    opt.zero_grad()
    b = Batch.from_data_list(items).to(dev)
    sync = cdist.GradSync(opt.flat_grad) if mode == "bucketed" else None
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
torch.save(out, os.path.join(sys.argv[2], f"b_w{world}_r{rank}.pt"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_all_reduce_under_backward_equals_the_flat_one(tmp_path, world):
    """the gradient all-reduce overlapped with backward.  cartnet_model_backward reports every bucket
    (head, layers L-1..0, encoder) on the weight-gradient stream; distributed.GradSync queues its all-reduce there.  Ranks
    share the one card and exchange over gloo: the SUM over ranks of every element must be what ONE flat all-reduce after
    backward gives -- bit for bit with two ranks (a + b has one order); with four the transport may associate the four
    terms differently for a slice than for the whole buffer, so the bound is one rounding of the sum."""
    import torch
    script = tmp_path / "bucket_child.py"
    script.write_text(_BUCKET_CHILD)
    env = dict(os.environ, CARTNET_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29550 + world), str(script), ROOT,
                          str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    outs = [torch.load(tmp_path / f"b_w{world}_r{r}.pt") for r in range(world)]
    for o in outs:
        assert o["buckets"] == 3 + 2 and o["flat_scale"] == o["bucketed_scale"] == 1.0 / world
        assert torch.equal(o["bucketed"], outs[0]["bucketed"])                 # every rank holds the same sum
        assert o["flat"].abs().max().item() > 0
        if world == 2:
            assert torch.equal(o["bucketed"], o["flat"])
        else:
            assert (o["bucketed"] - o["flat"]).abs().max().item() <= 2.0 ** -22 * o["flat"].abs().max().item()
