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
    line with the whole-job rate, n_gpus = 2, weak scaling, the roofline object, and no cpu_baseline (N = 1 only)."""
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
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and "cpu_baseline" not in d
    assert d["config"]["parallelism"] == "graph-sharded dp2"


def test_divergence_is_detected():
    """The checksum comparison itself, single process: identical replicas pass (world size 1 is a no-op)."""
    import torch
    from cartnet_amd import distributed as cdist
    cdist.assert_replicas_in_sync(torch.nn.Linear(4, 4))
