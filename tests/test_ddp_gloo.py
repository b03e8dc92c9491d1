"""Batched-graph sharding over 2 ranks (gloo, CPU): the sharding helpers partition crystals disjointly, and the flat
gradient all-reduce + mean reproduces the single-process gradient of the union of the per-rank losses.

The model used here is the ORACLE (CPU restatement) wrapped in a tiny nn.Module -- the distributed plumbing under
test (cartnet_amd.distributed, shard_range, flat-buffer all-reduce) is device-agnostic host logic."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _loss_and_flat_grad(sd_params, names, batch):
    from oracle import cartnet_ref as orc
    for n in names:
        sd_params[n].grad = None
    pred = orc.cartnet_forward(sd_params, batch, num_layers=2, training=True)
    loss = (pred - batch.y).abs().mean()
    loss.backward()
    return loss.detach(), torch.cat([sd_params[n].grad.reshape(-1) for n in names])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from cartnet_amd import distributed as cdist
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    r, w, _ = cdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sd = make_state_dict(16, 8, 2, seed=5)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "rbf" not in k]
    params = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    crystals = [make_crystal(40 + i, 6 + i) for i in range(5)]
    mine = cdist.shard_range(len(crystals), rank, world)
    batch = Batch.from_data_list([crystals[i] for i in mine])
    loss, flat = _loss_and_flat_grad(params, names, batch)
    scale = cdist.all_reduce_gradients(flat)          # SUM over ranks, returns 1/world
    assert scale == 1.0 / world
    cdist.barrier()
    tmax = cdist.max_over_ranks(float(rank), torch.device("cpu"))
    assert tmax == world - 1
    # replica check: identical parameters pass, a perturbed rank is reported on every rank
    lin = torch.nn.Linear(3, 3)
    with torch.no_grad():
        for p_ in lin.parameters():
            p_.fill_(0.5)
    cdist.assert_replicas_in_sync(lin)
    with torch.no_grad():
        lin.weight[0, 0] += float(rank)
    diverged = False
    try:
        cdist.assert_replicas_in_sync(lin)
    except RuntimeError as exc:
        diverged = "diverged" in str(exc)
    torch.save({"flat": flat * scale, "loss": loss, "idx": list(mine), "diverged": diverged},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert sorted(res[0]["idx"] + res[1]["idx"]) == [0, 1, 2, 3, 4]           # disjoint, exhaustive
    assert torch.equal(res[0]["flat"], res[1]["flat"])                         # every rank ends with the same gradient
    assert res[0]["diverged"] and res[1]["diverged"]                            # replica check fires on both ranks

    # single-process reference: mean over ranks of the per-rank losses (each rank normalises BatchNorm and the MAE
    # over its own crystals -- standard data-parallel semantics)
    sys.path.insert(0, ROOT)
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    sd = make_state_dict(16, 8, 2, seed=5)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "rbf" not in k]
    params = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    crystals = [make_crystal(40 + i, 6 + i) for i in range(5)]
    total = None
    for r in range(world):
        b = Batch.from_data_list([crystals[i] for i in res[r]["idx"]])
        _, flat = _loss_and_flat_grad(params, names, b)
        total = flat if total is None else total + flat
    assert torch.allclose(res[0]["flat"], total / world, rtol=1e-5, atol=1e-7)


def test_shard_range_partitions():
    from cartnet_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 162270):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ---------------------------------------------------------------------------------------------------------------
# World of EIGHT (what the driver's scaling run uses) rehearsed on the CPU over gloo: the loader's edge-balanced uneven
# shards, equal optimiser-step counts with EMPTY steps on the ranks that run out of crystals, the accumulation boundary
# and one flat-gradient all-reduce per optimiser step.  The per-crystal work is the oracle in eval-mode BatchNorm with a
# SUM loss, so that the summed gradient does not depend on how the crystals are cut into ranks and batches.
def _sum_loss_flat_grad(params, names, batch):
    from oracle import cartnet_ref as orc
    for n in names:
        params[n].grad = None
    pred = orc.cartnet_forward(params, batch, num_layers=2, training=False)
    (pred - batch.y).abs().sum().backward()
    return torch.cat([(params[n].grad if params[n].grad is not None else torch.zeros_like(params[n])).reshape(-1)
                      for n in names])


def _world8_worker(rank, world, port, out_dir, n_items, batch_size, accumulation):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from cartnet_amd import distributed as cdist
    from cartnet_amd.data import DataLoader
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    cdist.init_from_env(backend="gloo")
    sd = make_state_dict(16, 8, 2, seed=5)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "rbf" not in k]
    params = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    sizes = torch.randint(3, 25, (n_items,), generator=torch.Generator().manual_seed(3)).tolist()
    items = [make_crystal(300 + i, n) for i, n in enumerate(sizes)]
    loader = DataLoader(items, batch_size, shuffle=True, seed=11, rank=rank, world_size=world)
    chunks = loader._batches()
    n_iter = len(loader)
    flat = torch.zeros(sum(params[n].numel() for n in names))
    reduced, empty = [], 0
    for it, b in enumerate(loader):                       # the structure of cartnet_amd.train.train_epoch
        if b is None:
            empty += 1                                    # a rank without crystals for this step adds a zero gradient
        else:
            flat += _sum_loss_flat_grad(params, names, b)
        if (it + 1) % accumulation == 0 or it + 1 == n_iter:
            scale = cdist.all_reduce_gradients(flat)
            assert scale == 1.0 / world
            reduced.append(flat.clone())
            flat.zero_()
    lin = torch.nn.Linear(2, 2)
    with torch.no_grad():
        for p_ in lin.parameters():
            p_.fill_(0.25)
    cdist.assert_replicas_in_sync(lin)
    torch.save({"reduced": reduced, "chunks": chunks, "n_iter": n_iter, "empty": empty}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,batch_size,accumulation,expect_empty", [(37, 4, 1, False), (11, 1, 2, True)])
def test_world_of_eight_uneven_shards_and_empty_steps(tmp_path, n_items, batch_size, accumulation, expect_empty):
    world = 8
    port = _free_port()
    mp.spawn(_world8_worker, args=(world, port, str(tmp_path), n_items, batch_size, accumulation), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert len({r["n_iter"] for r in res}) == 1                               # the same number of steps on every rank
    n_iter = res[0]["n_iter"]
    assert n_iter == -(-n_items // (world * batch_size))
    seen = sorted(j for r in res for c in r["chunks"] for j in c)
    assert seen == list(range(n_items))                                       # nothing dropped, nothing doubled
    assert (sum(r["empty"] for r in res) > 0) == expect_empty
    sizes = [sum(len(c) for c in r["chunks"]) for r in res]
    assert max(sizes) > min(sizes) or n_items % world == 0                    # uneven shards are really exercised
    n_opt = len(res[0]["reduced"])
    assert n_opt == -(-n_iter // accumulation) and all(len(r["reduced"]) == n_opt for r in res)
    for r in res[1:]:
        for a, b in zip(r["reduced"], res[0]["reduced"]):
            assert torch.equal(a, b)                                          # every rank holds the same summed gradient
    # one process on the union: the sum over all crystals of the same per-crystal gradients
    sys.path.insert(0, ROOT)
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    sd = make_state_dict(16, 8, 2, seed=5)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "rbf" not in k]
    params = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    sizes = torch.randint(3, 25, (n_items,), generator=torch.Generator().manual_seed(3)).tolist()
    items = [make_crystal(300 + i, n) for i, n in enumerate(sizes)]
    union = _sum_loss_flat_grad(params, names, Batch.from_data_list(items))
    total = sum(res[0]["reduced"])
    assert torch.allclose(total, union, rtol=2e-4, atol=1e-5 * union.abs().max().item())
