"""BASELINE configs[3] in miniature on one card ("configs[3] is not exercised by any -m gpu test"): ragged
ADP-shaped crystals of 64..324 atoms at the real model size (L = 4, D = 256), edges from the GPU radius-graph builder, a
packed shard resident in HBM, every batch collated and SO(3)-augmented on the device, the epoch cut into EIGHT ranks'
edge-balanced shards by the same `rank_batches` the 8-GPU job uses -- and the eight ranks' work of each optimiser step
executed one after the other on the one card into one flat gradient buffer, which is what the RCCL SUM all-reduce
produces.  The first step's summed gradient is checked against the fp64 oracle run on the very batches the device loader
produced (augmentation included); the epoch must visit every crystal once with equal step counts and move the loss."""
import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu

WORLD = 8


def test_eight_rank_epoch_emulated_on_one_card_against_the_oracle():
    from cartnet_amd import shard
    from cartnet_amd.config import cfg
    from cartnet_amd.model import CartNet, make_state_dict
    from cartnet_amd.optim import FlatAdam
    from cartnet_amd.synthetic import make_geometry
    from oracle import cartnet_ref as orc
    cfg.radius = 5.0
    n, batch, D, R, L = 40, 2, 256, 64, 4
    geo = [make_geometry(41000 + g, None) for g in range(n)]                       # 64..324 atoms each
    arrays = shard.pack_with_gpu_graph(geo, 5.0, "cuda:0", chunk=16)
    sizes = np.diff(arrays["atom_ptr"])
    assert sizes.min() >= 64 and sizes.max() <= 324
    ds = shard.DeviceShard(arrays)
    sd = make_state_dict(D, R, L, seed=17)
    model = CartNet(D, R, L)
    model.load_state_dict(sd)
    model = model.cuda().train()
    opt = FlatAdam(model, lr=2e-4)
    names = [k for k, _ in model.named_parameters()]

    loaders = [shard.ShardLoader(ds, batch, shuffle=True, seed=5, rank=r, world_size=WORLD, augment=True) for r in range(WORLD)]
    plans = [ld._batches() for ld in loaders]
    steps = {len(p) for p in plans}
    assert len(steps) == 1                                                          # equal optimiser-step counts
    n_steps = steps.pop()
    assert n_steps == -(-n // (WORLD * batch))
    assert sorted(j for p in plans for c in p for j in c) == list(range(n))         # every crystal once, nothing dropped
    edges = [sum(int(arrays["edge_ptr"][j + 1] - arrays["edge_ptr"][j]) for c in p for j in c) for p in plans]
    assert max(edges) <= 1.35 * (sum(edges) / WORLD)                                 # edge-balanced shards (40 crystals only)

    iters = [iter(ld) for ld in loaders]
    losses = []
    for s in range(n_steps):
        opt.zero_grad()
        ref_grad = None
        step_loss = 0.0
        for r in range(WORLD):
            b = next(iters[r])
            if b is None:
                continue
            if s == 0:                                                              # the oracle on the device-made batch
                cpu = gu.clone_batch(b)
                for k, v in list(cpu.__dict__.items()):
                    if torch.is_tensor(v):
                        setattr(cpu, k, v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu())
                sd64 = {k: (v.double().requires_grad_(k in names) if v.is_floating_point() else v) for k, v in sd.items()}
                pr = orc.cartnet_forward(sd64, cpu, num_layers=L, training=True)
                (pr - cpu.y).abs().mean().backward()
                g = torch.cat([sd64[k].grad.reshape(-1) for k in names])
                ref_grad = g if ref_grad is None else ref_grad + g
            pred, true = model(b)
            loss = (pred - true).abs().mean()
            loss.backward()                                                          # accumulates into the flat buffer
            step_loss += float(loss.detach())
        if s == 0:
            got = opt.flat_grad.detach().double().cpu()
            gmax = ref_grad.abs().max().item()
            assert (got - ref_grad).abs().max().item() <= 3e-5 * gmax, (got - ref_grad).abs().max().item() / gmax
        opt.step(1.0 / WORLD)                                                        # the mean over ranks, as after the all-reduce
        losses.append(step_loss / WORLD)
    assert all(np.isfinite(losses)) and opt.step_count == n_steps
    model.flush_graph_checks()
