#!/usr/bin/env python3
"""Entry point with the reference's flag surface and call order (reference: main.py:121-227), on the MI355X path.

What is kept: the argparse flags and their defaults (main.py:123-154), the copy into the global ``cfg``
(:156-188), seed -> loaders -> ``create_model()`` -> Adam -> ``train`` (:196-227), OneCycleLR, gradient accumulation
with the last-iteration flush, best-validation checkpoint ``{"model_state", "optimizer_state"}`` under
``results/<name>/<seed>/ckpt/best.ckpt`` (train/train.py:91-102) and its reload for the test pass (:114-115),
``--inference`` / ``--montecarlo`` on a checkpoint (main.py:21-119, 212-225).
What differs: the datasets (CSD / Jarvis need licences or the network) are replaced by synthetic ADP-shaped crystals
(``--synthetic N`` graphs, ``--atoms lo hi``), wandb / GraphGym logging are dropped, ``--device`` replaces the hard-coded
"cuda:0", and under ``torch.distributed.run`` the crystals are sharded across ranks with one gradient all-reduce per
optimiser step.  There is no CPU path: the model runs on an AMD GPU only.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from cartnet_amd import distributed as cdist
from cartnet_amd.config import cfg, set_cfg
from cartnet_amd.data import DataLoader
from cartnet_amd.master import create_model
from cartnet_amd.optim import FlatAdam, one_cycle_lr, one_cycle_momentum
from cartnet_amd.synthetic import augment_data, make_crystal
from cartnet_amd.train import eval_epoch, train_epoch


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    # --- the reference's flags (main.py:123-154), same names / defaults / store_false quirks
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--name", type=str, default="CartNet")
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--batch_accumulation", type=int, default=16)
    p.add_argument("--dataset", type=str, default="ADP")
    p.add_argument("--dataset_path", type=str, default="./dataset/ADP_DATASET/")
    p.add_argument("--inference", action="store_true")
    p.add_argument("--montecarlo", action="store_true")
    p.add_argument("--checkpoint_path", type=str, default=None)
    p.add_argument("--inference_output", type=str, default="./inference.pkl")
    p.add_argument("--figshare_target", type=str, default="formation_energy_peratom")
    p.add_argument("--wandb_project", type=str, default="ADP")
    p.add_argument("--wandb_entity", type=str, default="aiquaneuro")
    p.add_argument("--loss", type=str, default="MAE")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--warmup", type=float, default=0.01)
    p.add_argument("--model", type=str, default="CartNet")
    p.add_argument("--max_neighbours", type=int, default=25)
    p.add_argument("--radius", type=float, default=5.0)
    p.add_argument("--num_layers", type=int, default=4)
    p.add_argument("--dim_in", type=int, default=256)
    p.add_argument("--dim_rbf", type=int, default=64)
    p.add_argument("--augment", action="store_true")
    p.add_argument("--invariant", action="store_true")
    p.add_argument("--disable_temp", action="store_false")
    p.add_argument("--no_standarize_temp", action="store_false")
    p.add_argument("--disable_envelope", action="store_false")
    p.add_argument("--disable_H", action="store_false")
    p.add_argument("--disable_atom_types", action="store_false")
    p.add_argument("--threads", type=int, default=8)
    p.add_argument("--workers", type=int, default=5)
    # --- this build
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--synthetic", type=int, default=32, help="number of synthetic crystals (80/10/10 split)")
    p.add_argument("--atoms", type=int, nargs=2, default=(30, 70), help="atoms per synthetic crystal: lo hi")
    p.add_argument("--gemm_precision", type=int, default=0, choices=(0, 1, 2),
                   help="GEMM arithmetic: 0 exact fp32 products (fp32 MFMA), 1 bf16x3 split operands (fp32-level "
                        "accuracy on the bf16 MFMA), 2 plain bf16 operands")
    p.add_argument("--montecarlo_rounds", type=int, default=100, help="passes of --montecarlo (the reference hard-codes 100)")
    p.add_argument("--fused_accumulation", action="store_true",
                   help="run the batch x batch_accumulation micro-batches of an optimiser step as ONE pass with BatchNorm "
                        "statistics and loss per micro-batch (CartnetGroups): the reference recipe's numbers at the "
                        "large-batch rate")
    p.add_argument("--bf16_storage", action="store_true",
                   help="with --gemm_precision 2: keep the layers' edge-sized intermediate tensors in HBM as bf16 (fp32 "
                        "accumulate, fp32 BatchNorm statistics)")
    p.add_argument("--sync_batchnorm", action="store_true",
                   help="data-parallel runs: BatchNorm statistics over the crystals of ALL ranks (one small all-reduce per "
                        "BatchNorm and direction) instead of per rank; CartNet only, not with --fused_accumulation")
    p.add_argument("--resident_dataset", action="store_true",
                   help="keep the splits as packed shards in HBM and build every batch (and its augmentation) on the GPU")
    return p


def fill_cfg(args) -> None:
    """main.py:156-188."""
    set_cfg()
    cfg.seed, cfg.name = args.seed, args.name
    cfg.run_dir = "results/" + cfg.name + "/" + str(cfg.seed)
    cfg.batch, cfg.batch_accumulation = args.batch, args.batch_accumulation
    cfg.dataset.name = args.dataset
    cfg.loss, cfg.lr, cfg.warmup = args.loss, args.lr, args.warmup
    cfg.optim.max_epoch = args.epochs
    cfg.model = args.model
    cfg.max_neighbours = -1 if cfg.model == "CartNet" else args.max_neighbours
    cfg.radius, cfg.num_layers, cfg.dim_in, cfg.dim_rbf = args.radius, args.num_layers, args.dim_in, args.dim_rbf
    cfg.augment = False if cfg.model in ("icomformer", "ecomformer") else args.augment
    cfg.invariant = args.invariant
    cfg.use_temp = False if cfg.dataset.name != "ADP" else args.disable_temp
    cfg.standarize_temp = args.no_standarize_temp
    cfg.envelope, cfg.use_H, cfg.use_atom_types = args.disable_envelope, args.disable_H, args.disable_atom_types
    cfg.workers = args.workers
    cfg.device = args.device
    cfg.gemm_precision = args.gemm_precision
    cfg.bn_group_size = 0
    cfg.half_storage = bool(args.bf16_storage) and cfg.model == "CartNet" and args.gemm_precision == 2
    cfg.sync_batchnorm = bool(args.sync_batchnorm) and cfg.model == "CartNet" and not args.fused_accumulation
    if args.fused_accumulation and cfg.model == "CartNet" and cfg.batch_accumulation > 1:
        # the loader hands out whole optimiser steps; the model normalises (and train_epoch averages the loss) per
        # micro-batch of the reference's size
        cfg.bn_group_size = cfg.batch
        cfg.batch, cfg.batch_accumulation = cfg.batch * cfg.batch_accumulation, 1


def create_loaders(args, rank: int, world: int):
    """Synthetic stand-in for loader/loader.py:create_loader: seed-123 80/10/10 split (loader.py:130-141)."""
    adp = cfg.dataset.name == "ADP"
    graphs = [make_crystal(g, None, cfg.radius, tuple(args.atoms), adp=adp) for g in range(args.synthetic)]
    perm = torch.randperm(len(graphs), generator=torch.Generator().manual_seed(123)).tolist()
    n_tr, n_va = int(0.8 * len(graphs)), int(0.1 * len(graphs))
    tr = [graphs[i] for i in perm[:n_tr]]
    va = [graphs[i] for i in perm[n_tr:n_tr + n_va]] or tr[:1]
    te = [graphs[i] for i in perm[n_tr + n_va:]] or tr[:1]
    if args.resident_dataset:                                             # cartnet_amd/shard.py
        from cartnet_amd.shard import DeviceShard, ShardLoader
        shards = [DeviceShard.from_data_list(part, cfg.device) for part in (tr, va, te)]
        return [ShardLoader(shards[0], cfg.batch, shuffle=True, seed=cfg.seed, rank=rank, world_size=world,
                            augment=cfg.augment),
                ShardLoader(shards[1], cfg.batch), ShardLoader(shards[2], 1 if adp else cfg.batch)]
    gen = torch.Generator().manual_seed(cfg.seed + 1000 * rank)
    aug = (lambda d: augment_data(d, gen)) if cfg.augment else None
    return [DataLoader(tr, cfg.batch, shuffle=True, seed=cfg.seed, rank=rank, world_size=world, transform=aug),
            DataLoader(va, cfg.batch), DataLoader(te, 1 if adp else cfg.batch)]


def inference(model, loader, device, output_path: str) -> dict:
    """main.py:21-60: eval mode, per test batch (batch size 1 on ADP, loader/loader.py:121) the prediction, the target
    and the per-atom IoU / MAE / similarity index (train/metrics.py, here on the GPU); everything is pickled to
    ``output_path``.  The synthetic crystals carry no refcode / original temperature: those two lists stay empty."""
    import pickle
    from cartnet_amd.metrics import compute_3D_IoU, get_similarity_index
    model.eval()
    out = {"pred": [], "true": [], "temp": [], "cell": [], "refcode": [], "pos": [], "atoms": [], "iou": [], "mae": [],
           "similarity_index": []}
    with torch.no_grad():
        for batch in loader:
            if batch is None:
                continue
            batch.to(device)
            out["cell"].append(batch.cell.detach().to("cpu"))
            out["atoms"].append(batch.x[batch.non_H_mask].detach().to("cpu"))     # read BEFORE forward overwrites x (main.py:40)
            if hasattr(batch, "pos"):
                out["pos"].append(batch.pos[batch.non_H_mask].detach().to("cpu"))
            pred, true = model(batch)
            out["pred"].append(pred.detach().to("cpu"))
            out["true"].append(true.detach().to("cpu"))
            out["iou"].append(compute_3D_IoU(pred, true).detach().to("cpu"))
            out["mae"].append((pred - true).abs().detach().to("cpu"))
            out["similarity_index"].append(get_similarity_index(pred, true).detach().to("cpu"))
    if hasattr(model, "flush_graph_checks"):
        model.flush_graph_checks()
    iou, mae, sim = (torch.cat(out[k]) for k in ("iou", "mae", "similarity_index"))
    with open(output_path, "wb") as f:
        pickle.dump(out, f)
    return {"iou_mean": float(iou.mean()), "iou_std": float(iou.std()), "mae_mean": float(mae.mean()),
            "mae_std": float(mae.std()), "similarity_index_mean": float(sim.mean()),
            "similarity_index_std": float(sim.std()), "output": output_path}


def montecarlo(model, loader, device, output_path: str, rounds: int = 100, seed: int = 0) -> dict:
    """main.py:62-119: for ``rounds`` passes over the test loader, predict, rotate ``cart_dir`` by a uniform random
    rotation R, predict again and compare with the rotated first prediction R^T pred R (IoU, MAE, similarity index): how
    equivariant the trained network has become.  One pickle per round, as the reference writes them."""
    import pickle
    from cartnet_amd.metrics import compute_3D_IoU, get_similarity_index
    from cartnet_amd.shard import random_rotations
    model.eval()
    gen = torch.Generator(device=device).manual_seed(seed)
    iou_all, mae_all, sim_all = [], [], []
    with torch.no_grad():
        for i in range(rounds):
            out = {"pred": [], "true": [], "cell": [], "refcode": [], "pos": [], "atoms": [], "mae": [], "iou": [],
                   "similarity_index": []}
            for batch in loader:
                if batch is None:
                    continue
                batch_copy = batch.clone()                                   # forward overwrites batch.x (main.py:87)
                batch_copy.num_graphs = batch.num_graphs
                batch.to(device)
                out["cell"].append(batch.cell.detach().to("cpu"))
                out["atoms"].append(batch.x[batch.non_H_mask].detach().to("cpu"))
                pseudo_true, _ = model(batch)
                R = random_rotations(1, gen, device)[0]
                batch_copy.to(device)
                batch_copy.cart_dir = batch_copy.cart_dir @ R
                pseudo_true = R.transpose(-1, -2) @ pseudo_true @ R
                pred, _ = model(batch_copy)
                out["pred"].append(pred.detach().to("cpu"))
                out["true"].append(pseudo_true.detach().to("cpu"))
                out["iou"].append(compute_3D_IoU(pred, pseudo_true).detach().to("cpu"))
                out["similarity_index"].append(get_similarity_index(pred, pseudo_true).detach().to("cpu"))
                out["mae"].append((pred - pseudo_true).abs().detach().to("cpu"))
            with open(output_path.replace(".pkl", f"_montecarlo_{i}.pkl"), "wb") as f:
                pickle.dump(out, f)
            iou_all += out["iou"]
            mae_all += out["mae"]
            sim_all += out["similarity_index"]
    iou, mae, sim = torch.cat(iou_all), torch.cat(mae_all), torch.cat(sim_all)
    return {"rounds": rounds, "iou_mean": float(iou.mean()), "iou_std": float(iou.std()), "mae_mean": float(mae.mean()),
            "mae_std": float(mae.std()), "similarity_index_mean": float(sim.mean()),
            "similarity_index_std": float(sim.std())}


def main(argv=None) -> dict:
    args = build_parser().parse_args(argv)
    fill_cfg(args)
    torch.set_num_threads(args.threads)
    rank, world, local = cdist.init_from_env()
    if world > 1:
        if os.environ.get("CARTNET_SHARE_GPU"):      # rehearsal on a box with fewer GPUs than ranks (gloo backend)
            local %= max(1, torch.cuda.device_count())
        cfg.device = f"cuda:{local}"
    torch.manual_seed(cfg.seed)
    loaders = create_loaders(args, rank, world)
    model = create_model()
    n_params = sum(p.numel() for p in model.parameters())
    if args.inference or args.montecarlo:                                  # main.py:212-225 (ADP only, trained checkpoint)
        assert cfg.dataset.name == "ADP", "ADPs inference only for ADP dataset."
        assert args.checkpoint_path is not None, "Weights not provided."
        ck = torch.load(args.checkpoint_path, map_location=cfg.device)
        model.load_state_dict(ck["model_state"])
        if args.inference:
            res = inference(model, loaders[-1], cfg.device, args.inference_output)
        else:
            res = montecarlo(model, loaders[-1], cfg.device, args.inference_output, rounds=args.montecarlo_rounds,
                             seed=cfg.seed)
        if rank == 0:
            print(json.dumps(res), flush=True)
        return res
    opt = FlatAdam(model, lr=cfg.lr)
    steps_per_epoch = len(loaders[0])
    total_steps = cfg.optim.max_epoch * steps_per_epoch // cfg.batch_accumulation + cfg.optim.max_epoch   # train.py:59
    sched_step = [0]

    def scheduler():          # OneCycleLR.step() (train/train.py:188): the learning rate and, with Adam, beta1
        sched_step[0] += 1
        k = min(sched_step[0], total_steps - 1)
        opt.set_lr(one_cycle_lr(k, total_steps, cfg.lr, cfg.warmup))
        opt.set_beta1(one_cycle_momentum(k, total_steps, cfg.warmup))

    opt.set_lr(one_cycle_lr(0, total_steps, cfg.lr, cfg.warmup))
    opt.set_beta1(one_cycle_momentum(0, total_steps, cfg.warmup))
    ckpt_dir = os.path.join(cfg.run_dir, "ckpt")
    best, history = float("inf"), []
    for epoch in range(cfg.optim.max_epoch):
        t0 = time.perf_counter()
        tr = train_epoch(loaders[0], model, opt, cfg.batch_accumulation, scheduler, device=cfg.device)
        va = eval_epoch(loaders[1], model, device=cfg.device)
        history.append({"epoch": epoch, "train_mae": tr["mae"], "val_mae": va["mae"],
                        "graphs_per_s": tr["graphs"] * world / tr["seconds"], "time_epoch": time.perf_counter() - t0})
        if rank == 0:
            print(json.dumps(history[-1]), flush=True)
            if va["mae"] < best:                                          # train/train.py:91-102
                best = va["mae"]
                os.makedirs(ckpt_dir, exist_ok=True)
                torch.save({"model_state": model.state_dict(), "optimizer_state": opt.state_dict()},
                           os.path.join(ckpt_dir, "best.ckpt"))
    cdist.assert_replicas_in_sync(model)
    cdist.barrier()
    result = {"params": n_params, "history": history, "best_val_mae": best}
    if rank == 0 and os.path.exists(os.path.join(ckpt_dir, "best.ckpt")):  # train/train.py:114-117
        ck = torch.load(os.path.join(ckpt_dir, "best.ckpt"), map_location=cfg.device)
        model.load_state_dict(ck["model_state"])
        adp = cfg.dataset.name == "ADP"
        test = eval_epoch(loaders[2], model, device=cfg.device, adp_metrics=adp, test_metrics=adp)
        result["test_mae"] = test["mae"]
        result["test_metrics"] = test                                      # train/metrics.py:201-214
        print(json.dumps({"params": n_params, "best_val_mae": best, "test": test}), flush=True)
    return result


if __name__ == "__main__":
    main()
