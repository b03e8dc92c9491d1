"""ORACLE -- test infrastructure only, never the product path.

CPU restatement of the reference's training loop (train/train.py:148-199) around the oracle forward of
``oracle/cartnet_ref.py``: unscaled gradient accumulation, the boundary rule with its last-iteration flush, Adam as
main.py:208 constructs it (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) and
OneCycleLR as train/train.py:59 constructs it (cosine annealing, two phases, div_factor 25, final_div_factor 1e4, and --
its default -- Adam's beta1 cycled between 0.95 and 0.85 inversely to the learning rate).
torch.optim is third-party to the reference (pytorch==1.13.1 / 2.4.0, environment.yml); its published update rules are
restated here in a few lines of numpy-style arithmetic so that the pinned numbers do not depend on the installed torch.

Pinning: tests/golden/train_epoch.npz holds the output of the reference's OWN ``train_epoch`` (imported read-only with a
stand-in for the module-level ``wandb`` import, tests/golden/_ref_import.py::import_reference_train) run for two epochs
on the tiny model; tests/test_train_epoch_golden.py replays it with the functions below.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence, Tuple

import torch

from . import cartnet_ref as orc

Tensor = torch.Tensor


def one_cycle_lr(step: int, total_steps: int, max_lr: float, pct_start: float, div_factor: float = 25.0,
                 final_div_factor: float = 1e4) -> float:
    """Learning rate after ``step`` calls of ``scheduler.step()`` (train/train.py:59,188): OneCycleLR, anneal_strategy
    'cos', three_phase False: initial = max/div, min = initial/final_div; phase ends at pct_start*total - 1 and total - 1."""
    initial = max_lr / div_factor
    floor = initial / final_div_factor
    end1, end2 = float(pct_start * total_steps) - 1.0, float(total_steps) - 1.0

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if step <= end1:
        return cos(initial, max_lr, step / end1)
    return cos(max_lr, floor, (step - end1) / (end2 - end1))


def one_cycle_momentum(step: int, total_steps: int, pct_start: float, base_momentum: float = 0.85,
                       max_momentum: float = 0.95) -> float:
    """Adam's beta1 after ``step`` calls of ``scheduler.step()``: OneCycleLR's ``cycle_momentum=True`` default, which
    train/train.py:59 does not switch off -- beta1 runs max -> base over the first phase and base -> max over the second
    (written into ``optimizer.param_groups[0]["betas"]`` together with the learning rate)."""
    end1, end2 = float(pct_start * total_steps) - 1.0, float(total_steps) - 1.0

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if step <= end1:
        return cos(max_momentum, base_momentum, step / end1)
    return cos(base_momentum, max_momentum, (step - end1) / (end2 - end1))


def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
              betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam's single-tensor update (main.py:208), in place; ``step`` counts from 1:
    m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)."""
    b1, b2 = betas
    exp_avg.mul_(b1).add_(grad, alpha=1.0 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
    denom = exp_avg_sq.sqrt() / math.sqrt(1.0 - b2 ** step) + eps
    param.addcdiv_(exp_avg, denom, value=-lr / (1.0 - b1 ** step))


def is_boundary(it: int, n_iter: int, accum: int) -> bool:
    """train/train.py:186: step after every ``accum``-th micro-batch and after the last one of the epoch."""
    return ((it + 1) % accum == 0) or (it + 1 == n_iter)


def train_epoch(sd: Dict[str, Tensor], param_names: Sequence[str], micro_batches: Sequence, accum: int,
                on_boundary: Callable[[Dict[str, Tensor]], None], loss: str = "MAE", **forward_kw) -> List[Tuple[float, float]]:
    """One pass of train/train.py:148-199 over ``micro_batches`` with the oracle forward (training mode).  ``sd`` maps
    names to tensors; those in ``param_names`` must require grad.  Gradients accumulate UNSCALED in ``.grad``
    (train/train.py:183: ``loss.mean().backward()``); at every boundary ``on_boundary(grads)`` is called (the optimiser +
    scheduler step) and the gradients are zeroed.  BatchNorm running statistics are updated in ``sd`` after every forward.
    Returns (MAE, MSE) per iteration."""
    out = []
    n_iter = len(micro_batches)
    for p in param_names:
        sd[p].grad = None
    for it, batch in enumerate(micro_batches):
        new_stats: Dict[str, Tensor] = {}
        pred = orc.cartnet_forward(sd, batch, training=True, new_stats=new_stats, **forward_kw)
        mae, mse = orc.compute_loss(pred, batch.y.to(pred.dtype))
        if loss == "MAE":
            chosen = mae
        elif loss == "MSE":
            chosen = mse
        else:
            raise Exception("Loss not implemented")             # train/train.py:180
        chosen.mean().backward()
        for k, v in new_stats.items():
            sd[k] = v
        out.append((float(mae.detach()), float(mse.detach())))
        if is_boundary(it, n_iter, accum):
            on_boundary({p: sd[p].grad for p in param_names})
            for p in param_names:
                sd[p].grad = None
    return out
