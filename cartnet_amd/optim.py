"""Flat-buffer Adam for CartNet: all parameters (2.5 M fp32 at D=256, L=4) live in ONE contiguous device buffer, all
gradients in another, so the optimiser is a single fused kernel launch (cartnet_adam_step) and the data-parallel
gradient exchange is a single RCCL all-reduce of 10 MB.

Update rule = torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) as the reference constructs it
(reference: main.py:208); ``set_lr`` / ``set_beta1`` let a scheduler (OneCycleLR in the reference, train/train.py:59,
which cycles the learning rate AND Adam's beta1) drive it.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import ops


class FlatAdam:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs the model on the GPU (there is no CPU path)")
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)      # parameters become views of the flat buffer
                p.grad = self.flat_grad[off:off + k].view(p.shape)       # autograd accumulates in place into the views
                off += k
        self.params = params
        # ``direct_grads`` (off by default; cartnet_amd.train.train_epoch and bench.py switch it on): after zero_grad() the
        # buffer is ``fresh`` -- all zeros, nothing accumulated since -- and a model whose whole backward is one native call
        # (CartNet) writes its gradients straight into flat_grad instead of adding a staging buffer to it.  Only valid when
        # nothing else contributes to the parameters' gradients between zero_grad() and that backward: no loss term that
        # reads the parameters outside the model (an L2 penalty written into the loss), no hand edits of ``p.grad``.
        self.direct_grads = False
        self.fresh = False
        if hasattr(model, "_flat_grad") and len(params) == len(list(model.parameters())):
            model._flat_grad = self.flat_grad     # CartNet.backward adds its gradients here in one pass
            model.__dict__["_flat_owner"] = weakref.ref(self)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.param_groups = [{"lr": self.lr}]     # enough of torch's surface for a scheduler / logger to read lr

    def set_lr(self, lr: float) -> None:
        self.lr = float(lr)
        self.param_groups[0]["lr"] = self.lr

    def set_beta1(self, beta1: float) -> None:
        """Adam's first-moment decay for the next step: the reference's OneCycleLR cycles it together with the learning
        rate (``one_cycle_momentum``; torch writes it into ``param_groups[0]["betas"]``, which Adam reads every step --
        the bias correction 1 - beta1^t uses the current value too)."""
        self.betas = (float(beta1), self.betas[1])

    def zero_grad(self) -> None:
        self.flat_grad.zero_()
        self.fresh = bool(self.direct_grads)
        for p in self.params:                      # re-attach views if a caller reset them to None
            if p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr():
                self._reattach()
                break

    def _reattach(self) -> None:
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    def step(self, grad_scale: float = 1.0) -> None:
        """One Adam update.  ``grad_scale`` multiplies the gradient first (1/world_size after an all-reduce SUM)."""
        self.step_count += 1
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count, grad_scale)

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout -- ``{"state": {i: {"step", "exp_avg", "exp_avg_sq"}},
        "param_groups": [{"lr", "betas", "eps", ..., "params": [0..n-1]}]}`` with parameters numbered in
        ``model.parameters()`` order -- so that ``best.ckpt["optimizer_state"]`` (train/train.py:92-95) is
        interchangeable with the reference's: a torch Adam over the same model loads it, and
        ``load_state_dict`` below reads a reference checkpoint.  Moments are copies of the flat buffers' slices."""
        state, off = {}, 0
        for i, p in enumerate(self.params):
            k = p.numel()
            if self.step_count > 0:           # torch creates a parameter's state at its first step
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).clone()}
            off += k
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd) -> None:
        """Accepts torch.optim.Adam's layout (above; what the reference's checkpoints hold) and the flat layout
        ``{"step", "lr", "exp_avg", "exp_avg_sq"}`` written by round-1 builds of this package."""
        if "param_groups" not in sd:
            self.step_count = int(sd["step"])
            self.set_lr(sd["lr"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError(f"optimizer state holds {sum(len(g['params']) for g in groups)} parameters in "
                             f"{len(groups)} group(s); this model has {len(self.params)} in one")
        g = groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam implements plain Adam (weight_decay=0, amsgrad=False), as the reference uses it")
        self.set_lr(float(g["lr"]))
        self.betas = (float(g["betas"][0]), float(g["betas"][1]))
        self.eps = float(g["eps"])
        state = sd["state"]
        steps = {int(float(st["step"])) for st in state.values()}
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state FlatAdam can hold")
        self.step_count = steps.pop() if steps else 0
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        off = 0
        for i, (pid, p) in enumerate(zip(g["params"], self.params)):
            k = p.numel()
            st = state.get(pid, state.get(str(pid)))
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of parameter {i}: shape {tuple(st['exp_avg'].shape)}, "
                                     f"expected {tuple(p.shape)}")
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            off += k


def one_cycle_lr(step: int, total_steps: int, max_lr: float, pct_start: float = 0.3, div_factor: float = 25.0,
                 final_div_factor: float = 1e4) -> float:
    """torch.optim.lr_scheduler.OneCycleLR (cosine annealing, two phases) as the reference configures it
    (train/train.py:59: max_lr=cfg.lr, pct_start=cfg.warmup): learning rate AFTER ``step`` scheduler steps."""
    import math
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps) - 1.0

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if step <= end1:
        return cos(initial_lr, max_lr, step / end1 if end1 > 0 else 1.0)
    return cos(max_lr, min_lr, (step - end1) / (end2 - end1) if end2 > end1 else 1.0)


def one_cycle_momentum(step: int, total_steps: int, pct_start: float = 0.3, base_momentum: float = 0.85,
                       max_momentum: float = 0.95) -> float:
    """The OTHER half of torch.optim.lr_scheduler.OneCycleLR as the reference constructs it (train/train.py:59 leaves
    ``cycle_momentum=True``, ``base_momentum=0.85``, ``max_momentum=0.95``): with Adam the scheduler rewrites beta1 at
    every step, inversely to the learning rate -- 0.95 at the start, 0.85 at the peak, back to 0.95 -- on the same two
    cosine phases.  Found by pinning the loop to the reference's own train_epoch (tests/golden/train_epoch.npz): with a
    fixed beta1 = 0.9 the parameters after the second optimiser step were off by 0.45 lr."""
    import math
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps) - 1.0

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if step <= end1:
        return cos(max_momentum, base_momentum, step / end1 if end1 > 0 else 1.0)
    return cos(base_momentum, max_momentum, (step - end1) / (end2 - end1) if end2 > end1 else 1.0)
