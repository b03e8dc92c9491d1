"""Training / evaluation loops with the reference's semantics (reference: train/train.py:148-243,
train/metrics.py:15-28): MAE or MSE loss, unscaled gradient accumulation over ``batch_accumulation`` micro-batches
with a flush on the last iteration, optimiser + scheduler step at the boundary.  Logging / wandb are out of scope;
the loops return plain numbers (graphs/s is what this build reports)."""
from __future__ import annotations

import os
import time
from typing import Callable, Optional

import torch

from .config import cfg
from . import distributed as cdist


_FUSED_LOSS = os.environ.get("CARTNET_FUSED_LOSS", "1") != "0"      # A/B switch (README)


class _FusedLoss(torch.autograd.Function):
    """(MAE, MSE) of device tensors from two small launches, their gradient from one more (``cartnet_loss_fwd`` /
    ``cartnet_loss_bwd``, include/cartnet_hip.h) instead of eight eager kernels per step."""

    @staticmethod
    def forward(ctx, pred, true):
        from . import lib as _l
        p, t = pred.contiguous(), true.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=p.device)
        lib = _l.load()
        n = p.numel()
        ctx.unit = None
        if n <= 4096 and pred.requires_grad:
            # one launch; it also leaves the gradients of either loss for a seed of exactly 1 (what backward() below feeds)
            ctx.unit = torch.empty(2 * n, dtype=torch.float32, device=p.device)
            _l.check(lib.cartnet_loss_fwd_unit(p.data_ptr(), t.data_ptr(), n, out.data_ptr(), ctx.unit.data_ptr(),
                                               _l.stream_ptr()), "cartnet_loss_fwd_unit")
        else:
            parts = torch.empty(2 * int(lib.cartnet_loss_nparts(n)), dtype=torch.float64, device=p.device)
            _l.check(lib.cartnet_loss_fwd(p.data_ptr(), t.data_ptr(), n, parts.data_ptr(), out.data_ptr(),
                                          _l.stream_ptr()), "cartnet_loss_fwd")
        ctx.save_for_backward(p, t)
        ctx.set_materialize_grads(False)        # the loss that is not used arrives as None, not as a zero tensor
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_mae, g_mse):
        from . import lib as _l
        p, t = ctx.saved_tensors
        if g_mae is None and g_mse is None:
            return None, None
        unit = getattr(ctx, "unit", None)
        if unit is not None and (g_mae is None) != (g_mse is None):
            # the seed is THE cached scalar one of cartnet_amd.train.backward: its gradient was written by the forward launch
            g, k = (g_mae, 0) if g_mse is None else (g_mse, 1)
            if g is _ONES.get((g.device, g.dtype)):
                n = p.numel()
                return unit[k * n:(k + 1) * n].view(p.shape), None
        dpred = torch.empty_like(p)
        ga = g_mae.contiguous() if g_mae is not None else None
        gs = g_mse.contiguous() if g_mse is not None else None
        _l.check(_l.load().cartnet_loss_bwd(p.data_ptr(), t.data_ptr(), p.numel(), _l.ptr(ga), _l.ptr(gs),
                                            dpred.data_ptr(), _l.stream_ptr()), "cartnet_loss_bwd")
        return dpred, None


_ONES = {}


def backward(loss: torch.Tensor) -> None:
    """``loss.backward()`` for the scalar loss of a training step with the seed gradient taken from a cached device
    scalar: autograd's own ``ones_like`` is one more launch on the step's critical chain (6 us of a 1.2 ms step at
    configs[2] shapes).  A non-scalar loss is reduced with ``mean()`` first, as the reference does (train/train.py:183)."""
    if loss.dim() != 0:
        loss = loss.mean()
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    if not _two_node_backward(loss, one):
        loss.backward(one)


def _two_node_backward(loss: torch.Tensor, one: torch.Tensor) -> bool:
    """The training step's graph is two nodes, each ONE native call: compute_loss (cartnet_loss_bwd) on the prediction of
    CartNet (cartnet_model_backward) or iComformer (cartnet_icomformer_backward), the parameter gradients going to the
    optimiser's flat buffer.  When the
    graph is exactly that, run the two backward functions here instead of handing them to the autograd engine (0.15 ms
    of host time per step, a seventh of a configs[2] step).  Anything else -- another loss term, a hook, anomaly mode,
    parameters without a FlatAdam -- returns False and the engine runs as usual."""
    fn = loss.grad_fn
    if fn is None or type(fn).__name__ != "_FusedLossBackward" or torch.is_anomaly_enabled():
        return False
    nxt = fn.next_functions
    if len(nxt) != 2 or nxt[0][0] is None or nxt[0][1] != 0 or nxt[1][0] is not None:
        return False
    net = nxt[0][0]
    saved = getattr(net, "saved", None)
    n_out = {"_CartNetFunctionBackward": 3, "_IcfNativeFunctionBackward": 2}.get(type(net).__name__)   # outputs of the forward
    if n_out is None or not saved:
        return False
    model = saved[0]
    if getattr(model, "_flat_grad", None) is None or loss._backward_hooks:
        return False
    for node in (fn, net):
        if getattr(node, "_backward_hooks", None) or getattr(node, "_backward_pre_hooks", None):
            return False
    k = loss.output_nr                      # 0: MAE, 1: MSE (the two outputs of _FusedLoss)
    if k not in (0, 1):
        return False
    if getattr(fn, "_cartnet_consumed", False):
        # (the engine raises "Trying to backward through the graph a second time" here; the shortcut must not be quieter)
        raise RuntimeError("cartnet_amd.train.backward: this loss has been back-propagated already (its saved "
                           "activations were consumed in place)")
    fn._cartnet_consumed = True
    with torch.no_grad():                   # what the engine does around every backward function
        dpred = fn.apply(one if k == 0 else None, one if k == 1 else None)[0]
        out = net.apply(dpred, *([None] * (n_out - 1)))
    if any(g is not None for g in out):     # (cannot happen with a FlatAdam attached: the gradients went to its buffer)
        raise RuntimeError("cartnet_amd.train.backward: the network returned gradients the shortcut does not deliver")
    return True


def compute_loss(pred: torch.Tensor, true: torch.Tensor):
    """(MAE, MSE) with mean reduction over all elements (train/metrics.py:26-27).  Device tensors go through the fused
    kernels; host tensors (the CPU tests of the loops' bookkeeping) through torch."""
    if _FUSED_LOSS and pred.is_cuda and pred.dtype == torch.float32 and true.dtype == torch.float32 and pred.shape == true.shape \
            and pred.numel() > 0:
        return _FusedLoss.apply(pred, true)
    diff = pred - true
    return diff.abs().mean(), (diff * diff).mean()


def grouped_loss(pred: torch.Tensor, true: torch.Tensor, batch, group_size: int):
    """(sum over groups of the group's MAE, same for MSE, number of groups) for a batch whose consecutive crystals
    form micro-batches of ``group_size`` (model.bn_group_size): the reference computes the mean loss of every
    micro-batch and accumulates their gradients unscaled (train/train.py:173-189), i.e. it descends the SUM of the
    per-micro-batch means.  Rows of ``pred`` are the non-H atoms (Cholesky head) or the crystals (scalar head).
    Device-side with static shapes only: no host synchronisation."""
    Bg = int(batch.num_graphs)
    G = (Bg + group_size - 1) // group_size
    dev = pred.device
    M = int(pred.shape[0])
    if pred.dim() == 3:                                   # [M, 3, 3]: row m belongs to the m-th non-H atom
        per_row = float(pred.shape[1] * pred.shape[2])
        mask = batch.non_H_mask
        gid_atom = torch.div(batch.batch, group_size, rounding_mode="floor")
        # rank of every non-H atom among the non-H atoms = its row; all shapes are static, nothing syncs
        dest = torch.where(mask, torch.cumsum(mask, 0) - 1, torch.full_like(gid_atom, M))
        gid = torch.zeros(M + 1, dtype=torch.int64, device=dev).scatter_(0, dest, gid_atom)[:M]
    else:                                                 # [Bg]: one row per crystal
        per_row = 1.0
        gid = torch.div(torch.arange(Bg, device=dev), group_size, rounding_mode="floor")
    rows = torch.zeros(G, dtype=pred.dtype, device=dev).index_add_(0, gid, torch.ones(M, dtype=pred.dtype, device=dev))
    w = (1.0 / (per_row * rows.clamp(min=1.0)))[gid]       # 1 / (elements of the row's group)
    diff = (pred - true).reshape(M, -1)
    return (diff.abs().sum(dim=1) * w).sum(), ((diff * diff).sum(dim=1) * w).sum(), G


def _pick_loss(mae, mse):
    if cfg.loss == "MAE":
        return mae
    if cfg.loss == "MSE":
        return mse
    raise Exception("Loss not implemented")


def train_epoch(loader, model, optimizer, batch_accumulation: int, scheduler: Optional[Callable[[], None]] = None,
                device="cuda:0"):
    """One pass over ``loader`` (train/train.py:148-199).  Returns dict(loss, mae, graphs, seconds)."""
    model.train()
    had_direct = getattr(optimizer, "direct_grads", None)
    if had_direct is not None:
        optimizer.direct_grads = True               # the loss below depends on the parameters through the model only
    try:
        return _train_epoch(loader, model, optimizer, batch_accumulation, scheduler, device)
    finally:
        if had_direct is not None:                  # a caller's own steps (extra loss terms, hand-edited p.grad) accumulate again
            optimizer.direct_grads = had_direct
            optimizer.fresh = False


def _train_epoch(loader, model, optimizer, batch_accumulation, scheduler, device):
    optimizer.zero_grad()
    n_iter = len(loader)
    tot_mae = torch.zeros((), device=device)
    graphs = 0
    micro = 0                                       # micro-batches seen (= iterations unless batches carry groups)
    t0 = time.perf_counter()
    flush = getattr(model, "flush_graph_checks", None)
    # Multi-rank runs of CartNet: the gradient all-reduce goes out in buckets DURING the backward of the micro-batch that
    # closes an accumulation window (distributed.GradSync, CartnetGradReadyFn) instead of as one flat call behind it.
    # Only when the model's backward really writes into THIS optimiser's flat buffer (FlatAdam over every parameter): a
    # FlatAdam over a model with a frozen parameter leaves model._flat_grad unset, backward then reports no bucket, and the
    # boundary below falls back to the flat all-reduce.
    sync = None
    n_buckets = 0
    if hasattr(optimizer, "flat_grad") and hasattr(model, "grad_bucket_order") and cdist._active() and \
            getattr(model, "_flat_grad", None) is optimizer.flat_grad:
        sync = cdist.GradSync(optimizer.flat_grad)
        n_buckets = len(model.grad_bucket_order())
    for it, batch in enumerate(loader):
        if batch is None and getattr(model, "sync_batchnorm", False):
            raise RuntimeError("sync_batchnorm: this rank has no crystals for a step the other ranks run -- every rank "
                               "must take part in each BatchNorm exchange (use more crystals per step than ranks)")
        if batch is not None:                       # None: this rank has no crystals left for the step (sharded
            batch.to(device)                        # loaders, tiny data sets) -- it adds a zero gradient
            pred, true = model(batch)
            gsz = int(getattr(model, "bn_group_size", 0) or 0)
            if gsz > 0 and int(batch.num_graphs) > gsz:
                # the batch carries several micro-batches of the reference recipe (CartnetGroups): sum of their losses
                mae, mse, n_groups = grouped_loss(pred, true, batch, gsz)
                micro += n_groups
            else:
                mae, mse = compute_loss(pred, true)
                micro += 1
            loss = _pick_loss(mae, mse)
            boundary = ((it + 1) % batch_accumulation == 0) or (it + 1 == n_iter)
            if sync is not None and boundary:
                model.grad_sync = sync              # this backward reports its buckets; their all-reduces overlap it
            try:
                backward(loss)                      # not divided by the accumulation count (train/train.py:183)
            finally:
                if sync is not None:
                    model.grad_sync = None
            tot_mae += mae.detach()
            graphs += int(batch.num_graphs)
        if ((it + 1) % batch_accumulation == 0) or (it + 1 == n_iter):
            if flush is not None:
                flush()                             # a malformed batch raises BEFORE its gradient reaches the weights
            if sync is not None:
                if batch is None:                   # no crystals this step: the same collectives, in the same order
                    for lo, hi in model.grad_bucket_order():
                        sync.bucket(lo, hi)
                seen, sync.buckets_seen = sync.buckets_seen, 0
                # The collective sequence of a step is a STATIC property of the run: with `sync` every rank issues exactly the
                # n_buckets bucket all-reduces (a rank without crystals issued them above).  A rank that chose a different
                # sequence on its own -- one flat all-reduce because its backward reported nothing -- would pair collectives
                # of different sizes with its peers' (ADVICE r5): any other count is an error on this rank, at once.
                if seen != n_buckets:
                    raise RuntimeError(f"gradient buckets: backward reported {seen} of {n_buckets} -- the all-reduces of "
                                       "this step would not match the other ranks' (a backward that bypasses "
                                       "cartnet_model_backward's bucket callbacks cannot be combined with GradSync)")
                scale = sync.finish()
            else:
                scale = cdist.all_reduce_gradients(optimizer.flat_grad) if hasattr(optimizer, "flat_grad") else 1.0
            optimizer.step(scale) if hasattr(optimizer, "flat_grad") else optimizer.step()
            if scheduler is not None:
                scheduler()
            optimizer.zero_grad()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"mae": float(tot_mae.item()) / max(micro, 1), "graphs": graphs, "seconds": dt}


def eval_epoch(loader, model, device="cuda:0", adp_metrics=False, test_metrics=False):
    """train/train.py:202-243: eval mode, no grad.  ``adp_metrics`` adds the per-batch means the reference logs for
    the ADP dataset (train/metrics.py:201-225): volume error and similarity index always, the voxel IoU only for the
    test pass (``test_metrics``); all on the GPU (cartnet_amd/metrics.py)."""
    from .metrics import adp_metrics as _adp_metrics
    model.eval()
    names = ["mae"] + (["volume_percentage_error", "similarity_index"] + (["iou"] if test_metrics else [])
                       if adp_metrics else [])
    tot = {k: torch.zeros((), device=device) for k in names}
    n = 0
    with torch.no_grad():
        for batch in loader:
            if batch is None:
                continue
            batch.to(device)
            pred, true = model(batch)
            mae, _ = compute_loss(pred, true)
            tot["mae"] += mae
            if adp_metrics:
                vol, sim, iou = _adp_metrics(pred, true, True, True, test_metrics)
                tot["volume_percentage_error"] += vol.mean()
                tot["similarity_index"] += sim.mean()
                if test_metrics:
                    tot["iou"] += iou.mean()
            n += 1
    if hasattr(model, "flush_graph_checks"):
        model.flush_graph_checks()
    return {k: float(v.item()) / max(n, 1) for k, v in tot.items()}
