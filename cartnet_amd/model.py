"""CartNet on MI355X: the reference's model surface over hand-written gfx950 kernels.

Drop-in for ``models/cartnet.py`` of the reference: same constructor signature (cartnet.py:37-46), same
``forward(batch) -> (pred, true)`` contract including the in-place update of ``batch.x`` / ``batch.edge_attr``
(cartnet.py:65-73,154-159,223-225), same ``state_dict`` keys / shapes / registration order, so
reference checkpoints load and optimiser state lines up.

What differs is everything underneath.  The nn sub-modules below only *hold* parameters; no torch op computes any
part of the forward or backward pass.  ``forward`` runs one autograd Function for the whole network, which enqueues
kernels from libcartnet_hip.so (include/cartnet_hip.h) on torch's current HIP stream:

  * the first Linear of the gate / sender MLPs, Linear(3D->D) on cat[x_i, x_j, e] (cartnet.py:237,256), is split
    algebraically: x @ W[:, :D]^T and x @ W[:, D:2D]^T are computed once per *node*, e @ W[:, 2D:]^T per edge, and
    the node terms are gathered in the edge GEMM's epilogue -- no [E,3D] concatenation, no [E,D] gathers, 46 % fewer
    FLOPs than the reference formulation;
  * edges stay sorted by target (CSR), the scatter-sum is a per-target segmented reduction in edge order (the CPU
    scatter_add_ order), the by-source gradient uses a stable CSC permutation -- no atomics anywhere;
  * training-mode BatchNorm statistics come from per-tile partial sums written by the producing kernel and are
    finalised in fp64.

There is no CPU path: tensors must live on a HIP device and the shared library must be built.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

import ctypes as C

from . import lib as _l
from . import ops
from .config import cfg

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
N_ATOM_TYPES = 119


class ExpNormalSmearing(nn.Module):
    """Holds the non-trainable RBF constants (reference: models/utils.py:10-49, trainable=False)."""

    def __init__(self, cutoff_lower: float = 0.0, cutoff_upper: float = 5.0, num_rbf: int = 50):
        super().__init__()
        self.cutoff_lower, self.cutoff_upper, self.num_rbf = cutoff_lower, cutoff_upper, num_rbf
        start = torch.exp(torch.scalar_tensor(-cutoff_upper + cutoff_lower, dtype=torch.float32))
        self.register_buffer("means", torch.linspace(start, 1, num_rbf, dtype=torch.float32))
        self.register_buffer("betas", torch.tensor([(2 / num_rbf * (1 - start)) ** -2] * num_rbf, dtype=torch.float32))


class Encoder(nn.Module):
    """Parameter container for the atom / edge encoders (reference: models/cartnet.py:98-138)."""

    def __init__(self, dim_in: int, dim_rbf: int, radius: float = 5.0, invariant: bool = False,
                 temperature: bool = True, atom_types: bool = True):
        super().__init__()
        self.dim_in, self.invariant, self.temperature, self.atom_types = dim_in, invariant, temperature, atom_types
        if atom_types:
            self.embedding = nn.Embedding(N_ATOM_TYPES, dim_in * 2)
            nn.init.xavier_uniform_(self.embedding.weight.data)
        elif not temperature:
            self.embedding = nn.Embedding(1, dim_in)
        if temperature:
            self.temperature_proj_atom = nn.Linear(1, dim_in * 2, bias=True)
        elif atom_types:
            self.bias = nn.Parameter(torch.zeros(dim_in * 2))
        if temperature or atom_types:
            self.encoder_atom = nn.Sequential(nn.SiLU(), nn.Linear(dim_in * 2, dim_in), nn.SiLU())
        dim_edge = dim_rbf if invariant else dim_rbf + 3
        self.encoder_edge = nn.Sequential(nn.Linear(dim_edge, dim_in * 2), nn.SiLU(),
                                          nn.Linear(dim_in * 2, dim_in), nn.SiLU())
        self.rbf = ExpNormalSmearing(0.0, radius, dim_rbf)


class CartNet_layer(nn.Module):
    """Parameter container for one message-passing layer (reference: models/cartnet.py:180-201)."""

    def __init__(self, dim_in: int, use_envelope: bool = True):
        super().__init__()
        self.dim_in = dim_in
        self.MLP_aggr = nn.Sequential(nn.Linear(dim_in * 3, dim_in), nn.SiLU(), nn.Linear(dim_in, dim_in))
        self.MLP_gate = nn.Sequential(nn.Linear(dim_in * 3, dim_in), nn.SiLU(), nn.Linear(dim_in, dim_in))
        self.norm = nn.BatchNorm1d(dim_in)
        self.norm2 = nn.BatchNorm1d(dim_in)
        self.use_envelope = use_envelope
        self.envelope_radius = float(cfg.radius)   # the reference reads the global cfg here (cartnet.py:201)


class Cholesky_head(nn.Module):
    """reference: models/cartnet.py:285-291."""

    def __init__(self, dim_in: int):
        super().__init__()
        self.MLP = nn.Sequential(nn.Linear(dim_in, dim_in // 2), nn.SiLU(), nn.Linear(dim_in // 2, 6))


class Scalar_head(nn.Module):
    """reference: models/cartnet.py:315-321."""

    def __init__(self, dim_in: int):
        super().__init__()
        self.MLP = nn.Sequential(nn.Linear(dim_in, dim_in // 2), nn.SiLU(), nn.Linear(dim_in // 2, 1))


def _fill_params(dst: "_l.Params", tensors: Dict[str, torch.Tensor], L: int) -> None:
    """Point a CartnetParams struct at the tensors of a reference-layout name -> tensor mapping."""
    for field, key in _l.PARAM_KEYS.items():
        t = tensors.get(key)
        setattr(dst, field, t.data_ptr() if t is not None else None)
    for l in range(L):
        lay = dst.layer[l]
        for field, suffix in _l.LAYER_PARAM_KEYS.items():
            setattr(lay, field, tensors[f"layers.{l}.{suffix}"].data_ptr())


class _GradBuffer(dict):
    """All parameter gradients of one backward pass as views of ONE flat buffer (parameter order), so the hand-off to
    the optimiser's flat gradient buffer is a single add instead of one autograd accumulation per parameter."""

    def __init__(self, model, dev, flat=None):
        super().__init__()
        total = sum(math.prod(sh) if len(sh) else 1 for sh in model._param_shapes.values())
        self.flat = torch.empty(total, dtype=torch.float32, device=dev) if flat is None else flat
        off = 0
        for name, sh in model._param_shapes.items():
            k = math.prod(sh) if len(sh) else 1
            self[name] = self.flat[off:off + k].view(sh)
            off += k


class _CartNetFunction(torch.autograd.Function):
    """Whole-network forward / backward: one call into libcartnet_hip.so each (cartnet_model_forward /
    cartnet_model_backward, include/cartnet_hip.h).  ``params`` follow ``model._param_names``."""

    @staticmethod
    def forward(ctx, model: "CartNet", batch, training: bool, *params):
        lib = _l.load()
        dev = params[0].device
        # (under torch.no_grad() the flags of ctx.needs_input_grad still mirror requires_grad, and inside a Function's forward
        #  the grad mode is always off: the module's forward() records the caller's grad mode in model._grad_mode)
        need_grad = bool(getattr(model, "_grad_mode", True)) and any(ctx.needs_input_grad)   # False in eval loops: nothing is kept for backward
        z = batch.x
        if not (torch.is_tensor(z) and z.dtype == torch.int64 and z.dim() == 1):
            raise ValueError("batch.x must hold int64 atomic numbers [N] (forward overwrites it with features; "
                             "clone the batch to run it twice, as the reference's montecarlo does, main.py:87)")
        N, D = int(z.shape[0]), model.dim_in
        ei = batch.edge_index
        if not (ei.dtype == torch.int64 and ei.dim() == 2 and ei.shape[0] == 2):
            raise ValueError("batch.edge_index must be int64 [2, E]")
        E = int(ei.shape[1])
        Bg = int(batch.num_graphs)
        enc = model.encoder

        def dev_tensor(t, dtype, numel, name):
            if not (torch.is_tensor(t) and t.dtype == dtype and t.numel() == numel):
                raise ValueError(f"batch.{name}: expected {dtype} with {numel} elements, got "
                                 f"{getattr(t, 'dtype', None)} {tuple(getattr(t, 'shape', ()))}")
            if t.device != dev:
                raise ValueError(f"batch.{name} is on {t.device}, the model on {dev}: call batch.to(device) first")
            return t.contiguous()

        keep = [dev_tensor(z, torch.int64, N, "x"), dev_tensor(batch.batch, torch.int64, N, "batch"),
                dev_tensor(batch.ptr, torch.int64, Bg + 1, "ptr"), dev_tensor(ei, torch.int64, 2 * E, "edge_index"),
                dev_tensor(batch.cart_dist, torch.float32, E, "cart_dist")]
        bd = _l.BatchDesc()
        bd.z, bd.batch, bd.graph_ptr, bd.edge_index, bd.cart_dist = (t.data_ptr() for t in keep)
        if not enc.invariant:
            keep.append(dev_tensor(batch.cart_dir, torch.float32, 3 * E, "cart_dir"))
            bd.cart_dir = keep[-1].data_ptr()
        if enc.temperature:
            keep.append(dev_tensor(batch.temperature, torch.float32, Bg, "temperature"))
            bd.temperature = keep[-1].data_ptr()
        if model.cholesky:
            M = int(batch.y.shape[0])
            keep.append(dev_tensor(batch.non_H_mask, torch.bool, N, "non_H_mask"))
            bd.non_h_mask = keep[-1].data_ptr()
            pred = torch.empty((M, 3, 3), dtype=torch.float32, device=dev)
        else:
            M = 0
            pred = torch.empty((Bg,), dtype=torch.float32, device=dev)
        bd.N, bd.Bg, bd.M, bd.E = N, Bg, M, E

        # (with a FlatAdam attached only ONE parameter travels through autograd -- enough for backward to be called; the
        #  gradients go to the optimiser's flat buffer, not back through these inputs: CartNet.forward)
        ctx.n_params = len(params)
        if len(params) != len(model._param_names):
            params = model._params_list()
        md = model._model_desc(dict(zip(model._param_names, params)))
        nbytes = int(lib.cartnet_workspace_bytes(C.byref(md), N, E, Bg, M, int(need_grad)))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        sync_cb = None
        if model.sync_batchnorm and training:
            if model.bn_group_size > 0:
                raise ValueError("sync_batchnorm and bn_group_size are mutually exclusive")
            sync_cb = _make_bn_allreduce(ws)
            md.bn_allreduce = sync_cb
            keep.append(sync_cb)            # the C side calls it again in backward: must outlive ctx.saved
        x_out = torch.empty((N, D), dtype=torch.float32, device=dev)
        e_store = torch.empty((max(E, 1), D), dtype=torch.float32, device=dev)   # never a null pointer, even for E = 0
        e_out = e_store[:E]
        status = torch.empty(1, dtype=torch.int32, device=dev)
        _l.check(lib.cartnet_model_forward(C.byref(md), C.byref(bd), ws.data_ptr(), nbytes, int(training),
                                           int(need_grad), pred.data_ptr(), x_out.data_ptr(), e_store.data_ptr(),
                                           status.data_ptr(), _l.stream_ptr(), model._aux_stream_ptr(dev)),
                 "cartnet_model_forward")
        _raise_callback_error(sync_cb)
        if model.validate_graph:
            ops.raise_on_graph_status(int(status.item()))
        else:
            model._defer_graph_check(status)
        if need_grad:
            ctx.saved = (model, md, bd, ws, nbytes, keep, x_out, bool(training))
        else:
            ctx.saved = None
        ctx.mark_non_differentiable(x_out, e_out)
        ctx.set_materialize_grads(False)   # no zero-filled [N, D] / [E, D] gradients for the two feature outputs
        return pred, x_out, e_out

    @staticmethod
    def backward(ctx, dpred, _dx_unused, _de_unused):
        if ctx.saved is None:
            raise RuntimeError("CartNet backward called without saved state (or called twice: the saved activations "
                               "are consumed in place)")
        model, md, bd, ws, nbytes, keep, x_out, training = ctx.saved
        ctx.saved = None
        lib = _l.load()
        if dpred is None:
            raise RuntimeError("CartNet backward: the loss does not depend on the prediction")
        dpred = dpred.contiguous()
        sink = model._flat_grad
        # The optimiser's flat gradient buffer was zeroed by zero_grad() and nothing has been added since (FlatAdam.fresh):
        # the C side writes every gradient straight into it -- no staging buffer, no accumulation launch.  A second
        # backward before the next zero_grad() (gradient accumulation) takes the staging path below.
        owner = model.__dict__.get("_flat_owner")
        opt = owner() if owner is not None else None
        direct = opt is not None and getattr(opt, "fresh", False) and sink is not None and opt.flat_grad is sink and \
            sink.device == dpred.device
        cache = model.__dict__.get("_grad_direct" if direct else "_grad_cache")
        if direct:
            if cache is None or cache[0].flat is not sink:
                G = _GradBuffer(model, dpred.device, flat=sink)
                if G.flat.numel() != sum(v.numel() for v in G.values()):
                    raise RuntimeError("CartNet backward: the optimiser's flat buffer does not match the parameters")
                gd = _l.Params()
                _fill_params(gd, G, model.num_layers)
                cache = (G, gd)
                model.__dict__["_grad_direct"] = cache
            G, gd = cache
            opt.fresh = False
        elif sink is not None and cache is not None and cache[0].flat.device == dpred.device and \
                cache[0].flat.numel() == sink.numel():
            G, gd = cache              # reused: its content is added to the sink below, in stream order, before the next use
        else:
            G = _GradBuffer(model, dpred.device)
            gd = _l.Params()
            _fill_params(gd, G, model.num_layers)
            if sink is not None and sink.numel() == G.flat.numel():
                model.__dict__["_grad_cache"] = (G, gd)
        aux = model._aux_stream_ptr(dpred.device)
        # gradient buckets (model.grad_sync: a cartnet_amd.distributed.GradSync armed for THIS backward by the training
        # loop at an accumulation boundary): as soon as a bucket's kernels are enqueued, its slice is accumulated into the
        # optimiser's flat buffer and its all-reduce queued -- on the stream the C side names, under the rest of backward
        sync = getattr(model, "grad_sync", None)
        bucketed = sync is not None and sink is not None and sink.numel() == G.flat.numel() and sync.flat is sink
        gr_cb = None
        if bucketed:
            gr_cb = _make_grad_ready(model, sync, sink, None if direct else G.flat)
            md.grad_ready = gr_cb
        rc = lib.cartnet_model_backward(C.byref(md), C.byref(bd), ws.data_ptr(), nbytes, int(training),
                                        dpred.data_ptr(), x_out.data_ptr(), C.byref(gd), _l.stream_ptr(), aux)
        for k in keep:                     # an exception inside a callback is the cause: raise it, not the C side's echo
            _raise_callback_error(k if isinstance(k, _l.ALLREDUCE_FN) else None)
        _raise_callback_error(gr_cb)
        _l.check(rc, "cartnet_model_backward")
        if bucketed:
            return (None, None, None) + (None,) * ctx.n_params
        if direct:
            return (None, None, None) + (None,) * ctx.n_params
        if sink is not None and sink.numel() == G.flat.numel():
            sink.add_(G.flat)              # one accumulation into the optimiser's flat gradient buffer
            if opt is not None:
                opt.fresh = False
            return (None, None, None) + (None,) * ctx.n_params
        if ctx.n_params != len(model._param_names):
            raise RuntimeError("CartNet backward: the optimiser's flat gradient buffer went away between forward and backward")
        return (None, None, None) + tuple(G[name] for name in model._param_names)


def _make_bn_allreduce(ws: torch.Tensor):
    """CartnetModel.bn_allreduce for one forward/backward pair: the C side hands over a pointer into the workspace; the
    2D+1 doubles behind it are SUM-all-reduced over the ranks with torch.distributed (RCCL on GPUs) in stream order."""
    from . import distributed as cdist
    base, nbytes = ws.data_ptr(), ws.numel()
    err = []

    def cb(_user, buf, count, _stream):
        try:
            off = int(buf) - base
            if off < 0 or off + 8 * int(count) > nbytes or off % 8:
                raise RuntimeError("sync-BatchNorm buffer outside the workspace")
            cdist.all_reduce_sum_(ws[off:off + 8 * int(count)].view(torch.float64))
            return 0
        except Exception as exc:        # must not propagate through the C frames
            err.append(exc)
            return 1

    fn = _l.ALLREDUCE_FN(cb)
    fn._cartnet_errors = err
    return fn


def _bucket_ranges(model) -> List[tuple]:
    """(lo, hi) of every gradient bucket in the flat parameter order: index 0..L-1 = layers, L = head, L+1 = encoder
    (CartnetGradReadyFn).  named_parameters() lists encoder.*, layers.0.* .. layers.L-1.*, head.* -- contiguous ranges."""
    cached = model.__dict__.get("_bucket_cache")
    if cached is not None:
        return cached
    L = model.num_layers
    spans: Dict[int, list] = {}
    off = 0
    for n in model._param_names:
        cnt = 1
        for d in model._param_shapes[n]:
            cnt *= d
        b = int(n.split(".")[1]) if n.startswith("layers.") else (L if n.startswith("head.") else L + 1)
        sp = spans.setdefault(b, [off, off])
        if sp[1] != off:
            raise RuntimeError(f"parameters of bucket {b} are not contiguous in the flat buffer ({n})")
        sp[1] = off + cnt
        off += cnt
    out = [tuple(spans[b]) for b in range(L + 2)]
    model.__dict__["_bucket_cache"] = out
    return out


def _make_grad_ready(model, sync, sink: torch.Tensor, gflat: torch.Tensor):
    """CartnetModel.grad_ready for one backward: bucket -> accumulate its slice into the optimiser's flat gradient buffer and
    hand it to the GradSync, both on the stream the C side passes (ordered behind the kernels that wrote the slice)."""
    ranges = _bucket_ranges(model)
    err = []

    def cb(_user, bucket, stream):
        try:
            lo, hi = ranges[int(bucket)]
            # (the caller's stream may be the legacy default stream: a null hipStream_t)
            ext = torch.cuda.ExternalStream(int(stream), device=sink.device) if stream else torch.cuda.default_stream(sink.device)
            with torch.cuda.stream(ext):
                if gflat is not None:          # (None: the kernels wrote the fresh flat buffer itself)
                    sink[lo:hi].add_(gflat[lo:hi])
                sync.bucket(lo, hi)
            return 0
        except Exception as exc:        # must not propagate through the C frames
            err.append(exc)
            return 1

    fn = _l.GRADREADY_FN(cb)
    fn._cartnet_errors = err
    return fn


def _raise_callback_error(fn) -> None:
    errs = getattr(fn, "_cartnet_errors", None)
    if errs:
        raise errs.pop(0)


class CartNet(nn.Module):
    """CartNet (reference: models/cartnet.py:14-73) on hand-written gfx950 kernels.

    Args mirror the reference constructor: dim_in, dim_rbf, num_layers, radius=5.0, invariant=False,
    temperature=True, use_envelope=True, atom_types=True, cholesky=True.
    ``forward(batch)`` returns ``(pred, true)`` with pred [M,3,3] (Cholesky head) or [Bg] (scalar head) and
    ``true = batch.y``; like the reference it replaces ``batch.x`` / ``batch.edge_attr`` with the final features.
    """

    def __init__(self, dim_in: int, dim_rbf: int, num_layers: int, radius: float = 5.0, invariant: bool = False,
                 temperature: bool = True, use_envelope: bool = True, atom_types: bool = True,
                 cholesky: bool = True):
        super().__init__()
        if dim_in % 8 != 0:
            raise ValueError("dim_in must be a multiple of 8 (rows are moved as 16-byte vectors; the head is dim_in/2)")
        if dim_in // 2 > 512:
            raise ValueError("dim_in up to 1024 is supported by the head kernels")
        if not (1 <= num_layers <= _l.MAX_LAYERS):
            raise ValueError(f"num_layers must be in 1..{_l.MAX_LAYERS}")
        self.encoder = Encoder(dim_in, dim_rbf=dim_rbf, radius=radius, invariant=invariant, temperature=temperature,
                               atom_types=atom_types)
        self.dim_in = dim_in
        self.num_layers = num_layers
        self.layers = nn.Sequential(*[CartNet_layer(dim_in, use_envelope) for _ in range(num_layers)])
        self.cholesky = cholesky
        self.head = Cholesky_head(dim_in) if cholesky else Scalar_head(dim_in)
        self.validate_graph = False     # set True to sync-check edge_index ordering / ranges once per batch
        self.gemm_precision = 0         # 0: fp32 MFMA.  1: bf16x3 split-operand MFMA.  2: plain bf16 operands (csrc/gemm_x3.h)
        self.overlap_weight_gradients = True   # run weight-gradient GEMMs on a second stream during backward
        # > 0: consecutive crystals of a batch form BatchNorm groups of this size -- the reference recipe's micro-batches
        # (batch 4 x accumulation 16, scripts/train_cartnet_adp.sh:4) travel through the network as ONE batch of 64 with
        # per-micro-batch statistics; pair it with cartnet_amd.train.grouped_loss (include/cartnet_hip.h: CartnetGroups)
        self.bn_group_size = 0
        # True: every BatchNorm's statistics (forward) and gradient sums (backward) are summed over the data-parallel ranks
        # (torch.distributed), so that N ranks with a shard each compute what one process would on the union batch
        # (SURVEY.md 8e "sync-BN", optional; default = per-rank statistics, the reference's single-process semantics)
        self.sync_batchnorm = False
        # gemm_precision 2 only: pre / gs / dpre [E, 2D] of every layer are kept in the workspace as bf16 ("bf16 storage /
        # fp32 accumulate"): half the bytes of the tensors that dominate the step's HBM traffic
        self.half_storage = False
        self._aux_stream = None
        self._status_ring = None        # in-flight pinned copies of the batches' graph status words (_defer_graph_check)
        self._param_names = [n for n, _ in self.named_parameters()]
        self._param_shapes = {n: tuple(p.shape) for n, p in self.named_parameters()}
        self._flat_grad = None          # set by cartnet_amd.optim.FlatAdam: gradients are accumulated here directly
        self.grad_sync = None           # a cartnet_amd.distributed.GradSync armed for the next backward (train_epoch)

    def _model_desc(self, P: Dict[str, torch.Tensor]) -> "_l.Model":
        """CartnetModel struct pointing at the current parameters / buffers (reference state_dict layout).  Cached while
        every parameter / buffer address and every switch it encodes are what they were (the optimiser updates in place):
        filling ~120 ctypes fields costs 0.1 ms, a tenth of the host time of a step at configs[2] shapes."""
        B = self._buffers_dict()
        key = (tuple(t.data_ptr() for t in P.values()), tuple(t.data_ptr() for t in B.values()),
               int(self.gemm_precision), int(self.bn_group_size), bool(self.half_storage),
               tuple(bool(l.use_envelope) for l in self.layers), float(self.layers[0].envelope_radius))
        cached = self.__dict__.get("_md_cache")
        if cached is None or cached[0] != key:
            cached = (key, self._model_desc_build(P))
            self.__dict__["_md_cache"] = cached
        # the cache is a TEMPLATE: every forward gets its own copy, which ctx.saved then owns -- the per-call fields
        # (bn_allreduce: bound to THIS call's workspace, NULL for an eval forward) must not change under a backward
        # that is still to run (two training forwards, then two backwards; an eval forward in between)
        md = _l.Model()
        C.memmove(C.byref(md), C.byref(cached[1]), C.sizeof(md))
        return md

    def _model_desc_build(self, P: Dict[str, torch.Tensor]) -> "_l.Model":
        enc = self.encoder
        md = _l.Model()
        md.D, md.R, md.L = self.dim_in, enc.rbf.num_rbf, self.num_layers
        md.invariant, md.use_temperature, md.atom_types = int(enc.invariant), int(enc.temperature), int(enc.atom_types)
        md.cholesky, md.n_types = int(self.cholesky), N_ATOM_TYPES
        md.radius, md.env_radius = float(enc.rbf.cutoff_upper), float(self.layers[0].envelope_radius)
        md.bn_eps, md.bn_momentum = BN_EPS, BN_MOMENTUM
        md.gemm_precision = int(self.gemm_precision)
        md.bn_group_size = int(self.bn_group_size)
        if self.half_storage and (int(self.gemm_precision) != 2 or self.bn_group_size > 0 or self.dim_in % 256 != 0):
            raise ValueError("half_storage needs gemm_precision = 2, dim_in % 256 == 0 and no BatchNorm groups")
        md.half_storage = int(bool(self.half_storage))
        B = self._buffers_dict()
        md.rbf_means, md.rbf_betas = B["encoder.rbf.means"].data_ptr(), B["encoder.rbf.betas"].data_ptr()
        for n, t in P.items():
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError(f"parameter {n} must be a contiguous fp32 CUDA tensor")
        _fill_params(md.p, P, self.num_layers)
        for l in range(self.num_layers):
            md.use_envelope[l] = int(self.layers[l].use_envelope)
            bl = md.buf[l]
            for field, key in (("norm_mean", "norm.running_mean"), ("norm_var", "norm.running_var"),
                               ("norm_nbt", "norm.num_batches_tracked"), ("norm2_mean", "norm2.running_mean"),
                               ("norm2_var", "norm2.running_var"), ("norm2_nbt", "norm2.num_batches_tracked")):
                setattr(bl, field, B[f"layers.{l}.{key}"].data_ptr())
        return md

    def grad_bucket_order(self) -> List[tuple]:
        """(lo, hi) slices of the flat parameter / gradient buffer in the order cartnet_model_backward reports them
        (head, layers L-1 .. 0, encoder).  A rank without crystals for a step issues its zero contribution to the
        bucketed all-reduce in exactly this order (cartnet_amd.train.train_epoch)."""
        r = _bucket_ranges(self)
        L = self.num_layers
        return [r[L]] + [r[l] for l in range(L - 1, -1, -1)] + [r[L + 1]]

    _STATUS_DEPTH = 64

    def _defer_graph_check(self, status: torch.Tensor) -> None:
        """Asynchronous check of the device status word of cartnet_csr_build (edge order / index range / crystal
        membership) and of cartnet_node_embed (atomic numbers / batch ids inside their tables): the word is copied to pinned host memory behind the forward's kernels and read at a later forward
        call, once its copy has arrived -- a malformed batch raises without a host sync on the hot path (the host
        only ever waits when 64 forward calls are still in flight).  ``validate_graph = True`` checks every batch at
        once, with one sync per batch; ``flush_graph_checks()`` drains what is in flight.  The kernels clamp every
        index they gather through (edge endpoints, atomic numbers, batch ids), so a bad batch cannot fault in the
        meantime."""
        if self._status_ring is None:                # (in flight: [pinned word, event] oldest first, free pinned words)
            pool = torch.zeros(self._STATUS_DEPTH + 1, dtype=torch.int32).pin_memory()    # one host allocation
            self._status_ring = ([], [pool[i:i + 1] for i in range(self._STATUS_DEPTH + 1)])
        pending, free = self._status_ring
        bad = None
        while pending and (len(pending) >= self._STATUS_DEPTH or pending[0][1].query()):
            word, ev = pending.pop(0)
            ev.synchronize()
            if bad is None and int(word.item()) != 0:
                bad = int(word.item())
            free.append(word)
        word = free.pop()
        ev = torch.cuda.Event()
        aux = self._aux_stream if self.overlap_weight_gradients else None
        if aux is not None and aux.device == status.device:
            # every writer of the word (the layout build on the main stream -- the side stream forked behind it --, the CSC
            # build and the atom encoder on the side stream) precedes this point of the SIDE stream: copy there, so the
            # main stream's chain of dependent launches is one short (5 us + a launch gap at configs[2] shapes)
            status.record_stream(aux)          # (the caching allocator must not hand the word out again before the copy)
            with torch.cuda.stream(aux):
                word.copy_(status, non_blocking=True)
                ev.record()
        else:
            word.copy_(status, non_blocking=True)
            ev.record()
        pending.append([word, ev])
        if bad is not None:
            try:
                ops.raise_on_graph_status(bad)
            except ValueError as exc:
                raise ValueError(f"{exc} (reported for a batch passed to an earlier forward call; set "
                                 "model.validate_graph = True to check every batch at once)") from None

    def flush_graph_checks(self) -> None:
        """Wait for every graph status word still in flight and raise if any batch was malformed.  The training /
        evaluation loops call this before each optimiser step and at the end of a pass (one host sync per optimiser
        step), so the last batches of an epoch are checked too and no gradient of a bad batch reaches the weights."""
        if self._status_ring is None:
            return
        pending, free = self._status_ring
        bad = None
        while pending:
            word, ev = pending.pop(0)
            ev.synchronize()
            if bad is None and int(word.item()) != 0:
                bad = int(word.item())
            free.append(word)
        if bad is not None:
            ops.raise_on_graph_status(bad)

    def _aux_stream_ptr(self, dev):
        """Second HIP stream for the parameter-gradient work of backward (None -> single stream)."""
        if not self.overlap_weight_gradients:
            return None
        if self._aux_stream is None or self._aux_stream.device != dev:
            self._aux_stream = torch.cuda.Stream(device=dev)
        return self._aux_stream.cuda_stream

    def _apply(self, fn, *a, **kw):
        # .to() / .cuda() / .float(): drop the cached parameter / buffer lists (tensors may be replaced)
        self._cached_params = None
        self._cached_buffers = None
        return super()._apply(fn, *a, **kw)

    def _params_list(self):
        """Parameters in ``named_parameters()`` order, cached: walking the module tree costs 0.3 ms per call, a fifth of
        the host time of a step at configs[2] shapes.  Invalidated by ``_apply``; the length check catches edits."""
        ps = getattr(self, "_cached_params", None)
        if ps is None or len(ps) != len(self._param_names):
            ps = [p for _, p in self.named_parameters()]
            self._cached_params = ps
        return ps

    def _buffers_dict(self):
        bs = getattr(self, "_cached_buffers", None)
        if bs is None:
            bs = dict(self.named_buffers())
            self._cached_buffers = bs
        return bs

    def forward(self, batch):
        params = self._params_list()
        if not params[0].is_cuda:
            raise RuntimeError("cartnet_amd.CartNet runs only on an AMD GPU (HIP kernels); move the model and the "
                               "batch to 'cuda' -- there is no CPU fallback")
        self._grad_mode = torch.is_grad_enabled()       # read by _CartNetFunction.forward (grad mode is off in there)
        owner = self.__dict__.get("_flat_owner")
        if self._flat_grad is not None and owner is not None and owner() is None:
            self._flat_grad = None                      # the FlatAdam that owned the buffer is gone: gradients go to .grad again
            self.__dict__.pop("_flat_owner", None)
        if self._flat_grad is not None and self._flat_grad.device == params[0].device and torch.is_grad_enabled():
            # every gradient goes to the optimiser's flat buffer (FlatAdam): one differentiable input is enough to have
            # backward called, and 60 fewer arguments through autograd are 0.05 ms of host time per step.  FlatAdam was
            # built over ALL parameters (optim.py); one frozen since then would still receive its gradient and be stepped
            if not all(p.requires_grad for p in params):
                raise RuntimeError("a parameter was frozen after FlatAdam took the model's gradients over: build the "
                                   "optimiser after freezing (it then accumulates through autograd instead)")
            pred, x, e = _CartNetFunction.apply(self, batch, self.training, params[0])
        else:
            pred, x, e = _CartNetFunction.apply(self, batch, self.training, *params)
        batch.x = x
        batch.edge_attr = e
        return pred, batch.y


def make_state_dict(dim_in: int, dim_rbf: int, num_layers: int, seed: int = 0, cholesky: bool = True,
                    temperature: bool = True, atom_types: bool = True, invariant: bool = False,
                    radius: float = 5.0) -> Dict[str, torch.Tensor]:
    """Deterministic reference-shaped state_dict (CPU tensors) from a seed: used by the parity fixtures so the 10 MB
    D=256 weights need not be committed.  BatchNorm affine/running stats are randomised so eval mode is exercised."""
    torch.manual_seed(seed)
    m = CartNet(dim_in, dim_rbf, num_layers, radius=radius, invariant=invariant, temperature=temperature,
                atom_types=atom_types, cholesky=cholesky)
    g = torch.Generator().manual_seed(seed + 1)
    sd = m.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
        elif ".norm" in k and k.endswith(".weight"):
            v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
        elif ".norm" in k and k.endswith(".bias"):
            v.copy_(0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith(".bias") and v.dim() == 1 and "encoder.bias" == k:
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
    return {k: v.clone() for k, v in sd.items()}
