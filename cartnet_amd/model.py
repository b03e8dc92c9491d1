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


def _split_k(K: int, tiles: int) -> int:
    """Workgroups along the reduction of a weight-gradient GEMM: aim at ~512 workgroups, >= 256 rows each."""
    s = max(1, min((512 + tiles - 1) // tiles, K // 256))
    return int(s)


class _Ctx:
    """Per-call state shared between the forward and backward halves (kept on the autograd ctx)."""


def _empty(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _parts(n, dev):
    """fp64 partial-sum rows (see include/cartnet_hip.h, "Partial sums")."""
    return torch.empty((n,), dtype=torch.float64, device=dev)


class _GradBuffer(dict):
    """All parameter gradients of one backward pass as views of ONE flat buffer (parameter order), so the hand-off to
    the optimiser's flat gradient buffer is a single add instead of one autograd accumulation per parameter."""

    def __init__(self, model, dev):
        super().__init__()
        shapes = model._param_shapes
        total = sum(math.prod(sh) if len(sh) else 1 for sh in shapes.values())
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for name, sh in shapes.items():
            k = math.prod(sh) if len(sh) else 1
            dict.__setitem__(self, name, self.flat[off:off + k].view(sh))
            off += k
        self.written = set()

    def out(self, name):
        """The view a kernel should write the gradient of ``name`` into."""
        self.written.add(name)
        return dict.__getitem__(self, name)

    def put(self, name, value):
        self.out(name).copy_(value.reshape(dict.__getitem__(self, name).shape))


def _finalize(parts: torch.Tensor, nparts: int, n: int) -> torch.Tensor:
    out = _empty((n,), parts.device)
    ops.colsum_finalize(parts, nparts, out)
    return out


def _wgrad(dY: List[torch.Tensor], X: List[torch.Tensor], outs: List[torch.Tensor], b_act: bool = False) -> None:
    """outs[g] = dY[g]^T @ (silu?)(X[g]) with the row reduction split over workgroups and summed in fixed order."""
    K, M = dY[0].shape
    N = X[0].shape[1]
    tiles = len(dY) * ((M + 127) // 128) * ((N + 255) // 256 if N > 128 else 1)
    S = _split_k(K, tiles)
    if S == 1:
        ops.gemm(dY, X, outs, a_kstrided=True, b_kstrided=True, b_act=b_act)
        return
    slabs = [_empty((S * M, N), dY[0].device) for _ in dY]
    ops.gemm(dY, X, slabs, a_kstrided=True, b_kstrided=True, b_act=b_act, splitk=S)
    ops.splitk_reduce(slabs, S, outs)


class _CartNetFunction(torch.autograd.Function):
    """Whole-network forward / backward on the HIP kernels.  ``params`` follow ``model._param_names``."""

    @staticmethod
    def forward(ctx, model: "CartNet", batch, training: bool, *params):
        P: Dict[str, torch.Tensor] = dict(zip(model._param_names, params))
        B: Dict[str, torch.Tensor] = dict(model.named_buffers())
        need_grad = any(ctx.needs_input_grad)   # False under no_grad / eval loops: nothing is kept for backward
        D, L = model.dim_in, model.num_layers
        dev = params[0].device
        st = _Ctx()
        st.model, st.training, st.P = model, training, P

        z, gid = batch.x, batch.batch
        if z.dtype != torch.int64 or z.dim() != 1:
            raise ValueError("batch.x must hold int64 atomic numbers [N] (forward overwrites it with features; "
                             "clone the batch to run it twice, as the reference's montecarlo does, main.py:87)")
        N = int(z.shape[0])
        E = int(batch.edge_index.shape[1])
        lay = getattr(batch, "_cartnet_layout", None)
        if lay is None or lay.N != N or lay.E != E:
            gptr = getattr(batch, "ptr", None)
            lay = ops.GraphLayout(batch.edge_index, N, gptr.to(dev) if gptr is not None else None)
            if model.validate_graph:
                lay.validate()
            batch._cartnet_layout = lay
        st.lay, st.N, st.E = lay, N, E
        dist = batch.cart_dist.contiguous()
        enc = model.encoder
        # weights as [in, out] for the forward GEMMs (k-strided B operand = coalesced weight rows); one small launch set
        tnames = [n for n in model._param_names if n.endswith(".weight") and P[n].dim() == 2 and n != "head.MLP.2.weight"
                  and ("MLP" in n or "encoder_edge" in n or "encoder_atom" in n)]
        WT = dict(zip(tnames, ops.transpose([P[n] for n in tnames])))

        # ---- encoder, edges: Cartesian features -> Linear -> SiLU -> Linear -> SiLU   (cartnet.py:159)
        R = enc.rbf.num_rbf
        kf = R if enc.invariant else R + 3
        ldf = (kf + 3) // 4 * 4
        feat = _empty((E, ldf), dev)
        env = _empty((max(E, 1),), dev)
        ops.edge_features(dist, None if enc.invariant else batch.cart_dir.contiguous(), B["encoder.rbf.means"],
                          B["encoder.rbf.betas"], enc.invariant, enc.rbf.cutoff_upper,
                          model.layers[0].envelope_radius if L else enc.rbf.cutoff_upper, feat, env)
        he_pre = _empty((E, 2 * D), dev)
        ops.gemm(feat[:, :kf], WT["encoder.encoder_edge.0.weight"], he_pre, b_kstrided=True,
                 bias=P["encoder.encoder_edge.0.bias"])
        e0_pre = _empty((E, D), dev)
        e = _empty((E, D), dev)
        ops.gemm(he_pre, WT["encoder.encoder_edge.2.weight"], e, b_kstrided=True, a_act=True, out_act=True,
                 bias=P["encoder.encoder_edge.2.bias"], cpre=e0_pre)
        st.feat, st.kf, st.he_pre, st.e0_pre, st.env = feat, kf, he_pre, e0_pre, env

        # ---- encoder, atoms   (cartnet.py:145-154)
        st.has_atom_mlp = enc.temperature or enc.atom_types
        if st.has_atom_mlp:
            x0 = _empty((N, 2 * D), dev)
            T = batch.temperature.contiguous() if enc.temperature else None
            ops.node_embed(z if enc.atom_types else None, gid if enc.temperature else None, T,
                           P.get("encoder.embedding.weight") if enc.atom_types else None,
                           P.get("encoder.temperature_proj_atom.weight"), P.get("encoder.temperature_proj_atom.bias"),
                           P.get("encoder.bias"), x0)
            xa_pre = _empty((N, D), dev)
            x = _empty((N, D), dev)
            ops.gemm(x0, WT["encoder.encoder_atom.1.weight"], x, b_kstrided=True, a_act=True, out_act=True,
                     bias=P["encoder.encoder_atom.1.bias"], cpre=xa_pre)
            st.x0, st.xa_pre, st.gid, st.T = x0, xa_pre, gid, T
            if need_grad and enc.atom_types:   # atoms grouped by element (stable) for the embedding gradient
                st.zperm, st.zptr, _ = ops.sort_by_key(z, N_ATOM_TYPES)
        else:  # cartnet.py:150-151: one learned row for every atom
            x = P["encoder.embedding.weight"].detach().repeat(N, 1).contiguous()

        # ---- message-passing layers   (cartnet.py:204-274)
        st.layers = []
        tiles_e = ops.gemm_tiles_m(E)
        gparts = ops.gate_nparts(N)
        for l in range(L):
            p = f"layers.{l}"
            W1g, W1a = P[p + ".MLP_gate.0.weight"], P[p + ".MLP_aggr.0.weight"]
            W1gT, W1aT = WT[p + ".MLP_gate.0.weight"], WT[p + ".MLP_aggr.0.weight"]       # [3D, D]
            Pn = _empty((N, 4 * D), dev)          # node-side halves of the first Linears: [gate_i | aggr_i | gate_j | aggr_j]
            ops.gemm([x, x, x, x], [W1gT[:D], W1aT[:D], W1gT[D:2 * D], W1aT[D:2 * D]],
                     [Pn[:, 0:D], Pn[:, D:2 * D], Pn[:, 2 * D:3 * D], Pn[:, 3 * D:]], b_kstrided=True,
                     bias=[P[p + ".MLP_gate.0.bias"], P[p + ".MLP_aggr.0.bias"], None, None])
            pre = _empty((E, 2 * D), dev)         # [gate | sender] pre-activations of the first Linears
            ops.gemm([e, e], [W1gT[2 * D:], W1aT[2 * D:]], [pre[:, :D], pre[:, D:]], b_kstrided=True,
                     gather_i=[Pn[:, 0:D], Pn[:, D:2 * D]], gather_j=[Pn[:, 2 * D:3 * D], Pn[:, 3 * D:]],
                     tgt=lay.tgt, src=lay.src)
            gs = _empty((E, 2 * D), dev)          # [g (pre-BatchNorm gate) | s (sender)]
            cs, cq = _parts(tiles_e * D, dev), _parts(tiles_e * D, dev)
            ops.gemm([pre[:, :D], pre[:, D:]], [WT[p + ".MLP_gate.2.weight"], WT[p + ".MLP_aggr.2.weight"]],
                     [gs[:, :D], gs[:, D:]], b_kstrided=True, a_act=True, bias=[P[p + ".MLP_gate.2.bias"], P[p + ".MLP_aggr.2.bias"]],
                     colsum=[cs, None], colsq=[cq, None])
            mr1 = _empty((2 * D,), dev)
            ops.bn_finalize(cs, cq, tiles_e, E, D, BN_EPS, BN_MOMENTUM, training, B[p + ".norm.running_mean"],
                            B[p + ".norm.running_var"], B[p + ".norm.num_batches_tracked"], mr1)
            e_out, aggr = _empty((E, D), dev), _empty((N, D), dev)
            ps, pq = _parts(gparts * D, dev), _parts(gparts * D, dev)
            use_env = model.layers[l].use_envelope
            ops.gate_scatter_fwd(gs, e, env if use_env else None, lay, mr1, P[p + ".norm.weight"],
                                 P[p + ".norm.bias"], e_out, aggr, ps, pq)
            mr2 = _empty((2 * D,), dev)
            ops.bn_finalize(ps, pq, gparts, N, D, BN_EPS, BN_MOMENTUM, training, B[p + ".norm2.running_mean"],
                            B[p + ".norm2.running_var"], B[p + ".norm2.num_batches_tracked"], mr2)
            x_out = _empty((N, D), dev)
            ops.node_update_fwd(aggr, x, mr2, P[p + ".norm2.weight"], P[p + ".norm2.bias"], x_out)
            if need_grad:
                st.layers.append((x, e, pre, gs, mr1, aggr, mr2, use_env))
            x, e = x_out, e_out

        # ---- head
        H = D // 2
        hid = _empty((N, H), dev)
        ops.gemm(x, WT["head.MLP.0.weight"], hid, b_kstrided=True, bias=P["head.MLP.0.bias"])
        if model.cholesky:
            M = int(batch.y.shape[0])
            idx = getattr(batch, "_cartnet_mask_index", None)
            if idx is None or idx.numel() != N:
                idx = torch.empty(N, dtype=torch.int32, device=dev)
                ops.mask_index(batch.non_H_mask.contiguous(), idx, None)
                batch._cartnet_mask_index = idx
            p6, pred = _empty((M, 6), dev), _empty((M, 3, 3), dev)
            ops.cholesky_head_fwd(hid, idx, P["head.MLP.2.weight"], P["head.MLP.2.bias"], p6, pred)
            st.idx, st.p6 = idx, p6
        else:
            Bg = int(batch.num_graphs)
            gptr = batch.ptr.to(dev)
            pred = _empty((Bg,), dev)
            ops.scalar_head_fwd(hid, P["head.MLP.2.weight"], P["head.MLP.2.bias"], gptr, pred)
            st.gptr, st.gid = gptr, gid
        st.hid, st.x_final = hid, x
        ctx.st = st if need_grad else None
        ctx.n_params = len(params)
        ctx.mark_non_differentiable(x, e)
        return pred, x, e

    @staticmethod
    def backward(ctx, dpred, _dx_unused, _de_unused):
        st = ctx.st
        if st is None:
            raise RuntimeError("CartNet backward called without saved state")
        ctx.st = None            # single use: the saved activations are overwritten in place below
        model, P, lay, N, E = st.model, st.P, st.lay, st.N, st.E
        D, L, H = model.dim_in, model.num_layers, model.dim_in // 2
        dev = dpred.device
        training = st.training
        G = _GradBuffer(model, dev)
        dpred = dpred.contiguous()
        nparts_n = ops.node_nparts(N)

        # ---- head
        dhid = _empty((N, H), dev)
        if model.cholesky:
            row = 7 * H + 8
            parts = _empty((nparts_n * row,), dev)
            ops.cholesky_head_bwd(st.hid, st.idx, P["head.MLP.2.weight"], st.p6, dpred, dhid, parts)
            tot = _finalize(parts, nparts_n, row)
            G.put("head.MLP.2.weight", tot[:6 * H])
            G.put("head.MLP.2.bias", tot[6 * H:6 * H + 6])
            G.put("head.MLP.0.bias", tot[6 * H + 8:])
        else:
            row = 2 * H + 8
            parts = _empty((nparts_n * row,), dev)
            ops.scalar_head_bwd(st.hid, P["head.MLP.2.weight"], st.gptr, st.gid, dpred, dhid, parts)
            tot = _finalize(parts, nparts_n, row)
            G.put("head.MLP.2.weight", tot[:H])
            G.put("head.MLP.2.bias", tot[H:H + 1])
            G.put("head.MLP.0.bias", tot[H + 8:])
        _wgrad([dhid], [st.x_final], [G.out("head.MLP.0.weight")])
        dx = _empty((N, D), dev)
        ops.gemm(dhid, P["head.MLP.0.weight"], dx, b_kstrided=True)
        de = None   # the head does not read the edge features

        # ---- layers, last to first
        gparts = ops.gate_nparts(N)
        for l in reversed(range(L)):
            p = f"layers.{l}"
            x_in, e_in, pre, gs, mr1, aggr, mr2, use_env = st.layers[l]
            env = st.env if use_env else None
            W1g, W1a = P[p + ".MLP_gate.0.weight"], P[p + ".MLP_aggr.0.weight"]
            W2g, W2a = P[p + ".MLP_gate.2.weight"], P[p + ".MLP_aggr.2.weight"]
            # node update: x_out = silu(bn2(aggr)) + x_in
            pa, pb = _parts(nparts_n * D, dev), _parts(nparts_n * D, dev)
            ops.node_update_bwd_stats(aggr, dx, mr2, P[p + ".norm2.weight"], P[p + ".norm2.bias"], pa, pb)
            sums2 = _empty((2 * D,), dev)
            ops.colsum_finalize([pa, pb], nparts_n, [sums2[:D], sums2[D:]])
            G.put(p + ".norm2.bias", sums2[:D])
            G.put(p + ".norm2.weight", sums2[D:])
            daggr = _empty((N, D), dev)
            ops.node_update_bwd_apply(aggr, dx, mr2, P[p + ".norm2.weight"], P[p + ".norm2.bias"], sums2, training,
                                      daggr)
            # gate * sender aggregation and the edge BatchNorm
            pa, pb = _parts(gparts * D, dev), _parts(gparts * D, dev)
            ops.gate_scatter_bwd_stats(gs, de, daggr, env, lay, mr1, P[p + ".norm.weight"], P[p + ".norm.bias"], pa,
                                       pb)
            sums1 = _empty((2 * D,), dev)
            ops.colsum_finalize([pa, pb], gparts, [sums1[:D], sums1[D:]])
            G.put(p + ".norm.bias", sums1[:D])
            G.put(p + ".norm.weight", sums1[D:])
            pdg, pds = _parts(gparts * D, dev), _parts(gparts * D, dev)
            ops.gate_scatter_bwd_apply(gs, de, daggr, env, lay, mr1, P[p + ".norm.weight"], P[p + ".norm.bias"],
                                       sums1, training, pdg, pds)          # gs now holds [dg | ds]
            ops.colsum_finalize([pdg, pds], gparts, [G.out(p + ".MLP_gate.2.bias"), G.out(p + ".MLP_aggr.2.bias")])
            # second Linears: weight gradients need silu(pre), then pre is overwritten with dpre
            _wgrad([gs[:, :D], gs[:, D:]], [pre[:, :D], pre[:, D:]],
                   [G.out(p + ".MLP_gate.2.weight"), G.out(p + ".MLP_aggr.2.weight")], b_act=True)
            tiles_e = ops.gemm_tiles_m(E)
            csg, csa = _parts(tiles_e * D, dev), _parts(tiles_e * D, dev)
            ops.gemm([gs[:, :D], gs[:, D:]], [W2g, W2a], [pre[:, :D], pre[:, D:]], b_kstrided=True,
                     dact=[pre[:, :D], pre[:, D:]], colsum=[csg, csa])      # pre now holds dpre = [dpre_gate | dpre_aggr]
            ops.colsum_finalize([csg, csa], tiles_e, [G.out(p + ".MLP_gate.0.bias"), G.out(p + ".MLP_aggr.0.bias")])
            dW1g, dW1a = G.out(p + ".MLP_gate.0.weight"), G.out(p + ".MLP_aggr.0.weight")
            _wgrad([pre[:, :D], pre[:, D:]], [e_in, e_in], [dW1g[:, 2 * D:], dW1a[:, 2 * D:]])
            # edge features: de_in = de_out + dpre @ W1[:, 2D:]  (layer 0: continue through the encoder's last SiLU)
            de_in = _empty((E, D), dev)
            if l == 0:
                cse = _parts(tiles_e * D, dev)
                ops.gemm([pre[:, :D], pre[:, D:]], [W1g[:, 2 * D:], W1a[:, 2 * D:]], de_in, b_kstrided=True,
                         segments=True, resid=de, dact=st.e0_pre, colsum=cse)
                ops.colsum_finalize(cse, tiles_e, G.out("encoder.encoder_edge.2.bias"))
            else:
                ops.gemm([pre[:, :D], pre[:, D:]], [W1g[:, 2 * D:], W1a[:, 2 * D:]], de_in, b_kstrided=True,
                         segments=True, resid=de)
            # node-side halves: reduce dpre over each node's incoming (target) and outgoing (source) edges
            dPn = _empty((N, 4 * D), dev)
            ops.segment_sum(pre, lay.rowptr, None, dPn[:, :2 * D])
            ops.segment_sum(pre, lay.colptr, lay.perm, dPn[:, 2 * D:])
            _wgrad([dPn[:, 0:D], dPn[:, D:2 * D], dPn[:, 2 * D:3 * D], dPn[:, 3 * D:]], [x_in] * 4,
                   [dW1g[:, :D], dW1a[:, :D], dW1g[:, D:2 * D], dW1a[:, D:2 * D]])
            dx_in = _empty((N, D), dev)
            segsA = [dPn[:, 0:D], dPn[:, D:2 * D], dPn[:, 2 * D:3 * D], dPn[:, 3 * D:]]
            segsB = [W1g[:, :D], W1a[:, :D], W1g[:, D:2 * D], W1a[:, D:2 * D]]
            if l == 0 and st.has_atom_mlp:
                csx = _parts(ops.gemm_tiles_m(N) * D, dev)
                ops.gemm(segsA, segsB, dx_in, b_kstrided=True, segments=True, resid=dx, dact=st.xa_pre, colsum=csx)
                ops.colsum_finalize(csx, ops.gemm_tiles_m(N), G.out("encoder.encoder_atom.1.bias"))
            else:
                ops.gemm(segsA, segsB, dx_in, b_kstrided=True, segments=True, resid=dx)
            dx, de = dx_in, de_in
            st.layers[l] = None

        # ---- encoder.  With L == 0 the SiLU derivatives were not folded into a layer GEMM; the reference default is L=4.
        if L == 0:
            raise NotImplementedError("backward with num_layers == 0 is not supported")
        # edges: de now holds d(e0_pre)
        _wgrad([de], [st.he_pre], [G.out("encoder.encoder_edge.2.weight")], b_act=True)
        tiles_e = ops.gemm_tiles_m(E)
        cs = _parts(tiles_e * 2 * D, dev)
        ops.gemm(de, P["encoder.encoder_edge.2.weight"], st.he_pre, b_kstrided=True, dact=st.he_pre, colsum=cs)
        ops.colsum_finalize(cs, tiles_e, G.out("encoder.encoder_edge.0.bias"))
        _wgrad([st.he_pre], [st.feat[:, :st.kf]], [G.out("encoder.encoder_edge.0.weight")])
        # atoms: dx now holds d(xa_pre)
        enc = model.encoder
        if st.has_atom_mlp:
            _wgrad([dx], [st.x0], [G.out("encoder.encoder_atom.1.weight")], b_act=True)
            dx0 = _empty((N, 2 * D), dev)
            ops.gemm(dx, P["encoder.encoder_atom.1.weight"], dx0, b_kstrided=True, dact=st.x0)
            pw, pb = _parts(nparts_n * 2 * D, dev), _parts(nparts_n * 2 * D, dev)
            ops.node_embed_bwd(st.gid if enc.temperature else None, st.T, dx0, pw, pb)
            if enc.atom_types:
                ops.segment_sum_long(dx0, st.zptr, st.zperm, N, G.out("encoder.embedding.weight"))
            if enc.temperature:
                ops.colsum_finalize([pw, pb], nparts_n, [G.out("encoder.temperature_proj_atom.weight").view(-1),
                                                         G.out("encoder.temperature_proj_atom.bias")])
            else:
                ops.colsum_finalize(pb, nparts_n, G.out("encoder.bias"))
        else:
            # x = embedding row repeated for every atom: its gradient is the column sum of dx
            ones = torch.ones(N, 1, device=dev)
            _wgrad([ones], [dx], [G.out("encoder.embedding.weight")])

        missing = [n for n in model._param_names if n not in G.written]
        if missing:
            raise RuntimeError(f"backward produced no gradient for {missing}")
        sink = model._flat_grad
        if sink is not None and sink.numel() == G.flat.numel():
            sink.add_(G.flat)              # one accumulation into the optimiser's flat gradient buffer
            return (None, None, None) + (None,) * len(model._param_names)
        return (None, None, None) + tuple(G[name] for name in model._param_names)


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
        self.encoder = Encoder(dim_in, dim_rbf=dim_rbf, radius=radius, invariant=invariant, temperature=temperature,
                               atom_types=atom_types)
        self.dim_in = dim_in
        self.num_layers = num_layers
        self.layers = nn.Sequential(*[CartNet_layer(dim_in, use_envelope) for _ in range(num_layers)])
        self.cholesky = cholesky
        self.head = Cholesky_head(dim_in) if cholesky else Scalar_head(dim_in)
        self.validate_graph = False     # set True to sync-check edge_index ordering / ranges once per batch
        self._param_names = [n for n, _ in self.named_parameters()]
        self._param_shapes = {n: tuple(p.shape) for n, p in self.named_parameters()}
        self._flat_grad = None          # set by cartnet_amd.optim.FlatAdam: gradients are accumulated here directly

    def forward(self, batch):
        params = [p for _, p in self.named_parameters()]
        if not params[0].is_cuda:
            raise RuntimeError("cartnet_amd.CartNet runs only on an AMD GPU (HIP kernels); move the model and the "
                               "batch to 'cuda' -- there is no CPU fallback")
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
