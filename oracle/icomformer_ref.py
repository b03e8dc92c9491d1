"""ORACLE -- test infrastructure only, never the product path.

CPU restatement (plain torch, any float dtype) of the reference's iComformer forward pass
(models/comformer.py:75-132, models/comformer_conv.py:21-193, models/utils.py:96-129) as pure functions over a
``state_dict``-style mapping.  Pinned against the reference's own code run in the build container through
tests/golden/icomformer_*.npz (tests/golden/make_golden.py); autograd of these functions is the gradient reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .cartnet_ref import batch_norm, cholesky_head

Tensor = torch.Tensor


def rbf_expansion(v: Tensor, vmin: float, vmax: float, bins: int) -> Tensor:
    """models/utils.py:96-129: exp(-gamma (v - c_k)^2), centers = linspace(vmin, vmax, bins),
    gamma = 1 / mean(diff(centers)) = (bins - 1) / (vmax - vmin)."""
    c32 = torch.linspace(vmin, vmax, bins, dtype=torch.float32)
    gamma = float(1 / np.diff(c32.numpy()).mean())        # float32 arithmetic, exactly as the reference evaluates it
    return torch.exp(-gamma * (v.unsqueeze(1) - c32.to(v.dtype)) ** 2)


def _lin(x: Tensor, sd: Dict[str, Tensor], prefix: str, bias: bool = True) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


def _mlp(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    return _lin(F.silu(_lin(x, sd, prefix + ".0")), sd, prefix + ".2")


def bond_cosine(r1: Tensor, r2: Tensor) -> Tensor:
    """models/comformer.py:18-23."""
    c = torch.sum(r1 * r2, dim=-1) / (torch.norm(r1, dim=-1) * torch.norm(r2, dim=-1))
    return torch.clamp(c, -1, 1)


def comformer_conv(sd, p: str, x: Tensor, edge_index: Tensor, e: Tensor, training: bool, new_stats) -> Tensor:
    """models/comformer_conv.py:71-99 (heads = 1): per edge (j -> i): key' = key_update(cat[k_i, k_j, lin_edge(e)]),
    alpha = q_i * key' / sqrt(C), msg = lin_msg_update(cat[v_i, v_j, lin_edge(e)]) * sigmoid(bn_att(alpha)),
    scatter-add over the target, lin_concate, softplus(x + bn(out))."""
    C = x.shape[1]
    src, tgt = edge_index[0], edge_index[1]
    q, k, v = _lin(x, sd, p + ".lin_query"), _lin(x, sd, p + ".lin_key"), _lin(x, sd, p + ".lin_value")
    ea = _lin(e, sd, p + ".lin_edge")
    key = _mlp(torch.cat((k[tgt], k[src], ea), dim=-1), sd, p + ".key_update")
    alpha = (q[tgt] * key) / math.sqrt(C)
    msg = _mlp(torch.cat((v[tgt], v[src], ea), dim=-1), sd, p + ".lin_msg_update")
    msg = msg * torch.sigmoid(batch_norm(alpha, sd, p + ".bn_att", training, new_stats))
    out = torch.zeros(x.shape[0], C, dtype=x.dtype).scatter_add_(0, tgt.unsqueeze(-1).expand_as(msg), msg)
    out = _lin(out, sd, p + ".lin_concate")
    return F.softplus(x + batch_norm(out, sd, p + ".bn", training, new_stats))


def comformer_conv_edge(sd, p: str, e: Tensor, nei_len: Tensor, nei_angle: Tensor, training: bool, new_stats) -> Tensor:
    """models/comformer_conv.py:156-193: every edge attends to the three lattice vectors.
    e [E,C], nei_len [E,3,C], nei_angle [E,3,C]."""
    C = e.shape[1]
    q = _lin(e, sd, p + ".lin_query").unsqueeze(1)
    kx = _lin(e, sd, p + ".lin_key").unsqueeze(1).expand(-1, 3, -1)
    vx = _lin(e, sd, p + ".lin_value").unsqueeze(1).expand(-1, 3, -1)
    ky = torch.stack([_lin(nei_len[:, i, :], sd, p + f".lin_key_e{i + 1}") for i in range(3)], dim=1)
    vy = torch.stack([_lin(nei_len[:, i, :], sd, p + f".lin_value_e{i + 1}") for i in range(3)], dim=1)
    exy = _lin(nei_angle, sd, p + ".lin_edge", bias=False)
    key = _mlp(torch.cat((kx, ky, exy), dim=-1), sd, p + ".key_update")
    alpha = (q * key) / math.sqrt(C)
    out = _mlp(torch.cat((vx, vy, exy), dim=-1), sd, p + ".lin_msg_update")
    gate = torch.sigmoid(batch_norm(alpha.reshape(-1, C), sd, p + ".bn_att", training, new_stats)).reshape(-1, 3, C)
    out = _lin(out * gate, sd, p + ".lin_concate").sum(dim=1)
    return F.softplus(e + batch_norm(out, sd, p + ".bn", training, new_stats))


def icomformer_forward(sd: Dict[str, Tensor], batch, training: bool = False,
                       new_stats: Optional[Dict[str, Tensor]] = None, trace: Optional[dict] = None) -> Tensor:
    """models/comformer.py:115-132.  Does not mutate ``batch``."""
    C = sd["embedding.weight"].shape[1]
    dt = sd["embedding.weight"].dtype
    src = batch.edge_index[0]
    x = F.embedding(batch.x, sd["embedding.weight"]) + \
        _lin(batch.temperature.unsqueeze(-1), sd, "temperature_proj_atom")[batch.batch]
    edge_feat = -0.75 / batch.cart_dist
    nl = -0.75 / torch.norm(batch.cell, dim=-1)                         # [Bg,3]
    nl = nl[batch.batch[src]]                                            # [E,3]
    na = bond_cosine(batch.cell[batch.batch[src]], batch.cart_dir.unsqueeze(1).repeat(1, 3, 1))   # [E,3]
    E = edge_feat.shape[0]
    rbf = lambda v: F.softplus(_lin(rbf_expansion(v, -4.0, 0.0, C).to(dt), sd, "rbf.1"))
    rbf_a = lambda v: F.softplus(_lin(rbf_expansion(v, -1.0, 1.0, C).to(dt), sd, "rbf_angle.1"))
    e = rbf(edge_feat)
    nei_len = rbf(nl.reshape(-1)).reshape(E, 3, -1)
    nei_angle = rbf_a(na.reshape(-1)).reshape(E, 3, -1)
    if trace is not None:
        trace["x0"], trace["e0"] = x, e
    x = comformer_conv(sd, "att_layers.0", x, batch.edge_index, e, training, new_stats)
    e = comformer_conv_edge(sd, "edge_update_layer", e, nei_len, nei_angle, training, new_stats)
    if trace is not None:
        trace["x1"], trace["e1"] = x, e
    for l in (1, 2, 3):
        x = comformer_conv(sd, f"att_layers.{l}", x, batch.edge_index, e, training, new_stats)
        if trace is not None:
            trace[f"x{l + 1}"] = x
    head = {"head.MLP.0.weight": sd["cholesky.MLP.0.weight"], "head.MLP.0.bias": sd["cholesky.MLP.0.bias"],
            "head.MLP.2.weight": sd["cholesky.MLP.2.weight"], "head.MLP.2.bias": sd["cholesky.MLP.2.bias"]}
    return cholesky_head(head, x, batch.non_H_mask)
