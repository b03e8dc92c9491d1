"""ORACLE -- test infrastructure only, never the product path.  PARITY UNPINNED (see below).

CPU restatement (plain torch, any float dtype) of the reference's eComformer forward pass (models/comformer.py:25-70,
models/comformer_conv.py:197-280) as pure functions over a ``state_dict``-style mapping.  The attention layers and the
head are the ones of oracle/icomformer_ref.py (pinned to the reference's golden vectors).  The equivariant update
``ComformerConvEqui`` is built by the reference on **e3nn** (``o3.spherical_harmonics``,
``o3.FullyConnectedTensorProduct(..., shared_weights=False)``), a third-party dependency that is neither vendored under
/root/reference nor installed in the build image (environment.yml pins e3nn==0.5.1); it is restated here from e3nn's
published algorithm, and NO golden vector of the reference pins it -- "parity unpinned" for this block.

What e3nn computes for the irreps the reference passes (derivation):

* ``spherical_harmonics('1x0e + 1x1o + 1x2e', r, normalize=True, normalization='component')``: Y_0 = 1,
  Y_1 = sqrt(3) r_hat, Y_2 = sqrt(5) (orthonormal real l = 2 harmonics of r_hat): |Y_l|^2 = 2l + 1.  e3nn orders /
  signs its real basis in its own way; the composition below only ever contracts Y_l with Y_l (or multiplies a scalar
  by Y_l and later contracts that with Y_l again), and sum_m Y_lm(a) Y_lm(b) = (2l+1) P_l(a.b) in every orthonormal
  basis, so the result does not depend on the basis.
* ``FullyConnectedTensorProduct(in1, sh, out)``: one 'uvw' instruction per (in1 irrep, sh irrep, out irrep) with
  ir_out in ir_1 x ir_2, in that nesting order; per-edge weights of shape (mul_1, mul_2 = 1, mul_out), flattened in
  instruction order.  With irrep_normalization='component' and path_normalization='element' (the defaults) every
  instruction carries the factor sqrt(dim(ir_out) / sum over the instructions into the same output of mul_1 mul_2),
  and the Wigner symbols that occur are w3j(0, l, l)_{0jk} = delta_jk / sqrt(2l+1) and
  w3j(l, l, 0)_{ij0} = delta_ij / sqrt(2l+1).
  - layer 1, 64x0e (x) sh -> 64x0e + 8x1o + 8x2e: three instructions (0e.0e->0e, 0e.1o->1o, 0e.2e->2e), weights
    [64x64 | 64x8 | 64x8] = 5120, factors sqrt(1/64), sqrt(3/64), sqrt(5/64); with the 1/sqrt(2l+1) of the Wigner
    symbol: out_l[w, m] = (1/8) sum_u x[u] W_l[u, w] Y_lm.
  - layer 2, (64x0e + 8x1o + 8x2e) (x) sh -> 64x0e: three instructions (0e.0e->0e, 1o.1o->0e, 2e.2e->0e), weights
    [64x64 | 8x64 | 8x64] = 5120, common factor sqrt(1/(64 + 8 + 8)):
    out[w] = (1/sqrt(80)) (sum_u s[u] Wa[u,w] + sum_u <v1[u], Y_1>/sqrt(3) Wb[u,w] + sum_u <v2[u], Y_2>/sqrt(5) Wc[u,w]).
* ``TensorProductConvLayer.forward``: ``edge_src, edge_dst = edge_index``; the product takes ``node_attr[edge_dst]`` and
  the result is scatter-MEANed over ``edge_src`` (atoms without outgoing edges get 0); the residual pads the input with
  zeros up to the output width.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .cartnet_ref import batch_norm, cholesky_head
from .icomformer_ref import _lin, comformer_conv, rbf_expansion

Tensor = torch.Tensor
NS, NV = 64, 8


def spherical_harmonics_12(vec: Tensor):
    """Y_1 [E,3] and Y_2 [E,5], component-normalised, of the normalised vectors (any orthonormal real basis)."""
    r = vec / vec.norm(dim=-1, keepdim=True)
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    y1 = math.sqrt(3.0) * r
    s3 = math.sqrt(3.0)
    y2 = math.sqrt(5.0) * torch.stack((s3 * x * y, s3 * y * z, 0.5 * (3 * z * z - 1), s3 * x * z,
                                       0.5 * s3 * (x * x - y * y)), dim=-1)
    return y1, y2


def _scatter_mean(v: Tensor, index: Tensor, n: int) -> Tensor:
    out = torch.zeros(n, v.shape[1], dtype=v.dtype).index_add_(0, index, v)
    cnt = torch.zeros(n, dtype=v.dtype).index_add_(0, index, torch.ones_like(index, dtype=v.dtype))
    return out / cnt.clamp(min=1).unsqueeze(-1)


def _edge_mlp(e: Tensor, sd, p: str) -> Tensor:
    """TensorProductConvLayer.fc (comformer_conv.py:209-213): Linear, Softplus, Linear -> per-edge weights."""
    return _lin(F.softplus(_lin(e, sd, p + ".fc.0")), sd, p + ".fc.2")


def tp_layer_1(sd, p: str, x0: Tensor, edge_index: Tensor, e: Tensor, y1: Tensor, y2: Tensor) -> Tensor:
    src, dst = edge_index[0], edge_index[1]
    w = _edge_mlp(e, sd, p)
    W0 = w[:, :NS * NS].reshape(-1, NS, NS)
    W1 = w[:, NS * NS:NS * NS + NS * NV].reshape(-1, NS, NV)
    W2 = w[:, NS * NS + NS * NV:].reshape(-1, NS, NV)
    xi = x0[dst]
    t0 = torch.einsum("eu,euw->ew", xi, W0) / 8.0
    t1 = torch.einsum("eu,euw->ew", xi, W1) / 8.0
    t2 = torch.einsum("eu,euw->ew", xi, W2) / 8.0
    out = torch.cat((t0, (t1.unsqueeze(-1) * y1.unsqueeze(1)).reshape(-1, 3 * NV),
                     (t2.unsqueeze(-1) * y2.unsqueeze(1)).reshape(-1, 5 * NV)), dim=-1)
    out = _scatter_mean(out, src, x0.shape[0])
    return out + F.pad(x0, (0, out.shape[1] - x0.shape[1]))


def tp_layer_2(sd, p: str, h1: Tensor, edge_index: Tensor, e: Tensor, y1: Tensor, y2: Tensor) -> Tensor:
    src, dst = edge_index[0], edge_index[1]
    w = _edge_mlp(e, sd, p).reshape(-1, NS + 2 * NV, NS)
    hi = h1[dst]
    s = hi[:, :NS]
    v1 = hi[:, NS:NS + 3 * NV].reshape(-1, NV, 3)
    v2 = hi[:, NS + 3 * NV:].reshape(-1, NV, 5)
    inp = torch.cat((s, (v1 * y1.unsqueeze(1)).sum(-1) / math.sqrt(3.0), (v2 * y2.unsqueeze(1)).sum(-1) / math.sqrt(5.0)),
                    dim=-1)
    out = torch.einsum("eu,euw->ew", inp, w) / math.sqrt(float(NS + 2 * NV))
    return _scatter_mean(out, src, h1.shape[0])


def comformer_conv_equi(sd, p: str, x: Tensor, edge_index: Tensor, e: Tensor, cart_dir: Tensor, training: bool,
                        new_stats) -> Tensor:
    """models/comformer_conv.py:266-279."""
    y1, y2 = spherical_harmonics_12(cart_dir)
    skip = x
    h = _lin(x, sd, p + ".node_linear")
    h = tp_layer_1(sd, p + ".nlayer_1", h, edge_index, e, y1, y2)
    h = tp_layer_2(sd, p + ".nlayer_2", h, edge_index, e, y1, y2)
    h = F.softplus(_lin(F.softplus(batch_norm(h, sd, p + ".bn", training, new_stats)), sd, p + ".node_linear_2"))
    return h + _lin(skip, sd, p + ".skip_linear")


def ecomformer_forward(sd: Dict[str, Tensor], batch, training: bool = False,
                       new_stats: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """models/comformer.py:56-70.  Does not mutate ``batch``."""
    C = sd["embedding.weight"].shape[1]
    dt = sd["embedding.weight"].dtype
    x = F.embedding(batch.x, sd["embedding.weight"]) + \
        _lin(batch.temperature.unsqueeze(-1), sd, "temperature_proj_atom")[batch.batch]
    edge_feat = -0.75 / batch.cart_dist
    e = F.softplus(_lin(rbf_expansion(edge_feat, -4.0, 0.0, C).to(dt), sd, "rbf.1"))
    x = comformer_conv(sd, "att_layers.0", x, batch.edge_index, e, training, new_stats)
    x = comformer_conv_equi(sd, "equi_update", x, batch.edge_index, e, batch.cart_dir, training, new_stats)
    x = comformer_conv(sd, "att_layers.1", x, batch.edge_index, e, training, new_stats)
    x = comformer_conv(sd, "att_layers.2", x, batch.edge_index, e, training, new_stats)
    head = {"head.MLP.0.weight": sd["cholesky.MLP.0.weight"], "head.MLP.0.bias": sd["cholesky.MLP.0.bias"],
            "head.MLP.2.weight": sd["cholesky.MLP.2.weight"], "head.MLP.2.bias": sd["cholesky.MLP.2.bias"]}
    return cholesky_head(head, x, batch.non_H_mask)
