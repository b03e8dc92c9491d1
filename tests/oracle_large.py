"""The fp64 oracle at the FULL benchmark batch (BASELINE configs[1]: 64 crystals x 194 atoms, E ~ 177k edges).

oracle.cartnet_ref.cartnet_forward keeps every intermediate of every layer for autograd: at this size that is ~25 GB of
fp64 ([E, 3D] concatenations of 1.1 GB each, a dozen [E, D] tensors per layer).  The same functions -- encoder,
cartnet_layer, cholesky_head, unchanged -- are composed here with torch.utils.checkpoint around each layer, so that only
the layer boundaries (x [N, D], e [E, D]) stay alive and a layer's interior is recomputed in its backward: ~8 GB, one
extra forward.  Test infrastructure, like the oracle itself."""
import time

import torch
from torch.utils.checkpoint import checkpoint

from oracle import cartnet_ref as orc


def train_step_fp64(sd, batch, num_layers, radius=5.0, threads=None):
    """One training-mode forward + MAE + backward of the oracle in fp64 on the host.  ``sd``: fp32 reference-layout
    state_dict (CPU).  Returns dict(pred, mae, grads {name: fp64}, new_stats {name: tensor}, seconds)."""
    if threads:
        torch.set_num_threads(int(threads))
    t0 = time.perf_counter()
    params = {k for k in sd if not ("running_" in k or "num_batches" in k or k.startswith("encoder.rbf."))}
    sd64 = {k: (v.double().clone().requires_grad_(k in params) if v.is_floating_point() else v.clone())
            for k, v in sd.items()}
    b = batch.clone()
    for k, v in list(b.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b, k, v.double())
    new_stats = {}
    x, e = orc.encoder(sd64, b.x, b.temperature, b.batch, b.cart_dist, b.cart_dir, radius)
    for l in range(num_layers):
        def layer(x_, e_, l=l):
            return orc.cartnet_layer(sd64, l, x_, e_, b.edge_index, b.cart_dist, radius, True, True, new_stats)
        x, e = checkpoint(layer, x, e, use_reentrant=False)
    pred = orc.cholesky_head(sd64, x, b.non_H_mask)
    mae, _ = orc.compute_loss(pred, b.y)
    mae.backward()
    grads = {k: sd64[k].grad for k in sd64 if torch.is_tensor(sd64[k]) and sd64[k].requires_grad}
    return {"pred": pred.detach(), "mae": float(mae.detach()), "grads": grads,
            "new_stats": {k: v.detach() for k, v in new_stats.items()}, "seconds": time.perf_counter() - t0}


def icomformer_train_step_fp64(sd, batch, threads=None):
    """The same for the iComformer oracle (oracle/icomformer_ref.py: models/comformer.py:115-132): a checkpoint around each
    of the four ComformerConv layers and around the edge-update layer, whose interior (a dozen [E, 3, C] tensors, 1.1 GB
    each in fp64 at the benchmark batch) is by far the largest.  Returns dict(pred, mae, grads, new_stats, seconds)."""
    import torch.nn.functional as F
    from oracle import icomformer_ref as icf
    if threads:
        torch.set_num_threads(int(threads))
    t0 = time.perf_counter()
    params = {k for k, v in sd.items() if v.is_floating_point() and not ("running_" in k)}
    sd64 = {k: (v.double().clone().requires_grad_(k in params) if v.is_floating_point() else v.clone())
            for k, v in sd.items()}
    b = batch.clone()
    for k, v in list(b.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b, k, v.double())
    new_stats = {}
    C = sd64["embedding.weight"].shape[1]
    src = b.edge_index[0]
    x = F.embedding(b.x, sd64["embedding.weight"]) + \
        icf._lin(b.temperature.unsqueeze(-1), sd64, "temperature_proj_atom")[b.batch]
    E = b.cart_dist.shape[0]
    nl = (-0.75 / torch.norm(b.cell, dim=-1))[b.batch[src]]
    na = icf.bond_cosine(b.cell[b.batch[src]], b.cart_dir.unsqueeze(1).repeat(1, 3, 1))

    def rbf(v):
        return F.softplus(icf._lin(icf.rbf_expansion(v, -4.0, 0.0, C).double(), sd64, "rbf.1"))

    def rbf_a(v):
        return F.softplus(icf._lin(icf.rbf_expansion(v, -1.0, 1.0, C).double(), sd64, "rbf_angle.1"))
    e = checkpoint(rbf, -0.75 / b.cart_dist, use_reentrant=False)
    nei_len = checkpoint(rbf, nl.reshape(-1), use_reentrant=False).reshape(E, 3, -1)
    nei_angle = checkpoint(rbf_a, na.reshape(-1), use_reentrant=False).reshape(E, 3, -1)

    def conv(l):
        return lambda x_, e_: icf.comformer_conv(sd64, f"att_layers.{l}", x_, b.edge_index, e_, True, new_stats)
    x = checkpoint(conv(0), x, e, use_reentrant=False)
    e = checkpoint(lambda e_, a_, c_: icf.comformer_conv_edge(sd64, "edge_update_layer", e_, a_, c_, True, new_stats),
                   e, nei_len, nei_angle, use_reentrant=False)
    for l in (1, 2, 3):
        x = checkpoint(conv(l), x, e, use_reentrant=False)
    head = {"head.MLP.0.weight": sd64["cholesky.MLP.0.weight"], "head.MLP.0.bias": sd64["cholesky.MLP.0.bias"],
            "head.MLP.2.weight": sd64["cholesky.MLP.2.weight"], "head.MLP.2.bias": sd64["cholesky.MLP.2.bias"]}
    pred = orc.cholesky_head(head, x, b.non_H_mask)
    mae = (pred - b.y).abs().mean()
    mae.backward()
    grads = {k: v.grad for k, v in sd64.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    return {"pred": pred.detach(), "mae": float(mae.detach()), "grads": grads,
            "new_stats": {k: v.detach() for k, v in new_stats.items()}, "seconds": time.perf_counter() - t0}
