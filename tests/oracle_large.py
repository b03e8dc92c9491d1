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
