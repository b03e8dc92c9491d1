"""Loader for the iComformer golden vectors (tests/golden/icomformer_*.npz)."""
import os

import numpy as np
import torch

from cartnet_amd.comformer import make_icomformer_state_dict
from cartnet_amd.data import Batch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["icomformer_tiny", "icomformer_c32"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    b = Batch()
    for k in z.files:
        if k.startswith("in_") and k != "in_num_graphs":
            setattr(b, k[3:], torch.from_numpy(z[k]))
    b.num_graphs = int(z["in_num_graphs"])
    if any(k.startswith("w_") for k in z.files):
        sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_")}
    else:
        sd = make_icomformer_state_dict(int(z["hp_dim_in"]), seed=int(z["weights_seed"]))
    abs_sum = sum(v.double().abs().sum().item() for v in sd.values())
    assert abs(abs_sum - float(z["weights_abs_sum"])) <= 1e-9 * abs(float(z["weights_abs_sum"]))
    return z, b, sd


def to64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def batch64(b):
    c = b.clone()
    c.num_graphs = b.num_graphs
    for k, v in list(c.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(c, k, v.double())
    return c
