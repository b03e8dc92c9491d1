"""Load the committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from cartnet_amd.data import Batch
from cartnet_amd.model import make_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_FIXTURES = ["tiny_adp", "tiny_scalar", "tiny_invariant", "tiny_noatom", "tiny_nothing", "config1", "config2"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp_")}
    b = Batch()
    for k in z.files:
        if k.startswith("in_") and k != "in_num_graphs":
            setattr(b, k[3:], torch.from_numpy(z[k]))
    b.num_graphs = int(z["in_num_graphs"])
    if any(k.startswith("w_") for k in z.files):
        sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_")}
    else:
        sd = make_state_dict(hp["dim_in"], hp["dim_rbf"], hp["num_layers"], seed=int(z["weights_seed"]),
                             cholesky=hp["cholesky"], temperature=hp["temperature"], atom_types=hp["atom_types"],
                             invariant=hp["invariant"], radius=hp["radius"])
    abs_sum = sum(v.double().abs().sum().item() for v in sd.values())
    assert abs(abs_sum - float(z["weights_abs_sum"])) <= 1e-9 * abs(float(z["weights_abs_sum"])), \
        "regenerated weights differ from the ones the fixture was made with"
    return z, hp, b, sd


def clone_batch(b):
    c = b.clone()
    c.num_graphs = b.num_graphs
    return c


def oracle_kwargs(hp):
    return dict(num_layers=hp["num_layers"], radius=hp["radius"], invariant=hp["invariant"],
                use_temperature=hp["temperature"], use_envelope=hp["use_envelope"], atom_types=hp["atom_types"],
                cholesky=hp["cholesky"])
