"""The ``train_epoch`` fixture (tests/golden/train_epoch.npz, made by tests/golden/make_golden.py::train_epoch_fixture from
the reference's own train/train.py:148-199 run in the build container): loader + helpers shared by the CPU test that
replays it with the oracle and the GPU test that replays it with cartnet_amd.train.train_epoch."""
import os

import numpy as np
import torch

from cartnet_amd.data import Batch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_MICRO = 5


def load():
    z = np.load(os.path.join(GOLDEN, "train_epoch.npz"))
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp_")}
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_")}
    micro = []
    for i in range(N_MICRO):
        b = Batch()
        pre = f"b{i}_in_"
        for k in z.files:
            if k.startswith(pre) and k != pre + "num_graphs":
                setattr(b, k[len(pre):], torch.from_numpy(z[k]))
        b.num_graphs = int(z[pre + "num_graphs"])
        micro.append(b)
    names = [str(n) for n in z["param_names"]]
    sizes = [int(s) for s in z["param_sizes"]]
    return z, hp, sd, micro, names, sizes


def unflatten(flat, names, sizes, like):
    """name -> tensor views of a flat parameter / gradient vector in named_parameters() order."""
    out, off = {}, 0
    for n, k in zip(names, sizes):
        out[n] = torch.as_tensor(flat[off:off + k]).reshape(like[n].shape)
        off += k
    return out


def boundaries(n_iter, accum):
    """Iterations (0-based) after which the reference steps the optimiser: every ``accum``-th and the last one
    (train/train.py:186)."""
    return [it for it in range(n_iter) if (it + 1) % accum == 0 or it + 1 == n_iter]


def well_conditioned(z, n_steps, frac=1e-2):
    """Elements whose reference gradient is >= ``frac`` of the largest one at EVERY optimiser step.  Adam divides by
    sqrt(v): where the true gradient is ~0 (the bias in front of a training-mode BatchNorm has an exactly zero gradient
    and receives pure rounding noise of ~1e-10) the update is +-lr times the SIGN of that noise, in the reference as much
    as anywhere else -- no implementation can be compared there."""
    mask = None
    for k in range(n_steps):
        g = np.abs(z[f"step{k}_grad"])
        m = g >= frac * g.max()
        mask = m if mask is None else (mask & m)
    return mask
