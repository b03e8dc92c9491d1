"""Pin the training loop to the reference's own ``train_epoch`` (train/train.py:148-199).

tests/golden/train_epoch.npz was produced by running that function itself (tests/golden/make_golden.py::
train_epoch_fixture) for two epochs of five micro-batches, accumulation 3 -- optimiser steps after micro-batches 3 and 5,
the second one being the last-iteration flush of two micro-batches -- with torch's Adam (main.py:208) and OneCycleLR
(train/train.py:59).  Here the oracle's restatement of the loop (oracle/train_ref.py) replays it on the CPU: losses,
accumulated gradients, Adam updates, learning rates, BatchNorm buffers."""
import numpy as np
import torch

import train_epoch_utils as tu
from oracle import train_ref as otr
from cartnet_amd.optim import one_cycle_lr, one_cycle_momentum


def _setup():
    z, hp, sd, micro, names, sizes = tu.load()
    epochs, accum = int(z["epochs"]), int(z["accum"])
    total = epochs * len(micro) // accum + epochs                                   # train/train.py:59
    kw = dict(num_layers=hp["num_layers"], radius=hp["radius"], invariant=hp["invariant"],
              use_temperature=hp["temperature"], use_envelope=hp["use_envelope"], atom_types=hp["atom_types"],
              cholesky=hp["cholesky"])
    return z, sd, micro, names, sizes, epochs, accum, total, kw


def test_learning_rate_follows_the_reference_schedule():
    z, sd, micro, names, sizes, epochs, accum, total, kw = _setup()
    lr_max, warm = float(z["lr"]), float(z["warmup"])
    done, want = 0, []
    for ep in range(epochs):
        for it in range(len(micro)):
            if otr.is_boundary(it, len(micro), accum):
                done += 1
            want.append(otr.one_cycle_lr(done, total, lr_max, warm))               # logged AFTER scheduler.step() (:188,:195)
    assert tu.boundaries(len(micro), accum) == [2, 4]                               # steps after micro-batches 3 and 5 (flush)
    np.testing.assert_allclose(z["iter_lr"], want, rtol=1e-12, atol=0)
    # the product's own schedule function (what main.py hands to train_epoch) gives the same numbers
    got = [one_cycle_lr(k, total, lr_max, warm) for k in range(total)]
    np.testing.assert_allclose(got, [otr.one_cycle_lr(k, total, lr_max, warm) for k in range(total)], rtol=1e-15)
    np.testing.assert_allclose([one_cycle_momentum(k, total, warm) for k in range(total)],
                               [otr.one_cycle_momentum(k, total, warm) for k in range(total)], rtol=1e-15)


def test_oracle_loop_reproduces_the_reference_gradients_losses_and_buffers():
    """Teacher-forced by window: every accumulation window starts from the parameters the REFERENCE held (so one
    window's rounding cannot leak into the next through Adam's sign-like first steps); fp64 oracle against the fp32 run."""
    z, sd, micro, names, sizes, epochs, accum, total, kw = _setup()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for n in names:
        sd64[n].requires_grad_(True)
    micro64 = []
    for b in micro:
        c = b.clone()
        c.num_graphs = b.num_graphs
        for key, v in list(c.__dict__.items()):
            if torch.is_tensor(v) and v.is_floating_point():
                setattr(c, key, v.double())
        micro64.append(c)
    micro = micro64
    k = [0]
    rows = []

    def on_boundary(grads):
        g = torch.cat([grads[n].reshape(-1) for n in names]).numpy()
        ref = z[f"step{k[0]}_grad"].astype(np.float64)
        assert np.abs(g - ref).max() <= 3e-5 * np.abs(ref).max(), k[0]
        # continue from the reference's parameters after this step
        new = tu.unflatten(z[f"step{k[0]}_param"].astype(np.float64), names, sizes, sd64)
        with torch.no_grad():
            for n in names:
                sd64[n].copy_(new[n])
        k[0] += 1

    for ep in range(epochs):
        rows += otr.train_epoch(sd64, names, micro, accum, on_boundary, **kw)
        for key in z.files:
            pre = f"state_ep{ep}_"
            if key.startswith(pre):
                got, ref = sd64[key[len(pre):]], torch.from_numpy(z[key])
                if ref.is_floating_point():
                    assert torch.allclose(got, ref.double(), rtol=2e-5, atol=1e-7), key
                else:
                    assert int(got) == int(ref) == (ep + 1) * len(micro), key
    assert k[0] == 4
    rows = np.array(rows)
    np.testing.assert_allclose(rows[:, 0], z["iter_mae"], rtol=1e-5)
    np.testing.assert_allclose(rows[:, 1], z["iter_mse"], rtol=1e-5)
    np.testing.assert_allclose(z["iter_loss"], z["iter_mae"], rtol=0)              # cfg.loss = "MAE"


def test_adam_restatement_reproduces_the_reference_parameters():
    """The reference's own accumulated gradients through oracle.train_ref.adam_step with the schedule's learning rates
    and beta1 values: the parameters after each of the four optimiser steps, to fp32 rounding of the update.  (With a
    constant beta1 = 0.9 -- what this build's loop did until round 5 -- step 1 is off by 0.45 lr.)"""
    z, sd, micro, names, sizes, epochs, accum, total, kw = _setup()
    lr_max, warm = float(z["lr"]), float(z["warmup"])
    p = torch.cat([sd[n].reshape(-1) for n in names]).double()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for k in range(4):
        lr = otr.one_cycle_lr(k, total, lr_max, warm)           # optimiser step k runs BEFORE scheduler step k + 1
        b1 = otr.one_cycle_momentum(k, total, warm)             # OneCycleLR cycles Adam's beta1 too (0.95 -> 0.85 -> 0.95)
        otr.adam_step(p, torch.from_numpy(z[f"step{k}_grad"]).double(), m, v, k + 1, lr, betas=(b1, 0.999))
        ref = torch.from_numpy(z[f"step{k}_param"]).double()
        assert (p - ref).abs().max().item() <= 2e-6 * lr + 1.2e-7 * ref.abs().max().item(), k
        p = ref.clone()                                         # fp32-rounded, as the reference continues
