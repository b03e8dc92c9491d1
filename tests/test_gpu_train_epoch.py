"""cartnet_amd.train.train_epoch + FlatAdam + the one-cycle schedule against the reference's OWN training loop
(SURVEY.md 8(a)13).  tests/golden/train_epoch.npz holds what train/train.py:148-199 produced in the build container for
two epochs of five micro-batches with accumulation 3 (optimiser steps after micro-batches 3 and 5 -- the last-iteration
flush), torch Adam (main.py:208) and OneCycleLR (train/train.py:59, which also cycles Adam's beta1).  The GPU path runs
the same two epochs FREE-RUNNING -- its own parameters carry from step to step, nothing is re-seeded from the fixture."""
import numpy as np
import pytest
import torch

import train_epoch_utils as tu

pytestmark = pytest.mark.gpu


def test_two_epochs_against_the_reference_train_epoch():
    from cartnet_amd.config import cfg
    from cartnet_amd.model import CartNet
    from cartnet_amd.optim import FlatAdam, one_cycle_lr, one_cycle_momentum
    from cartnet_amd.train import train_epoch
    z, hp, sd, micro, names, sizes = tu.load()
    epochs, accum, lr_max, warm = int(z["epochs"]), int(z["accum"]), float(z["lr"]), float(z["warmup"])
    total = epochs * len(micro) // accum + epochs                                   # train/train.py:59
    cfg.radius, cfg.loss = hp["radius"], "MAE"
    m = CartNet(hp["dim_in"], hp["dim_rbf"], hp["num_layers"])
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    assert [n for n, _ in m.named_parameters()] == names

    seen = []

    class Recording(FlatAdam):
        def step(self, grad_scale=1.0):
            seen.append({"grad": self.flat_grad.detach().cpu().numpy().copy(), "lr": self.lr, "beta1": self.betas[0]})
            super().step(grad_scale)
            seen[-1]["param"] = self.flat_param.detach().cpu().numpy().copy()

    opt = Recording(m, lr=lr_max)
    k = [0]

    def scheduler():              # what main.py hands to train_epoch
        k[0] += 1
        opt.set_lr(one_cycle_lr(min(k[0], total - 1), total, lr_max, warm))
        opt.set_beta1(one_cycle_momentum(min(k[0], total - 1), total, warm))
    opt.set_lr(one_cycle_lr(0, total, lr_max, warm))
    opt.set_beta1(one_cycle_momentum(0, total, warm))

    def loader():
        out = []
        for b in micro:
            c = b.clone()
            c.num_graphs = b.num_graphs
            out.append(c)
        return out

    mask = tu.well_conditioned(z, 4)
    assert mask.sum() >= 500                                    # (most embedding rows belong to absent elements: zero gradient)
    for ep in range(epochs):
        stats = train_epoch(loader(), m, opt, accum, scheduler)
        want = float(np.mean(z["iter_mae"][ep * len(micro):(ep + 1) * len(micro)]))
        assert abs(stats["mae"] - want) <= 2e-5 * want, (ep, stats["mae"], want)
        assert stats["graphs"] == 2 * len(micro)
        st = m.state_dict()
        for key in z.files:
            pre = f"state_ep{ep}_"
            if key.startswith(pre):
                got, ref = st[key[len(pre):]].cpu(), torch.from_numpy(z[key])
                if ref.is_floating_point():
                    # The gate BatchNorm's running mean contains MLP_gate.2.bias, the one parameter whose true gradient is
                    # zero (a constant in front of a training-mode BatchNorm): Adam moves it by +-lr per step along the
                    # sign of rounding noise, in the reference as here, so that mean may differ by the learning rates spent
                    drift_ok = sum(r["lr"] for r in seen) if key.endswith(".norm.running_mean") else 0.0
                    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-6 + drift_ok), key
                else:
                    assert int(got) == int(ref), key
    assert len(seen) == 4                                       # 2 epochs x (micro-batches 3 and 5)
    drift = 0.0
    for s, rec in enumerate(seen):
        # the optimiser step used the learning rate / beta1 the reference's scheduler had set
        assert abs(rec["lr"] - one_cycle_lr(s, total, lr_max, warm)) < 1e-15
        assert abs(rec["beta1"] - one_cycle_momentum(s, total, warm)) < 1e-15
        g_ref, p_ref = z[f"step{s}_grad"], z[f"step{s}_param"]
        gmax = np.abs(g_ref).max()
        dd = np.abs(rec["param"] - p_ref)
        print(f"step {s}: lr {rec['lr']:.3e} beta1 {rec['beta1']:.4f} grad err {np.abs(rec['grad'] - g_ref).max() / gmax:.2e} "
              f"param err well-conditioned {dd[mask].max() / rec['lr']:.2e} lr, all {dd.max() / rec['lr']:.2e} lr")
        # accumulated gradient of the window: unscaled sum over its micro-batches, the model tests' bound (measured:
        # 4e-6 .. 1.4e-5 over the four free-running windows)
        assert np.abs(rec["grad"] - g_ref).max() <= 3e-5 * gmax, s
        # parameters: where the gradient is well above rounding noise the update is determined to O(relative gradient
        # error) of the learning rate (measured: 0, 3e-5, 5e-5, 2e-4 of the step's learning rate)
        drift += 2e-3 * rec["lr"]
        d = np.abs(rec["param"] - p_ref)
        assert d[mask].max() <= drift + 2e-7 * np.abs(p_ref).max(), (s, d[mask].max(), rec["lr"])
        # everywhere (noise-driven elements included) a step moves a parameter by about lr at most
        assert d.max() <= 2.5 * sum(r["lr"] for r in seen[:s + 1]), s
    # the learning rate after the last scheduler step is what the reference logged last
    assert abs(opt.lr - float(z["iter_lr"][-1])) <= 1e-12
