"""BatchNorm groups (include/cartnet_hip.h: CartnetGroups; model.bn_group_size): the reference recipe's micro-batches
(batch 4 x accumulation 16: scripts/train_cartnet_adp.sh:4, train/train.py:183-189) carried through the network as ONE
batch.  Must equal, within the usual parity budget, what the reference does -- one forward / backward per micro-batch
with BatchNorm statistics over that micro-batch only, running statistics updated after every micro-batch, gradients of
the per-micro-batch mean losses accumulated unscaled -- here replayed by the fp64 oracle micro-batch by micro-batch."""
import pytest
import torch

import golden_utils as gu
from conftest import rel_err
from test_gpu_model import PRED_TOL, _check_grads, _model

pytestmark = pytest.mark.gpu


def _f64(b):
    c = gu.clone_batch(b)
    for k, v in list(c.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(c, k, v.double())
    return c


def _oracle_micro_batches(sd, items, group_size, hp):
    """Sequential micro-batches through the oracle: returns (pred rows in batch order, grads of the summed losses, final
    BatchNorm buffers)."""
    from cartnet_amd.data import Batch
    from oracle import cartnet_ref as orc
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "rbf" not in k]
    sd64 = {k: (v.double().requires_grad_(k in names) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    preds, total = [], 0.0
    for s in range(0, len(items), group_size):
        mb = _f64(Batch.from_data_list(items[s:s + group_size]))
        new_stats = {}
        pred = orc.cartnet_forward(sd64, mb, training=True, new_stats=new_stats, **gu.oracle_kwargs(hp))
        total = total + (pred - mb.y).abs().mean()
        preds.append(pred.detach())
        for k, v in new_stats.items():                      # the next micro-batch starts from the updated buffers
            sd64[k] = v
    total.backward()
    return torch.cat(preds), {k: sd64[k].grad for k in names}, {k: v for k, v in sd64.items() if "running" in k or
                                                                "num_batches" in k}


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("case", ["d32_groups_of_2", "d256_groups_of_4", "d64_ragged_last_group", "scalar_head"])
def test_grouped_pass_equals_sequential_micro_batches_of_the_oracle(case, precision):
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    from cartnet_amd.train import grouped_loss
    D, L, gsz, sizes, chol = {"d32_groups_of_2": (32, 2, 2, (7, 12, 5, 9, 16, 3), True),
                              "d256_groups_of_4": (256, 2, 4, (20, 31, 12, 25, 18, 40, 9, 22), True),
                              "d64_ragged_last_group": (64, 3, 2, (6, 11, 8, 14, 10), True),
                              "scalar_head": (32, 2, 3, (4, 9, 6, 12, 5, 7), False)}[case]
    hp = dict(dim_in=D, dim_rbf=16, num_layers=L, radius=5.0, invariant=False, temperature=chol, use_envelope=True,
              atom_types=True, cholesky=chol)
    items = [make_crystal(9000 + i, n, adp=chol) for i, n in enumerate(sizes)]
    sd = make_state_dict(D, 16, L, seed=31, cholesky=chol, temperature=chol)
    m = _model(hp, sd, precision).train()
    m.bn_group_size = gsz
    b = Batch.from_data_list(items).to("cuda:0")
    pred, true = m(b)
    mae, _, G = grouped_loss(pred, true, b, gsz)
    assert G == -(-len(items) // gsz)
    mae.backward()
    ref_pred, ref_grads, ref_state = _oracle_micro_batches(sd, items, gsz, hp)
    assert rel_err(pred, ref_pred) < PRED_TOL
    _check_grads({k: p.grad for k, p in m.named_parameters()}, ref_grads, case)
    got = m.state_dict()
    for k, v in ref_state.items():
        if v.is_floating_point():
            assert rel_err(got[k], v) < 1e-5, k
        else:
            assert int(got[k]) == int(sd[k]) + G, k           # num_batches_tracked advanced once per micro-batch


def test_grouped_pass_equals_separate_hip_passes_and_eval_ignores_groups():
    """The same micro-batches as separate forward / backward calls of the HIP path (gradient accumulation in the
    optimiser's flat buffer): identical predictions, gradients and BatchNorm buffers up to fp32 summation order; in
    eval mode BatchNorm uses the running statistics, so groups change nothing at all."""
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    from cartnet_amd.train import grouped_loss
    hp = dict(dim_in=256, dim_rbf=64, num_layers=4, radius=5.0, invariant=False, temperature=True, use_envelope=True,
              atom_types=True, cholesky=True)
    items = [make_crystal(9100 + i, n) for i, n in enumerate((64, 90, 75, 120, 66, 81, 70, 101))]
    sd = make_state_dict(256, 64, 4, seed=32)
    ma, mb = _model(hp, sd).train(), _model(hp, sd).train()
    ma.bn_group_size = 4
    b = Batch.from_data_list(items).to("cuda:0")
    pa, ta = ma(b)
    la, _, _ = grouped_loss(pa, ta, b, 4)
    la.backward()
    preds = []
    for s in (0, 4):
        bb = Batch.from_data_list(items[s:s + 4]).to("cuda:0")
        p, t = mb(bb)
        (p - t).abs().mean().backward()                       # autograd accumulates into .grad
        preds.append(p.detach())
    assert rel_err(pa, torch.cat(preds)) < 2e-6
    ga = torch.cat([p.grad.flatten() for p in ma.parameters()])
    gb = torch.cat([p.grad.flatten() for p in mb.parameters()])
    assert rel_err(ga, gb) < 2e-5
    for (k, va), vb in zip(ma.state_dict().items(), mb.state_dict().values()):
        if "running" in k:
            assert rel_err(va, vb) < 1e-6, k
        elif "num_batches" in k:
            assert int(va) == int(vb) == int(sd[k]) + 2
    ma.eval(); mb.eval()
    with torch.no_grad():
        ea, _ = ma(Batch.from_data_list(items).to("cuda:0"))
        mb.load_state_dict(ma.state_dict())
        eb, _ = mb(Batch.from_data_list(items).to("cuda:0"))
    assert torch.equal(ea, eb)


def test_train_epoch_with_groups_matches_the_micro_batch_recipe():
    """train_epoch(batch 8, accumulation 1, bn_group_size 2) == train_epoch(batch 2, accumulation 4) on the same
    crystals in the same order: same parameters after the epoch (Adam is fed the same summed gradient)."""
    from cartnet_amd.data import DataLoader
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.optim import FlatAdam
    from cartnet_amd.synthetic import make_crystal
    from cartnet_amd.train import train_epoch
    hp = dict(dim_in=64, dim_rbf=16, num_layers=2, radius=5.0, invariant=False, temperature=True, use_envelope=True,
              atom_types=True, cholesky=True)
    items = [make_crystal(9200 + i, 10 + (3 * i) % 17) for i in range(16)]
    sd = make_state_dict(64, 16, 2, seed=33)
    ma, mb = _model(hp, sd).train(), _model(hp, sd).train()
    ma.validate_graph = mb.validate_graph = False
    ma.bn_group_size = 2
    oa, ob = FlatAdam(ma, lr=1e-3), FlatAdam(mb, lr=1e-3)
    ra = train_epoch(DataLoader(items, 8), ma, oa, 1)
    rb = train_epoch(DataLoader(items, 2), mb, ob, 4)
    assert ra["graphs"] == rb["graphs"] == 16 and abs(ra["mae"] - rb["mae"]) < 1e-5 * abs(rb["mae"])
    # Adam normalises the step: compare where the gradient is not rounding noise (cf. test_gpu_robustness)
    da, db = oa.flat_param - torch.cat([v.flatten() for k, v in sd.items() if k in dict(ma.named_parameters())]).cuda(), None
    assert oa.step_count == ob.step_count == 2
    assert rel_err(oa.exp_avg, ob.exp_avg) < 1e-4 and rel_err(oa.exp_avg_sq, ob.exp_avg_sq) < 1e-4
