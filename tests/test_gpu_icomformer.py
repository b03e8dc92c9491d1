"""iComformer (BASELINE.json configs[4]) on the GPU against golden vectors generated from the reference's own
models/comformer.py.  Same tolerances as the CartNet path (tests/test_gpu_model.py)."""
import numpy as np
import pytest
import torch

import icomformer_utils as iu
from conftest import rel_err
from test_gpu_model import PRED_TOL, _check_grads

pytestmark = pytest.mark.gpu


def _model(z, sd):
    from cartnet_amd.comformer import iComformer
    m = iComformer(int(z["hp_dim_in"]))
    m.load_state_dict(sd, strict=True)
    m.validate_graph = True
    return m.to("cuda:0")


def _clone(b):
    c = b.clone()
    c.num_graphs = b.num_graphs
    return c


@pytest.mark.parametrize("name", iu.FIXTURES)
def test_forward_against_reference_golden(name):
    z, b, sd = iu.load(name)
    m = _model(z, sd)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        bb = _clone(b).to("cuda:0")
        with torch.no_grad():
            pred, true = m(bb)
        assert true is bb.y
        assert rel_err(pred, torch.from_numpy(z[f"{mode}_pred_f64"])) < PRED_TOL, (name, mode)
        if mode == "train":
            assert rel_err(bb.x, torch.from_numpy(z["train_x_final_f64"])) < PRED_TOL


@pytest.mark.parametrize("name", iu.FIXTURES)
def test_train_step_gradients_and_bn_state_against_reference_golden(name):
    z, b, sd = iu.load(name)
    m = _model(z, sd).train()
    bb = _clone(b).to("cuda:0")
    pred, true = m(bb)
    loss = (pred - true).abs().mean()
    assert abs(loss.item() - float(z["train_mae"])) < 1e-5 * abs(float(z["train_mae"]))
    loss.backward()
    params = dict(m.named_parameters())
    unused = set(z["unused_params"].tolist())
    got, ref = {}, {}
    for k, p in params.items():
        if k in unused:
            assert p.grad is None, k        # the reference leaves these without a gradient too
            continue
        assert p.grad is not None, f"no gradient for {k}"
        got[k], ref[k] = p.grad, torch.from_numpy(z["grad64_" + k])
    _check_grads(got, ref, name)
    sd_new = m.state_dict()
    for k in z.files:
        if k.startswith("state_"):
            r = torch.from_numpy(z[k])
            if r.is_floating_point():
                assert rel_err(sd_new[k[6:]], r) < 1e-5, k
            else:
                assert int(sd_new[k[6:]]) == int(r), k


@pytest.mark.parametrize("precision", [0, 1])
def test_against_oracle_at_width_256(precision):
    """configs[4] width (C=256) on two small crystals: HIP path vs the fp64 oracle (forward and gradients), with the
    fp32-MFMA and the bf16x3 GEMMs (same budget)."""
    from cartnet_amd.comformer import iComformer, make_icomformer_state_dict
    from cartnet_amd.data import Batch
    from cartnet_amd.synthetic import make_crystal
    from oracle import icomformer_ref as orc
    b = Batch.from_data_list([make_crystal(950, 12), make_crystal(951, 20)])
    sd = make_icomformer_state_dict(256, seed=7)
    m = iComformer(256)
    m.load_state_dict(sd)
    m.gemm_precision = precision
    m = m.to("cuda:0").train()
    bb = _clone(b).to("cuda:0")
    pred, true = m(bb)
    (pred - true).abs().mean().backward()
    names = [k for k, p in m.named_parameters() if p.grad is not None]
    sd64 = {k: (v.double().requires_grad_(k in names) if v.is_floating_point() else v) for k, v in sd.items()}
    ref = orc.icomformer_forward(sd64, iu.batch64(b), training=True)
    assert rel_err(pred, ref) < PRED_TOL
    (ref - b.y.double()).abs().mean().backward()
    _check_grads({k: p.grad for k, p in m.named_parameters() if p.grad is not None},
                 {k: sd64[k].grad for k in names}, "icomformer256")


@pytest.mark.parametrize("precision", [0, 1])
def test_properties_at_the_adp_shape(precision):
    """BASELINE.json configs[4] at its real size -- 64 crystals x 194 atoms, E ~ 177k (3E ~ 531k rows in the
    edge-update layer), C = 256 -- which the fp64 oracle cannot check element by element in seconds.  Size-independent
    properties instead: (i) eval mode uses running BatchNorm statistics, so a crystal's prediction must not depend on
    which crystals share its batch; (ii) two training steps from the same state agree bit for bit (no atomics);
    (iii) every gradient the reference produces is finite and non-trivial."""
    from cartnet_amd.comformer import iComformer, make_icomformer_state_dict
    from cartnet_amd.data import Batch
    from cartnet_amd.synthetic import make_crystal
    items = [make_crystal(8000 + g, 194) for g in range(64)]
    sd = make_icomformer_state_dict(256, seed=9)
    m = iComformer(256)
    m.load_state_dict(sd)
    m.gemm_precision = precision
    m = m.to("cuda:0")

    def batch(idx):
        return Batch.from_data_list([items[i] for i in idx]).to("cuda:0")

    m.eval()
    with torch.no_grad():
        full, _ = m(batch(range(64)))
        parts = torch.cat([m(batch(range(s, s + 16)))[0] for s in range(0, 64, 16)])
        odd, _ = m(batch([63, 5, 17]))
    assert torch.isfinite(full).all() and rel_err(full, parts) < 2e-6
    per = torch.split(full, [int(it.non_H_mask.sum()) for it in items])
    assert rel_err(odd, torch.cat([per[63], per[5], per[17]])) < 2e-6

    outs = []
    for _ in range(2):
        m.load_state_dict(sd)
        m.train()
        m.zero_grad(set_to_none=True)
        pred, true = m(batch(range(64)))
        (pred - true).abs().mean().backward()
        outs.append((pred.detach().clone(),
                     {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) >= 40
    for k, g in outs[0][1].items():
        assert torch.isfinite(g).all(), k
        assert torch.equal(g, outs[1][1][k]), k
    assert sum(float(g.abs().sum()) for g in outs[0][1].values()) > 0


@pytest.mark.parametrize("train_mode", [True, False])
def test_native_sequence_equals_the_python_sequence(train_mode):
    """cartnet_icomformer_forward / _backward (csrc/icomformer.hip: one C-ABI call per direction) against the same kernels
    sequenced launch by launch from Python (`native_sequence = False`, the path eComformer still takes): predictions,
    every gradient and the BatchNorm buffers, at a golden width (no weight images) and at C = 256 (DMA-fed kernels).
    The two sequences differ in more than order since round 5: the C++ one never writes alpha, takes the gate's
    BatchNorm-backward sums from per-segment sums (in eval mode too) and the bias gradients from their producers.
    train_mode False: gradients through running-statistics BatchNorm (fine-tuning with frozen statistics)."""
    from cartnet_amd.comformer import iComformer, make_icomformer_state_dict
    from cartnet_amd.data import Batch
    from cartnet_amd.synthetic import make_crystal
    b = Batch.from_data_list([make_crystal(960 + i, 10 + 7 * i) for i in range(3)])
    # (320 > 256 columns: the C++ sequence takes its stored-alpha form there -- the one-chunk kernels of the alpha-free form
    #  need C <= 256)
    for width in (32, 256) + ((320,) if train_mode else ()):
        sd = make_icomformer_state_dict(width, seed=11)
        res = []
        for native in (True, False):
            m = iComformer(width)
            m.load_state_dict(sd)
            m.native_sequence = native
            m = m.to("cuda:0").train(train_mode)
            pred, true = m(_clone(b).to("cuda:0"))
            (pred - true).abs().mean().backward()
            res.append((pred.detach(), {k: p.grad for k, p in m.named_parameters()}, m.state_dict()))
        (p1, g1, s1), (p2, g2, s2) = res
        assert rel_err(p1, p2) < 5e-6, width
        gmax = max(float(g.abs().max()) for g in g2.values() if g is not None)
        for k in g2:
            assert (g1[k] is None) == (g2[k] is None), k
            if g2[k] is not None:
                # (the C++ sequence folds lin_edge into the row block of the first Linears: another summation order; biases
                #  in front of a training-mode BatchNorm have true gradient 0 and carry only rounding noise -- the budget
                #  is the one every gradient has against the reference, tests/test_gpu_model.py)
                assert float((g1[k] - g2[k]).abs().max()) <= 3e-5 * gmax, (width, k)
        for k in s2:
            if "running" in k:
                assert rel_err(s1[k], s2[k]) < 1e-6, k
            elif "num_batches" in k:
                assert int(s1[k]) == int(s2[k]), k


def test_training_helper_backward_without_the_autograd_engine():
    """cartnet_amd.train.backward on compute_loss(iComformer(batch)) with a FlatAdam attached: the two backward functions
    run in the calling thread (no engine), bitwise the gradients of loss.backward()."""
    from cartnet_amd import train as ctrain
    from cartnet_amd.comformer import iComformer, make_icomformer_state_dict
    from cartnet_amd.data import Batch
    from cartnet_amd.optim import FlatAdam
    from cartnet_amd.synthetic import make_crystal
    b = Batch.from_data_list([make_crystal(970 + i, 9 + 5 * i) for i in range(3)])
    sd = make_icomformer_state_dict(256, seed=12)

    def grads(helper):
        m = iComformer(256)
        m.load_state_dict(sd)
        m = m.to("cuda:0").train()
        opt = FlatAdam(m, lr=1e-3)
        opt.zero_grad()
        pred, true = m(_clone(b).to("cuda:0"))
        loss = ctrain.compute_loss(pred, true)[0]
        if helper:
            assert ctrain._two_node_backward(loss, torch.ones((), device="cuda:0"))
        else:
            loss.backward()
        return opt.flat_grad.clone()

    ref = grads(False)
    assert ref.abs().max().item() > 0 and torch.equal(grads(True), ref)
