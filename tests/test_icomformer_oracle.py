"""Pin the iComformer oracle (oracle/icomformer_ref.py) against golden vectors generated from the reference's own
models/comformer.py (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import icomformer_utils as iu
from conftest import rel_err
from oracle import icomformer_ref as orc


@pytest.mark.parametrize("name", iu.FIXTURES)
def test_forward_matches_reference(name):
    z, b, sd = iu.load(name)
    for mode, training in (("train", True), ("eval", False)):
        pred = orc.icomformer_forward(sd, b, training=training)
        assert rel_err(pred, torch.from_numpy(z[f"{mode}_pred"])) < 1e-5, (name, mode)
        pred64 = orc.icomformer_forward(iu.to64(sd), iu.batch64(b), training=training)
        assert rel_err(pred64, torch.from_numpy(z[f"{mode}_pred_f64"])) < 1e-11, (name, mode)


@pytest.mark.parametrize("name", iu.FIXTURES)
def test_gradients_and_bn_state_match_reference(name):
    z, b, sd = iu.load(name)
    unused = set(z["unused_params"].tolist())
    sd64 = {k: (v.double().requires_grad_(("grad64_" + k) in z.files) if v.is_floating_point() else v)
            for k, v in sd.items()}
    new_stats = {}
    pred = orc.icomformer_forward(sd64, iu.batch64(b), training=True, new_stats=new_stats)
    mae = (pred - b.y.double()).abs().mean()
    assert abs(mae.item() - float(z["train_mae"])) < 1e-5 * abs(float(z["train_mae"]))
    mae.backward()
    names = [k for k in sd64 if ("grad64_" + k) in z.files]
    assert len(names) > 50 and unused == {"edge_update_layer.lemb.weight", "edge_update_layer.lin_edge_len.weight",
                                          "edge_update_layer.lin_edge_len.bias"}
    gmax = max(float(np.abs(z["grad64_" + k]).max()) for k in names)
    for k in names:
        ref = torch.from_numpy(z["grad64_" + k])
        assert (sd64[k].grad - ref).abs().max().item() < 1e-9 * gmax, k
    for k, v in new_stats.items():
        ref = torch.from_numpy(z["state_" + k])
        if v.is_floating_point():
            assert rel_err(v, ref) < 1e-5, k
        else:
            assert int(v) == int(ref), k
