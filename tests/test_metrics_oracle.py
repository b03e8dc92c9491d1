"""The metrics oracle against the reference's train/metrics.py outputs (tests/golden/adp_metrics.npz)."""
import numpy as np
import torch

import golden_utils as gu
from oracle import metrics_ref as om


def _load():
    return {k: torch.from_numpy(v) for k, v in np.load(gu.GOLDEN + "/adp_metrics.npz").items()}


def test_oracle_matches_reference_metrics():
    z = _load()
    for name in ("close", "far"):
        pred, true = z[f"{name}_pred"], z[f"{name}_true"]
        assert torch.allclose(om.get_error_volume(pred, true), z[f"{name}_volume_error"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(om.get_similarity_index(pred, true), z[f"{name}_similarity_index"], rtol=1e-4, atol=1e-3)
        assert torch.allclose(om.compute_3d_iou(pred, true), z[f"{name}_iou"], rtol=0, atol=1e-4)
        # fp64 evaluation of the same formulas reproduces the reference's fp64 run
        assert torch.allclose(om.get_error_volume(pred.double(), true.double()), z[f"{name}_volume_error64"],
                              rtol=1e-12, atol=1e-14)
        assert torch.allclose(om.get_similarity_index(pred.double(), true.double()),
                              z[f"{name}_similarity_index64"], rtol=1e-9, atol=1e-10)


def test_metric_identities():
    z = _load()
    t = z["close_true"][:8]
    assert torch.all(om.compute_3d_iou(t, t) == 1.0)
    assert om.get_error_volume(t, t).abs().max() == 0
    assert om.get_similarity_index(t.double(), t.double()).abs().max() < 1e-10
    # a sphere of covariance I/sqrt(3) * s: radius^2 = s/sqrt(3) after normalisation by |.|_F = s
    eye = torch.eye(3).unsqueeze(0)
    m = om.ellipsoid_masks(eye / 3 ** 0.5, 64)
    frac = m.float().mean().item()
    # 64 lattice points span [-1, 1] inclusive: one point per (2/63)^3 of volume
    assert abs(frac - (4 / 3) * np.pi * (3 ** -0.25) ** 3 / 8 * (63 / 64) ** 3) < 2e-3
