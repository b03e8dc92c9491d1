"""GPU ADP metrics (cartnet_adp_metrics through cartnet_amd.metrics) against the reference's train/metrics.py outputs
(golden fixture), the oracle, and size-independent identities at a full test batch."""
import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu

IOU_TOL = 1e-4      # one voxel of a ~1e4..1e5-voxel union: voxels within fp32 rounding of a surface may flip


def _load():
    return {k: torch.from_numpy(v) for k, v in np.load(gu.GOLDEN + "/adp_metrics.npz").items()}


def test_matches_reference_golden_fixture():
    from cartnet_amd import metrics as gm
    z = _load()
    for name in ("close", "far"):
        pred, true = z[f"{name}_pred"].cuda(), z[f"{name}_true"].cuda()
        vol, sim, iou = gm.adp_metrics(pred, true)
        vol64, sim64 = z[f"{name}_volume_error64"], z[f"{name}_similarity_index64"]
        # against the reference evaluated in fp64: at least as close as the reference's own fp32 run
        ref_vol_err = (z[f"{name}_volume_error"].double() - vol64).abs().max().item()
        ref_sim_err = (z[f"{name}_similarity_index"].double() - sim64).abs().max().item()
        assert (vol.cpu().double() - vol64).abs().max().item() <= max(ref_vol_err, 1e-6 * vol64.abs().max().item())
        assert (sim.cpu().double() - sim64).abs().max().item() <= max(ref_sim_err, 1e-5)
        assert torch.allclose(vol.cpu(), z[f"{name}_volume_error"], rtol=1e-3, atol=1e-6)
        assert torch.allclose(sim.cpu(), z[f"{name}_similarity_index"], rtol=1e-3, atol=1e-2)
        assert (iou.cpu() - z[f"{name}_iou"]).abs().max().item() <= IOU_TOL
        # the single-metric entry points return the same numbers
        assert torch.equal(gm.get_error_volume(pred, true), vol)
        assert torch.equal(gm.get_similarity_index(pred, true), sim)
        assert torch.equal(gm.compute_3D_IoU(pred, true), iou)


def test_matches_oracle_on_other_grids_and_sizes():
    from cartnet_amd import metrics as gm
    from oracle import metrics_ref as om
    z = _load()
    pred, true = z["far_pred"][:9], z["far_true"][:9]
    for P in (1, 7, 32):
        iou = gm.compute_3D_IoU(pred.cuda(), true.cuda(), num_points=P).cpu()
        ref = om.compute_3d_iou(pred, true, num_points=P)
        assert (iou - ref).abs().max().item() <= (1.0 / 3 if P == 7 else IOU_TOL), P
    assert gm.adp_metrics(pred[:0].cuda(), true[:0].cuda())[2].numel() == 0
    with pytest.raises(ValueError):
        gm.adp_metrics(pred.cuda(), true[:3].cuda())
    with pytest.raises(ValueError):
        gm.adp_metrics(pred, true)                       # host tensors: no CPU path


def test_identities_at_a_full_test_batch():
    """12,416 atoms (64 crystals x 194): IoU(A, A) = 1, S12(A, A) = 0, symmetry of the IoU, invariance of the IoU to
    a common scale (both matrices are divided by the larger norm) and to swapping two axes of both ellipsoids."""
    from cartnet_amd import metrics as gm
    g = torch.Generator().manual_seed(3)
    M = 64 * 194
    a = torch.randn(M, 3, 3, generator=g)
    true = (a @ a.transpose(1, 2) * 0.01 + 0.005 * torch.eye(3)).cuda()
    b = torch.randn(M, 3, 3, generator=g) * 0.03
    pred = (true.cpu() + b @ b.transpose(1, 2)).cuda()
    vol, sim, iou = gm.adp_metrics(pred, true)
    assert torch.isfinite(vol).all() and torch.isfinite(sim).all()
    assert bool((iou > 0).all()) and bool((iou <= 1).all()) and bool((sim > -1e-3).all())
    v0, s0, i0 = gm.adp_metrics(true, true)
    assert torch.all(i0 == 1.0) and v0.abs().max().item() < 1e-12 and s0.abs().max().item() < 1e-5
    assert torch.equal(gm.compute_3D_IoU(true, pred), iou)
    assert (gm.compute_3D_IoU(pred * 4.0, true * 4.0) - iou).abs().max().item() <= IOU_TOL   # exact power-of-two scale
    perm = torch.tensor([1, 0, 2], device="cuda")
    sw = lambda m: m[:, perm][:, :, perm]
    assert (gm.compute_3D_IoU(sw(pred), sw(true)) - iou).abs().max().item() <= 5 * IOU_TOL
    assert torch.allclose(gm.get_similarity_index(true, pred), sim, rtol=1e-4, atol=1e-4)     # S12 is symmetric
