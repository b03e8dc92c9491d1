"""ORACLE -- test infrastructure only, never the product path.

CPU restatement (plain torch) of the reference's ADP evaluation metrics (train/metrics.py:30-180).  Pinned against the
reference's own functions run in the build container through tests/golden/adp_metrics.npz
(tests/golden/make_golden.py: metrics_fixture).
"""
from __future__ import annotations

import math

import torch

SMOOTH = 1e-8                                                       # train/metrics.py:11


def get_volume(A: torch.Tensor) -> torch.Tensor:
    """train/metrics.py:30-40."""
    return (4.0 / 3.0) * math.pi * torch.sqrt(torch.linalg.det(A))


def get_error_volume(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """train/metrics.py:42-58; the reference normalises by the PREDICTION's volume (its variable names are swapped)."""
    v1, v2 = get_volume(pred), get_volume(true)
    return torch.abs(v1 - v2) / (v1 + SMOOTH)


def get_similarity_index(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """train/metrics.py:76-94."""
    it, ip = torch.linalg.inv(true), torch.linalg.inv(pred)
    num = 2 ** 1.5 * torch.linalg.det(it @ ip) ** 0.25
    den = torch.linalg.det(it + ip) ** 0.5
    return 100 * (1 - num / den)


def ellipsoid_masks(cov: torch.Tensor, num_points: int = 64) -> torch.Tensor:
    """train/metrics.py:114-146: [M, P, P, P] boolean, True where sqrt(x^T cov^-1 x) < 1 on linspace(-1, 1, P)^3."""
    g = torch.linspace(-1, 1, num_points, dtype=cov.dtype)
    pts = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
    inv = torch.linalg.inv(cov)
    mult = pts.unsqueeze(0) @ inv                                    # [M, P^3, 3]
    maha = torch.sqrt(torch.sum(mult * pts.unsqueeze(0), dim=-1))
    return (maha < 1).reshape(-1, num_points, num_points, num_points)


def compute_3d_iou(pred: torch.Tensor, true: torch.Tensor, num_points: int = 64, chunk: int = 16) -> torch.Tensor:
    """train/metrics.py:148-180 (+ :96-112): both matrices divided by the larger Frobenius norm, voxel IoU."""
    out = []
    for s in range(0, pred.shape[0], chunk):
        p, t = pred[s:s + chunk], true[s:s + chunk]
        np_, nt_ = torch.linalg.matrix_norm(p), torch.linalg.matrix_norm(t)
        nrm = torch.where(np_ > nt_, np_, nt_).unsqueeze(-1).unsqueeze(-1)
        mp, mt = ellipsoid_masks(p / nrm, num_points), ellipsoid_masks(t / nrm, num_points)
        inter = (mp & mt).float().sum((1, 2, 3))
        union = (mp | mt).float().sum((1, 2, 3))
        out.append((inter + SMOOTH) / (union + SMOOTH))
    return torch.cat(out)
