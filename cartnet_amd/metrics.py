"""ADP evaluation metrics on the GPU: the step right after the hot path at test time.

Same names and argument meaning as the reference's ``train/metrics.py`` (:30-180) so that
``compute_metrics_and_logging`` (:201-214) and the inference loops (main.py:47-49,101-102) can call them unchanged;
all three metrics of a batch come from one kernel launch (``adp_metrics``).  There is no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import lib as _l

SMOOTH = 1e-8          # train/metrics.py:11


def _check_pair(pred: torch.Tensor, true: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, int]:
    for name, t in (("pred", pred), ("true", true)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and tuple(t.shape[1:]) == (3, 3)):
            raise ValueError(f"{name} must be a CUDA fp32 tensor [M,3,3]")
    if pred.shape[0] != true.shape[0]:
        raise ValueError("pred and true must hold the same number of atoms")
    return pred.detach().contiguous(), true.detach().contiguous(), int(pred.shape[0])


def adp_metrics(pred: torch.Tensor, true: torch.Tensor, volume: bool = True, similarity: bool = True,
                iou: bool = True, num_points: int = 64
                ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(volume_error [M], similarity_index [M], iou [M]) of the requested metrics, ``None`` for the others."""
    pred, true, M = _check_pair(pred, true)
    lib = _l.load()
    dev = pred.device
    outs = [torch.empty(M, dtype=torch.float32, device=dev) if want else None for want in (volume, similarity, iou)]
    if M == 0 or not any((volume, similarity, iou)):
        return tuple(outs)
    grid = torch.linspace(-1, 1, num_points, device=dev, dtype=torch.float32) if iou else None   # metrics.py:128
    _l.check(lib.cartnet_adp_metrics(pred.data_ptr(), true.data_ptr(), M, grid.data_ptr() if iou else None,
                                     int(num_points), *[o.data_ptr() if o is not None else None for o in outs],
                                     _l.stream_ptr()), "cartnet_adp_metrics")
    return tuple(outs)


def get_error_volume(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """train/metrics.py:42-58."""
    return adp_metrics(pred, true, True, False, False)[0]


def get_similarity_index(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """train/metrics.py:76-94."""
    return adp_metrics(pred, true, False, True, False)[1]


def compute_3D_IoU(pred: torch.Tensor, true: torch.Tensor, num_points: int = 64) -> torch.Tensor:
    """train/metrics.py:148-180 (with get_ellipsoids :114-146 and iou_pytorch3D :96-112 fused)."""
    return adp_metrics(pred, true, False, False, True, num_points)[2]
