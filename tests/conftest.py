import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HAVE_REFERENCE = os.path.isdir("/root/reference/models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    have_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU in this container")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(skip_gpu)
        if "reference" in item.keywords and not HAVE_REFERENCE:
            item.add_marker(skip_ref)


def rel_err(a, b):
    """Norm-wise relative error max|a-b| / max|b| (the parity metric of BASELINE.md §2)."""
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.abs().max().item()
    if denom == 0:
        return (a - b).abs().max().item()
    return (a - b).abs().max().item() / denom
