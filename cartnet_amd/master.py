"""Model factory with the reference's contract (reference: models/master.py:9-46): reads the global ``cfg``."""
from __future__ import annotations

from .config import cfg


def create_model():
    """cfg.model == "CartNet" -> CartNet(dim_in, dim_rbf, num_layers, invariant, use_temp, envelope, use_atom_types,
    cholesky = (dataset == "ADP")) moved to ``cfg.device`` (the reference hard-codes "cuda:0").
    cfg.model == "icomformer" / "ecomformer" -> iComformer(dim_in) / eComformer(dim_in) (ADP only,
    models/master.py:34-43); eComformer's equivariant block is restated without e3nn (parity with e3nn unpinned)."""
    if cfg.model == "CartNet":
        from .model import CartNet
        model = CartNet(dim_in=cfg.dim_in, dim_rbf=cfg.dim_rbf, num_layers=cfg.num_layers, invariant=cfg.invariant,
                        temperature=cfg.use_temp, use_envelope=cfg.envelope, atom_types=cfg.use_atom_types,
                        cholesky=True if cfg.dataset.name == "ADP" else False).to(getattr(cfg, "device", "cuda:0"))
    elif cfg.model == "icomformer":
        from .comformer import iComformer
        assert cfg.dataset.name == "ADP", "iComformer only for ADP dataset"
        model = iComformer(dim_in=cfg.dim_in).to(getattr(cfg, "device", "cuda:0"))
    elif cfg.model == "ecomformer":
        from .comformer import eComformer
        assert cfg.dataset.name == "ADP", "eComformer only for ADP dataset"
        model = eComformer(dim_in=cfg.dim_in).to(getattr(cfg, "device", "cuda:0"))
    else:
        raise Exception("Model not implemented")
    model.gemm_precision = int(getattr(cfg, "gemm_precision", 0))
    if hasattr(model, "bn_group_size"):
        model.bn_group_size = int(getattr(cfg, "bn_group_size", 0))
    if hasattr(model, "half_storage"):
        model.half_storage = bool(getattr(cfg, "half_storage", False))
    if hasattr(model, "sync_batchnorm"):
        model.sync_batchnorm = bool(getattr(cfg, "sync_batchnorm", False))
    return model
