"""cartnet_amd -- CartNet's message-passing hot path as hand-written gfx950 (MI355X) kernels behind the reference's
model API.  See DESIGN.md.  Importing the package does not need a GPU; running the model does."""
from .config import cfg, set_cfg  # noqa: F401
from .data import Batch, Data, DataLoader  # noqa: F401
from .master import create_model  # noqa: F401
from .model import CartNet  # noqa: F401

__all__ = ["cfg", "set_cfg", "Batch", "Data", "DataLoader", "create_model", "CartNet"]
