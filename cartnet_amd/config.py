"""Global run configuration, mirroring the attribute names the reference reads from GraphGym's
yacs ``cfg`` singleton (reference: main.py:156-188; models/master.py:23-32; models/cartnet.py:156,201).

yacs / torch_geometric.graphgym are not available on the target image, so this is a plain attribute bag with
the same field names and defaults.  Only the fields the hot path and its immediate callers read are present.
"""
from __future__ import annotations


class _Node:
    """Attribute bag; unknown attributes raise AttributeError like a frozen yacs node would."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        inner = ", ".join(f"{k}={v!r}" for k, v in sorted(self.__dict__.items()))
        return f"cfg({inner})"

    def as_dict(self):
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.as_dict() if isinstance(v, _Node) else v
        return out


def _defaults() -> _Node:
    # Defaults are the argparse defaults of the reference CLI (main.py:123-154).
    return _Node(
        seed=0,
        name="CartNet",
        run_dir="results/CartNet/0",
        batch=4,
        batch_accumulation=16,
        dataset=_Node(name="ADP", task_type="regression"),
        loss="MAE",
        optim=_Node(max_epoch=50),
        lr=1e-3,
        warmup=0.01,
        model="CartNet",
        max_neighbours=-1,
        radius=5.0,
        num_layers=4,
        dim_in=256,
        dim_rbf=64,
        augment=False,
        invariant=False,
        use_temp=True,
        standarize_temp=True,
        envelope=True,
        use_H=True,
        use_atom_types=True,
        workers=0,
        device="cuda:0",
    )


cfg = _defaults()


def set_cfg(node: _Node | None = None) -> _Node:
    """Reset ``cfg`` to defaults in place (reference: main.py:156 ``set_cfg(cfg)``)."""
    target = cfg if node is None else node
    fresh = _defaults()
    target.__dict__.clear()
    target.__dict__.update(fresh.__dict__)
    return target
