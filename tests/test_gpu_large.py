"""Size-independent properties at and beyond the benchmark batch (BASELINE.json configs[1]: 64 crystals x 194 atoms,
E ~ 177k) -- sizes the CPU oracle cannot check element by element in seconds:

* eval mode: BatchNorm uses running statistics, so crystals are independent and the prediction for a crystal must not
  depend on which other crystals share its batch, nor on where its rows fall in the GEMM / CSR tiling;
* train mode: replicating every crystal R times leaves all BatchNorm statistics and the mean loss unchanged, so the
  parameter gradient of the replicated batch equals that of the base batch;
* a batch whose [E, 2D] activations exceed 4 GiB (E ~ 2.2 M edges, 71 GiB workspace) runs through the same entry
  points: byte offsets beyond 2^32, the general-kernel fallback of the DMA-fed GEMMs, 288 GB-sized workspaces.
"""
import pytest
import torch

from conftest import rel_err
from test_gpu_model import _model

pytestmark = pytest.mark.gpu

HP = dict(dim_in=256, dim_rbf=64, num_layers=4, radius=5.0, invariant=False, temperature=True, use_envelope=True,
          atom_types=True, cholesky=True)


def _batch(items, idx):
    from cartnet_amd.data import Batch
    return Batch.from_data_list([items[i] for i in idx]).to("cuda:0")


def _grads(m):
    return torch.cat([p.grad.flatten() for p in m.parameters()])


@pytest.mark.parametrize("precision", [0, 1])
def test_eval_prediction_does_not_depend_on_batch_composition_at_bench_size(precision):
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    items = [make_crystal(5000 + g, 194) for g in range(64)]
    m = _model(HP, make_state_dict(256, 64, 4, seed=12), precision).eval()
    with torch.no_grad():
        full, _ = m(_batch(items, range(64)))
        parts = [m(_batch(items, range(s, s + 16)))[0] for s in range(0, 64, 16)]
        odd = m(_batch(items, [63, 5, 17]))[0]
    cat = torch.cat(parts)
    assert full.shape == cat.shape and torch.isfinite(full).all()
    assert rel_err(full, cat) < 1e-6
    sizes = [int(it.non_H_mask.sum()) for it in items]
    per = torch.split(full, sizes)
    ref = torch.cat([per[63], per[5], per[17]])
    assert rel_err(odd, ref) < 1e-6


@pytest.mark.parametrize("precision", [0, 1])
def test_replicated_batch_beyond_4GiB_activations(precision):
    """16 distinct crystals x 50 replicas = 800 crystals, E ~ 2.2 M: [E, 2D] fp32 = 4.5 GB per saved tensor."""
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    R = 50
    items = [make_crystal(6000 + g, 194) for g in range(16)]
    sd = make_state_dict(256, 64, 4, seed=13)
    m = _model(HP, sd, precision)

    def train_step(idx):
        m.train()
        m.load_state_dict(sd)
        m.zero_grad(set_to_none=True)
        b = _batch(items, idx)
        E = int(b.edge_index.shape[1])
        pred, true = m(b)
        (pred - true).abs().mean().backward()
        return pred.detach(), _grads(m).clone(), E

    p_base, g_base, E_base = train_step(list(range(16)))
    p_big, g_big, E_big = train_step(list(range(16)) * R)
    assert E_big == R * E_base and E_big * 2 * 256 * 4 > 2 ** 32
    assert torch.isfinite(p_big).all() and torch.isfinite(g_big).all()
    # every replica predicts what the base batch predicted (same BatchNorm statistics, same rows)
    n_base = p_base.shape[0]
    assert p_big.shape[0] == R * n_base
    reps = p_big.view(R, n_base, 3, 3)
    assert rel_err(reps[0], p_base) < 2e-5
    assert rel_err(reps[R - 1], p_base) < 2e-5
    assert rel_err(reps[R // 2], reps[0]) < 2e-5
    # gradient of the mean loss is unchanged by replication
    assert rel_err(g_big, g_base) < 2e-4
    del p_big, g_big
    torch.cuda.empty_cache()
