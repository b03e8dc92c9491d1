"""Edge cases and host-integration properties of the GPU path that the golden fixtures do not cover."""
import pytest
import torch

import golden_utils as gu
from conftest import rel_err
from test_gpu_model import PRED_TOL, _check_grads, _model

pytestmark = pytest.mark.gpu


def _fresh(b):
    c = b.clone()
    c.num_graphs = b.num_graphs
    return c.to("cuda:0")


def test_crystal_without_edges_and_isolated_atoms():
    """A crystal whose atoms have no neighbour inside the cutoff (huge cell) next to normal ones: rows of the CSR are
    empty, the aggregation is zero there, nothing divides by a zero degree; checked against the fp64 oracle."""
    from cartnet_amd.data import Batch, Data
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    from oracle import cartnet_ref as orc
    lonely = make_crystal(1, 3)
    lonely.edge_index = torch.zeros(2, 0, dtype=torch.int64)
    lonely.cart_dist, lonely.cart_dir = torch.zeros(0), torch.zeros(0, 3)
    b = Batch.from_data_list([make_crystal(2, 9), lonely, make_crystal(3, 14)])
    hp = dict(dim_in=32, dim_rbf=16, num_layers=2, radius=5.0, invariant=False, temperature=True, use_envelope=True,
              atom_types=True, cholesky=True)
    sd = make_state_dict(32, 16, 2, seed=4)
    m = _model(hp, sd).train()
    pred, true = m(_fresh(b))
    (pred - true).abs().mean().backward()
    sd64 = {k: (v.double().requires_grad_(k in dict(m.named_parameters())) if v.is_floating_point() else v)
            for k, v in sd.items()}
    b64 = gu.clone_batch(b)
    for k, v in list(b64.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b64, k, v.double())
    ref = orc.cartnet_forward(sd64, b64, training=True, **gu.oracle_kwargs(hp))
    assert rel_err(pred, ref) < PRED_TOL
    (ref - b64.y).abs().mean().backward()
    _check_grads({k: p.grad for k, p in m.named_parameters()}, {k: sd64[k].grad for k, _ in m.named_parameters()},
                 "lonely")


def test_gradient_accumulation_and_flat_adam_match_torch_adam():
    """Two micro-batches accumulated (train/train.py:183-189: no rescaling) then one optimiser step: FlatAdam on the
    flat buffers == torch.optim.Adam on autograd's per-parameter gradients."""
    from cartnet_amd.optim import FlatAdam
    z, hp, b, sd = gu.load("config1")
    from cartnet_amd.data import Batch
    ma, mb = _model(hp, sd).train(), _model(hp, sd).train()
    opt_a = FlatAdam(ma, lr=1e-3)
    opt_b = torch.optim.Adam(mb.parameters(), lr=1e-3)
    opt_a.zero_grad()
    opt_b.zero_grad()
    for _ in range(2):
        for m in (ma, mb):
            m.load_state_dict({k: v for k, v in m.state_dict().items()})      # no-op, keeps buffers in place
            pred, true = m(_fresh(b))
            (pred - true).abs().mean().backward()
    ga = torch.cat([p.grad.flatten() for p in ma.parameters()])
    gb = torch.cat([p.grad.flatten() for p in mb.parameters()])
    assert rel_err(ga, gb) < 1e-6                       # same kernels, same order: only the accumulation path differs
    opt_a.step()
    opt_b.step()
    pa = torch.cat([p.detach().flatten() for p in ma.parameters()])
    pb = torch.cat([p.detach().flatten() for p in mb.parameters()])
    assert rel_err(pa, pb) < 1e-6


def test_eval_mode_backward_and_second_backward_is_rejected():
    z, hp, b, sd = gu.load("tiny_adp")
    m = _model(hp, sd).eval()                              # running statistics, but gradients requested
    pred, true = m(_fresh(b))
    loss = (pred - true).abs().mean()
    loss.backward(retain_graph=True)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    from oracle import cartnet_ref as orc
    sd64 = {k: (v.double().requires_grad_(k in dict(m.named_parameters())) if v.is_floating_point() else v)
            for k, v in sd.items()}
    b64 = gu.clone_batch(b)
    for k, v in list(b64.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b64, k, v.double())
    ref = orc.cartnet_forward(sd64, b64, training=False, **gu.oracle_kwargs(hp))
    (ref - b64.y).abs().mean().backward()
    _check_grads({k: p.grad for k, p in m.named_parameters()}, {k: sd64[k].grad for k, _ in m.named_parameters()},
                 "eval-backward")
    with pytest.raises(RuntimeError, match="called twice|without saved state"):
        loss.backward()                                    # the saved activations were consumed in place


def test_input_validation_errors_are_raised_on_the_host():
    z, hp, b, sd = gu.load("tiny_adp")
    m = _model(hp, sd).eval()
    bad = _fresh(b)
    bad.x = bad.x.float()
    with pytest.raises(ValueError, match="int64 atomic numbers"):
        m(bad)
    bad = _fresh(b)
    bad.cart_dist = bad.cart_dist[:-1]
    with pytest.raises(ValueError, match="cart_dist"):
        m(bad)
    bad = gu.clone_batch(b)          # batch left on the CPU
    with pytest.raises(ValueError, match="batch.to"):
        m(bad)
    unsorted = _fresh(b)
    unsorted.edge_index = unsorted.edge_index.flip(1).contiguous()
    with pytest.raises(ValueError, match="sorted"):
        m(unsorted)


@pytest.mark.parametrize("damage", ["unsorted", "index_out_of_range", "edge_across_crystals"])
def test_malformed_graph_without_validation_cannot_fault_and_is_reported_later(damage):
    """validate_graph = False (the default, no host sync per batch): a malformed edge_index must neither fault the GPU
    (gather indices are clamped, missing CSR/CSC entries default to empty segments) nor pass silently -- the status
    word travels to pinned memory behind the kernels and a later forward call raises."""
    z, hp, b, sd = gu.load("tiny_adp")
    m = _model(hp, sd).train()
    m.validate_graph = False
    bad = _fresh(b)
    N = int(bad.x.shape[0])
    ei = bad.edge_index.clone()
    if damage == "unsorted":
        ei = ei.flip(1).contiguous()
    elif damage == "index_out_of_range":
        ei[0, 0] = N + 12345
        ei[0, 1] = -7
    else:
        ei[0, 0] = N - 1 if int(ei[1, 0]) == 0 else 0     # first edge now starts in another crystal
        if int(bad.num_graphs) == 1:
            pytest.skip("needs a batch of several crystals")
    bad.edge_index = ei
    pred, true = m(bad)                                   # enqueues; numbers are meaningless, nothing may fault
    (pred - true).abs().mean().backward()
    torch.cuda.synchronize()
    m.zero_grad(set_to_none=True)
    with pytest.raises(ValueError, match="earlier forward call"):
        for _ in range(9):                                # good batches; the report arrives with one of them
            m(_fresh(b))
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    pred, _ = m(_fresh(b))                                # the model keeps working afterwards
    assert torch.isfinite(pred).all()


def test_large_ragged_batch_is_finite_and_reproducible():
    """48 crystals of 64..324 atoms (the ADP size distribution of SURVEY.md 8d): finite, bit-reproducible."""
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_regular_batch, make_batch
    b = make_batch(12, None, first=2000)                  # variable sizes, real periodic graphs
    hp = dict(dim_in=256, dim_rbf=64, num_layers=4, radius=5.0, invariant=False, temperature=True,
              use_envelope=True, atom_types=True, cholesky=True)
    m = _model(hp, make_state_dict(256, 64, 4, seed=8)).train()
    outs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m.load_state_dict(make_state_dict(256, 64, 4, seed=8))
        pred, true = m(_fresh(b))
        (pred - true).abs().mean().backward()
        outs.append((pred.detach().clone(), torch.cat([p.grad.flatten() for p in m.parameters()]).clone()))
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("precision", [0, 1])
def test_batch_without_any_edge_at_width_256(precision):
    """E = 0 for the whole batch at the width that takes the DMA-fed kernels: every E-row GEMM is an empty launch, the
    weight-gradient GEMMs reduce over zero rows (gradient 0), BatchNorm over zero edges must not divide by zero."""
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    from oracle import cartnet_ref as orc
    items = []
    for i, n in enumerate((3, 1, 5)):
        d = make_crystal(40 + i, n)
        d.edge_index = torch.zeros(2, 0, dtype=torch.int64)
        d.cart_dist, d.cart_dir = torch.zeros(0), torch.zeros(0, 3)
        items.append(d)
    b = Batch.from_data_list(items)
    hp = dict(dim_in=256, dim_rbf=64, num_layers=2, radius=5.0, invariant=False, temperature=True, use_envelope=True,
              atom_types=True, cholesky=True)
    sd = make_state_dict(256, 64, 2, seed=6)
    m = _model(hp, sd).train()
    m.gemm_precision = precision
    pred, true = m(_fresh(b))
    (pred - true).abs().mean().backward()
    sd64 = {k: (v.double().requires_grad_(k in dict(m.named_parameters())) if v.is_floating_point() else v)
            for k, v in sd.items()}
    b64 = gu.clone_batch(b)
    for k, v in list(b64.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b64, k, v.double())
    ref = orc.cartnet_forward(sd64, b64, training=True, **gu.oracle_kwargs(hp))
    assert torch.isfinite(pred).all() and rel_err(pred, ref) < PRED_TOL
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


@pytest.mark.parametrize("what", ["z_high", "z_negative", "batch_id"])
def test_atomic_number_or_batch_id_outside_its_table_is_reported_not_faulted(what):
    """nn.Embedding raises for an index outside its 119 rows (cartnet.py:113,145); the HIP path clamps what it gathers
    through (no out-of-bounds read of the embedding table / the temperature vector, forward or backward) and raises
    through the graph status word -- immediately with validate_graph, otherwise at the next flush_graph_checks()."""
    z, hp, b, sd = gu.load("tiny_adp")
    for immediate in (True, False):
        m = _model(hp, sd).train()
        m.validate_graph = immediate
        bad = _fresh(b)
        if what == "z_high":
            bad.x[1] = 119
        elif what == "z_negative":
            bad.x[0] = -1
        else:
            bad.batch[2] = 10 ** 6
        needle = "batch id" if what == "batch_id" else "atomic number"
        if immediate:
            with pytest.raises(ValueError, match=needle):
                m(bad)
        else:
            pred, true = m(bad)
            (pred - true).abs().mean().backward()      # backward also reads the clamped ids: nothing may fault
            torch.cuda.synchronize()
            with pytest.raises(ValueError, match=needle):
                m.flush_graph_checks()
            m.flush_graph_checks()                      # drained: a second call is a no-op
        pred, _ = m(_fresh(b))                          # the model keeps working afterwards
        assert torch.isfinite(pred).all()


def test_train_epoch_flushes_the_graph_checks_before_the_optimiser_step():
    """A malformed LAST batch of an epoch used to go unreported (its status word was only read by a later forward);
    train_epoch now drains the pending words before every optimiser step."""
    from cartnet_amd.optim import FlatAdam
    from cartnet_amd.train import train_epoch
    z, hp, b, sd = gu.load("tiny_adp")
    m = _model(hp, sd).train()
    m.validate_graph = False
    opt = FlatAdam(m, lr=1e-3)
    before = opt.flat_param.clone()
    bad = gu.clone_batch(b)
    bad.edge_index = b.edge_index.flip(1)               # no longer sorted by target
    with pytest.raises(ValueError, match="not sorted"):
        train_epoch([gu.clone_batch(b), bad], m, opt, batch_accumulation=2)
    assert torch.equal(opt.flat_param, before)          # the bad gradient never reached the weights


def test_optimizer_state_is_interchangeable_with_torch_adam():
    """best.ckpt["optimizer_state"] (train/train.py:92-95) in torch.optim.Adam's layout, both directions: a torch Adam
    over the same model loads FlatAdam's state and takes the same next step; FlatAdam loads a torch Adam state.  Both
    optimisers are fed the SAME gradient tensors (Adam divides by sqrt(v): for the biases in front of a training-mode
    BatchNorm, whose true gradient is 0, two independent backward passes would hand it different rounding noise)."""
    from cartnet_amd.optim import FlatAdam
    z, hp, b, sd = gu.load("tiny_adp")

    def grads(m):
        pred, true = m(_fresh(b))
        (pred - true).abs().mean().backward()

    def copy_model(m):
        return _model(hp, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}).train()

    ma = _model(hp, sd).train()
    fa = FlatAdam(ma, lr=2e-3)
    for _ in range(2):                                     # two real steps: a non-trivial state
        fa.zero_grad()
        grads(ma)
        fa.step()
    state = fa.state_dict()
    assert set(state) == {"state", "param_groups"} and len(state["state"]) == len(list(ma.parameters()))
    assert state["param_groups"][0]["params"] == list(range(len(fa.params)))
    # FlatAdam -> torch: same state, same gradients, same third step
    mc = copy_model(ma)
    tc = torch.optim.Adam(mc.parameters(), lr=2e-3)
    tc.load_state_dict(state)
    fa.zero_grad()
    grads(ma)
    for pa, pc in zip(ma.parameters(), mc.parameters()):
        pc.grad = pa.grad.detach().clone()
    fa.step()
    tc.step()
    for (k, pa), pc in zip(ma.named_parameters(), mc.parameters()):
        assert rel_err(pa, pc) < 1e-6, k
    # torch -> FlatAdam: load torch's state (three steps) into a fresh FlatAdam and take the same fourth step
    md = copy_model(mc)
    fd = FlatAdam(md, lr=1e-3)
    fd.load_state_dict(tc.state_dict())
    assert fd.step_count == 3 and fd.lr == 2e-3
    tc.zero_grad(set_to_none=False)
    grads(mc)
    fd.zero_grad()
    for pc, pd in zip(mc.parameters(), md.parameters()):
        pd.grad.copy_(pc.grad)
    fd.step()
    tc.step()
    for (k, pd), pc in zip(md.named_parameters(), mc.parameters()):
        assert rel_err(pd, pc) < 1e-6, k
    # the flat layout of round-1 checkpoints still loads
    fd.load_state_dict({"step": 7, "lr": 5e-4, "exp_avg": fd.exp_avg.clone(), "exp_avg_sq": fd.exp_avg_sq.clone()})
    assert fd.step_count == 7 and fd.lr == 5e-4


@pytest.mark.parametrize("case", ["no_edges", "isolated_atoms_and_ragged", "width_512"])
def test_half_storage_edge_cases(case):
    """CartNet.half_storage (gemm_precision 2, bf16 storage): a batch without any edge, a batch with isolated atoms and a
    ragged edge count (E % 16 != 0: the weight-gradient kernel masks its last K-step), and D = 512 (two column tiles per
    group) -- finite, reproducible, and as close to the fp64 oracle as the same model with fp32 storage has to be."""
    from cartnet_amd.data import Batch
    from cartnet_amd.model import make_state_dict
    from cartnet_amd.synthetic import make_crystal
    D = 512 if case == "width_512" else 256
    items = [make_crystal(70 + i, n) for i, n in enumerate((3, 1, 17, 6))]
    if case == "no_edges":
        for d in items:
            d.edge_index = torch.zeros(2, 0, dtype=torch.int64)
            d.cart_dist, d.cart_dir = torch.zeros(0), torch.zeros(0, 3)
    elif case == "isolated_atoms_and_ragged":
        d = items[2]                                  # drop every edge into atom 5 and one more: isolated target, odd E
        keep = d.edge_index[1] != 5
        keep[int(torch.nonzero(keep)[0])] = False
        d.edge_index, d.cart_dist, d.cart_dir = d.edge_index[:, keep], d.cart_dist[keep], d.cart_dir[keep]
    b = Batch.from_data_list(items)
    hp = dict(dim_in=D, dim_rbf=64, num_layers=2, radius=5.0, invariant=False, temperature=True, use_envelope=True,
              atom_types=True, cholesky=True)
    sd = make_state_dict(D, 64, 2, seed=8)
    res = []
    for half in (False, True, True):
        m = _model(hp, sd).train()
        m.gemm_precision, m.half_storage = 2, half
        pred, true = m(_fresh(b))
        (pred - true).abs().mean().backward()
        g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        assert torch.isfinite(pred).all() and torch.isfinite(g).all()
        res.append((pred.detach().clone(), g.clone()))
    assert torch.equal(res[1][0], res[2][0]) and torch.equal(res[1][1], res[2][1])
    # both storage modes against the fp64 oracle, at the bf16 mode's tolerances (tests/test_gpu_model.py)
    from test_gpu_model import BF16_GRAD_TOL, BF16_PRED_TOL, _oracle_run
    names = [k for k, _ in m.named_parameters()]
    ref, gref = _oracle_run(b, hp, sd, set(names))
    gflat = torch.cat([gref[k].reshape(-1) for k in names]).cuda()
    for pred, g in (res[0], res[1]):
        assert rel_err(pred, ref) < BF16_PRED_TOL
        # 27 atoms, MAE loss: the worst entry is a bias gradient of the aggregation branch (a column sum with heavy
        # cancellation) -- 0.04 of the largest gradient with fp32 storage at D = 512, 0.08 with bf16 storage; twice the
        # 64-crystal tolerance
        assert (g.double() - gflat).abs().max().item() <= 2 * BF16_GRAD_TOL * gflat.abs().max().item()
