"""GPU periodic radius graph against the reference's radius_graph_pbc (golden fixture) and the CPU restatement."""
import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu


def _check(ei, dist, dirs, ref_ei, ref_dist, ref_dir):
    assert torch.equal(ei.cpu(), ref_ei)                                   # integers: bit-exact, same order
    assert torch.allclose(dist.cpu(), ref_dist, rtol=1e-6, atol=0)         # a few fp32 ulps (image offsets round differently)
    assert torch.allclose(dirs.cpu(), ref_dir, rtol=0, atol=1e-6)


def test_matches_reference_golden_fixture():
    from cartnet_amd.graph import radius_graph_pbc
    z = np.load(gu.GOLDEN + "/radius_graph.npz")
    for i in range(3):
        pos, cell = torch.from_numpy(z[f"pos{i}"]), torch.from_numpy(z[f"cell{i}"])
        ptr = torch.tensor([0, pos.shape[0]])
        ei, dist, dirs = radius_graph_pbc(pos.cuda(), cell.view(1, 3, 3).cuda(), ptr.cuda(), 5.0)
        _check(ei, dist, dirs, torch.from_numpy(z[f"edge_index{i}"]), torch.from_numpy(z[f"dist{i}"]),
               torch.from_numpy(z[f"dir{i}"]))


def test_batch_of_crystals_matches_cpu_builder_and_feeds_the_model():
    from cartnet_amd.data import Batch
    from cartnet_amd.graph import radius_graph_pbc
    from cartnet_amd.synthetic import make_crystal
    items = [make_crystal(600 + g, n) for g, n in enumerate((1, 2, 37, 194, 90))]
    b = Batch.from_data_list(items)
    ei, dist, dirs = radius_graph_pbc(b.pos.cuda(), b.cell.cuda(), b.ptr.cuda(), 5.0)
    _check(ei, dist, dirs, b.edge_index, b.cart_dist, b.cart_dir)
    assert bool((ei[1][1:] >= ei[1][:-1]).all())
    # end to end: the GPU-built graph drives the network to the same prediction as the CPU-built one
    from cartnet_amd.model import CartNet, make_state_dict
    m = CartNet(64, 32, 2)
    m.load_state_dict(make_state_dict(64, 32, 2, seed=9))
    m = m.cuda().eval()
    b1 = b.clone(); b1.num_graphs = b.num_graphs; b1.to("cuda:0")
    b2 = b.clone(); b2.num_graphs = b.num_graphs; b2.to("cuda:0")
    b2.edge_index, b2.cart_dist, b2.cart_dir = ei, dist, dirs
    with torch.no_grad():
        p1, _ = m(b1)
        p2, _ = m(b2)
    assert (p1 - p2).abs().max().item() <= 1e-5 * p1.abs().max().item()
