"""Self-check of the third-party stand-ins the golden vectors were generated through (tests/golden/_ref_import.py).

The reference's arithmetic for this path lives in PyTorch-Geometric 2.5.2 (`MessagePassing.propagate`) and
pytorch-scatter 2.1.1 (`scatter`), neither installed here; the golden fixtures come from the reference's own
models/cartnet.py executed over small stand-ins for those two.  What the fixtures therefore cannot pin by themselves
is the stand-ins' reading of PyG's published convention for flow='source_to_target':

    edge_index[0] = source j,  edge_index[1] = target i;   <arg>_i = arg[edge_index[1]],  <arg>_j = arg[edge_index[0]];
    messages are summed at index = edge_index[1]

This file fixes that reading on an ASYMMETRIC three-node graph with hand-computed expectations, for the stand-in, for
the oracle's layer, and (tests/test_gpu_equivariance.py) for the HIP path: swapping the roles of the two rows of
edge_index changes every number below.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

from _ref_import import _MessagePassing, _scatter  # noqa: E402

# edges (source -> target): 0 -> 1, 2 -> 1, 1 -> 0; node 2 receives nothing.  Sorted by target like the reference's.
EDGE_INDEX = torch.tensor([[1, 0, 2],      # row 0: source j
                           [0, 1, 1]])     # row 1: target i
X = torch.tensor([[1.0], [2.0], [4.0]])
E_ATTR = torch.tensor([[0.1], [0.2], [0.3]])


class _Probe(_MessagePassing):
    """message = 100 x_i + 10 x_j + e: the digits tell target, source and edge apart."""

    def forward(self, x, e, edge_index):
        return self.propagate(edge_index, x=x, e=e)

    def message(self, x_i, x_j, e):
        return 100.0 * x_i + 10.0 * x_j + e


def test_propagate_i_is_target_row1_and_j_is_source_row0():
    out = _Probe()(X, E_ATTR, EDGE_INDEX)
    # edge 0 (1 -> 0): 100*x[0] + 10*x[1] + 0.1 = 120.1        summed at node 0
    # edge 1 (0 -> 1): 100*x[1] + 10*x[0] + 0.2 = 210.2   \
    # edge 2 (2 -> 1): 100*x[1] + 10*x[2] + 0.3 = 240.3   /    summed at node 1 = 450.5
    expect = torch.tensor([[120.1], [450.5], [0.0]])
    assert torch.allclose(out, expect, atol=1e-5)
    # the transposed reading (i = row 0) would give [[210.2 + ...]] -- make sure it is NOT what we get
    wrong = torch.tensor([[100 * 2 + 10 * 1 + 0.2 + 0.0], [0.0], [0.0]])
    assert not torch.allclose(out, wrong)


def test_scatter_sum_and_mean():
    src = torch.tensor([[1.0, 10.0], [2.0, 20.0], [4.0, 40.0]])
    idx = torch.tensor([0, 1, 1])
    assert torch.equal(_scatter(src, idx, 0, None, 3, "sum"), torch.tensor([[1.0, 10.0], [6.0, 60.0], [0.0, 0.0]]))
    assert torch.equal(_scatter(src, idx, 0, None, 3, "mean"), torch.tensor([[1.0, 10.0], [3.0, 30.0], [0.0, 0.0]]))


def test_oracle_layer_uses_the_same_convention():
    """oracle/cartnet_ref.py:cartnet_layer on the same graph with hand-set weights: the gate is switched to a constant
    (BatchNorm weight 0, bias b -> sigma = sigmoid(b)), the sender MLP reads out x_i, x_j, e separately."""
    from oracle import cartnet_ref as orc
    D = 4
    x = torch.tensor([[1.0, 0, 0, 0], [2.0, 0, 0, 0], [4.0, 0, 0, 0]], dtype=torch.float64)
    e = torch.tensor([[0.1, 0, 0, 0], [0.2, 0, 0, 0], [0.3, 0, 0, 0]], dtype=torch.float64)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64)
    sd = {}
    p = "layers.0"
    W0 = z(D, 3 * D)
    W0[0, 0] = 1.0          # hidden[0] = x_i[0]
    W0[1, D] = 1.0          # hidden[1] = x_j[0]
    W0[2, 2 * D] = 1.0      # hidden[2] = e[0]
    sd[p + ".MLP_aggr.0.weight"], sd[p + ".MLP_aggr.0.bias"] = W0, z(D)
    W2 = z(D, D)
    W2[0, 0], W2[1, 1], W2[2, 2] = 1.0, 1.0, 1.0
    sd[p + ".MLP_aggr.2.weight"], sd[p + ".MLP_aggr.2.bias"] = W2, z(D)
    sd[p + ".MLP_gate.0.weight"], sd[p + ".MLP_gate.0.bias"] = z(D, 3 * D), z(D)
    sd[p + ".MLP_gate.2.weight"], sd[p + ".MLP_gate.2.bias"] = z(D, D), z(D)
    sd[p + ".norm.weight"], sd[p + ".norm.bias"] = z(D), z(D)                    # sigma = sigmoid(0) = 0.5
    sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"] = z(D), torch.ones(D, dtype=torch.float64)
    sd[p + ".norm2.weight"], sd[p + ".norm2.bias"] = torch.ones(D, dtype=torch.float64), z(D)
    sd[p + ".norm2.running_mean"] = z(D)
    sd[p + ".norm2.running_var"] = torch.ones(D, dtype=torch.float64) - orc.BN_EPS   # BatchNorm = identity in eval
    dist = torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64)
    x_out, e_out = orc.cartnet_layer(sd, 0, x, e, EDGE_INDEX, dist, 5.0, False, False)
    silu = torch.nn.functional.silu
    s = lambda v: silu(torch.tensor(v, dtype=torch.float64)).item()
    # aggregated sender (before norm2 / SiLU / residual): column 0 sums silu(x_i), column 1 silu(x_j), column 2 silu(e)
    agg = torch.tensor([[0.5 * s(1.0), 0.5 * s(2.0), 0.5 * s(0.1), 0.0],
                        [0.5 * (s(2.0) + s(2.0)), 0.5 * (s(1.0) + s(4.0)), 0.5 * (s(0.2) + s(0.3)), 0.0],
                        [0.0, 0.0, 0.0, 0.0]], dtype=torch.float64)
    assert torch.allclose(x_out, silu(agg) + x, atol=1e-12)
    assert torch.allclose(e_out, e + 0.5, atol=1e-12)
