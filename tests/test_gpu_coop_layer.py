"""The one-launch layer forward prototype (csrc/coop_layer.hip; not on the model's path): its results against
the fp64 oracle's cartnet_layer at BASELINE configs[2] sizes (N = 736, E = 9,970), through the C ABI.  The tool asserts
x_out / e_out within the precision-2 budget (bf16 operands: 3e-2 of the largest value) and that no grid barrier gave up."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cooperative_layer_forward_matches_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_small_batch_layer.py"), "--check-only"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if "against the fp64 oracle" in l]
    assert line, out.stdout
