"""PyTorch-ROCm bundles its own HIP runtime; the product must work whichever of the two is imported
first in a process (uzkge_amd/_native.py loads torch's copy when torch is installed)."""
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SNIPPET = """
import sys
sys.path.insert(0, %r)
%s
from uzkge_amd import backend as b
b.init(0)
import numpy as np
x = np.zeros((16, 4), dtype=np.uint64); x[1, 0] = 5
assert b.ntt(x).shape == (16, 4)
import torch
t = torch.zeros(8, device="cuda")
assert float(t.sum()) == 0.0
print("ok")
"""


@pytest.mark.parametrize("first", ["", "import torch"])
def test_library_and_torch_in_either_order(first):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", SNIPPET % (root, first)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
