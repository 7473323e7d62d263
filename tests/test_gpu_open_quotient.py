"""Opening quotient on the device (div_rem / multi-point Horner) against the oracle's
literal restatement of pcs.rs:119-135 + field_polynomial.rs:519-550, and the identity it must satisfy:
q(r) (r - z) = sum_k alpha^k (p_k(r) - p_k(z)) at a random r.  Parity unpinned (no reference fixture)."""
import numpy as np
import pytest
import torch

import bn254_py as opy
import oracle_c as oc
from util import rand_fr_wire

pytestmark = pytest.mark.gpu


def _run(gpu, polys, z, alpha):
    batch, n = polys.shape[0], polys.shape[1]
    d = torch.from_numpy(polys.view(np.int64)).to("cuda")
    q = torch.full((n, 4), -1, dtype=torch.int64, device="cuda")
    ev = gpu.open_quotient_device(d.data_ptr(), n, batch, z, alpha, q.data_ptr())
    return q.cpu().numpy().view(np.uint64), ev


@pytest.mark.parametrize("n,batch", [(1, 1), (2, 1), (17, 3), (4096, 16), (4097, 2), (16387, 16), (100000, 4), (1 << 20, 1)])
def test_matches_reference_loop(gpu, n, batch):
    polys = rand_fr_wire(n * batch, 7 + n).reshape(batch, n, 4)
    s = rand_fr_wire(2, 8 + n)
    q, ev, ok = oc.open_quotient(polys, s[0], s[1])
    assert ok
    for rep in range(2):              # sizes on both sides of 2^16: 4 / 16 coefficients per lane in the division kernels, both evaluation kernels
        q_gpu, ev_gpu = _run(gpu, polys, s[0], s[1])
        assert np.array_equal(ev_gpu, ev), rep
        assert np.array_equal(q_gpu, q), rep


def test_division_identity(gpu):
    n, batch = 5000, 5
    polys = rand_fr_wire(n * batch, 99).reshape(batch, n, 4)
    z, alpha, r = rand_fr_wire(3, 100)
    q, ev = _run(gpu, polys, z, alpha)
    zi, ai, ri = oc.fr_to_ints(np.stack([z, alpha, r]))
    lhs = oc.fr_to_ints(oc.poly_eval(q, r)[None, :])[0] * (ri - zi) % opy.R
    rhs = 0
    for k in range(batch):
        pk_r = oc.fr_to_ints(oc.poly_eval(polys[k], r)[None, :])[0]
        rhs = (rhs + pow(ai, k, opy.R) * (pk_r - oc.fr_to_ints(ev[k][None, :])[0])) % opy.R
    assert lhs == rhs


def test_special_points(gpu):
    """z = 0 (q is a shift of h), z = 1, and a polynomial with trailing zero coefficients."""
    n, batch = 300, 2
    polys = rand_fr_wire(n * batch, 5).reshape(batch, n, 4)
    polys[:, 250:] = 0
    alpha = rand_fr_wire(1, 6)[0]
    for zv in (0, 1):
        z = oc.fr_from_ints([zv])[0]
        q_gpu, ev_gpu = _run(gpu, polys, z, alpha)
        q, ev, ok = oc.open_quotient(polys, z, alpha)
        assert ok and np.array_equal(q_gpu, q) and np.array_equal(ev_gpu, ev)


def test_argument_errors(gpu):
    from uzkge_amd.errors import UzkgeError
    polys = rand_fr_wire(8, 1).reshape(1, 8, 4)
    d = torch.from_numpy(polys.view(np.int64)).to("cuda")
    q = torch.empty((8, 4), dtype=torch.int64, device="cuda")
    z, alpha = rand_fr_wire(2, 2)
    with pytest.raises(UzkgeError):
        gpu.open_quotient_device(d.data_ptr(), 0, 1, z, alpha, q.data_ptr())
    with pytest.raises(UzkgeError):
        gpu.open_quotient_device(d.data_ptr(), (1 << 20) + 1, 1, z, alpha, q.data_ptr())
    with pytest.raises(UzkgeError):
        gpu.open_quotient_device(d.data_ptr(), 8, 1, z, alpha, d.data_ptr())
