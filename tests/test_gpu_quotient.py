"""The fused t_poly quotient kernel (uzkge/src/plonk/helpers.rs:284-656) against
the oracle's term-by-term restatement of the reference loop.  The GPU kernel factors the expression
differently (shared selector sums, shared fifth powers), so agreement checks the algebra, not a copy.
Parity for this row is unpinned: the reference stores no fixture for these intermediate values."""
import numpy as np
import pytest
import torch

import bn254_py as opy
import oracle_c as oc
from util import rand_fr_wire

pytestmark = pytest.mark.gpu


def _scalars(seed):
    s = rand_fr_wire(12, seed)
    return dict(alpha=s[0], beta=s[1], gamma=s[2], k=s[3:8], anemoi_g=s[8],
                anemoi_g_inv=oc.fr_inv(s[8]), edwards_a=s[9])


def _run(gpu, n, factor, vecs, sc, zhi):
    m = n * factor
    dev = torch.from_numpy(vecs.view(np.int64)).to("cuda")          # [56, m, 4]
    out = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    ptrs = [dev[i].data_ptr() for i in range(56)]
    gpu.t_quotient_device(n, factor, ptrs, sc["alpha"], sc["beta"], sc["gamma"], sc["k"], sc["anemoi_g"],
                          sc["anemoi_g_inv"], sc["edwards_a"], zhi, out.data_ptr())
    return out.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("n,factor", [(4, 16), (8, 16), (16, 6), (1024, 6), (16384, 6)])
def test_quotient_matches_reference_loop(gpu, n, factor):
    m = n * factor
    vecs = rand_fr_wire(56 * m, 1000 + n).reshape(56, m, 4)
    sc = _scalars(n)
    zhi = oc.z_h_inv(sc["k"][1], n, factor)
    got = _run(gpu, n, factor, vecs, sc, zhi)
    want = oc.t_quotient(n, factor, vecs, sc["alpha"], sc["beta"], sc["gamma"], sc["k"], sc["anemoi_g"],
                         sc["anemoi_g_inv"], sc["edwards_a"], zhi)
    assert np.array_equal(got, want)


def test_quotient_prover_like_values(gpu):
    """Selector-like inputs: boolean w_sel / q_ecc / qb / q_prk3, zero public input, many zero wires."""
    n, factor = 256, 6
    m = n * factor
    rng = np.random.default_rng(5)
    vecs = rand_fr_wire(56 * m, 77).reshape(56, m, 4)
    one = oc.fr_from_ints([1])[0]
    zero = np.zeros(4, dtype=np.uint64)
    for slot in (5, 6, 7, 25, 28, 55):                 # w_sel[0..2], qb, q_prk3, q_ecc
        bits = rng.integers(0, 2, m)
        vecs[slot] = np.where(bits[:, None] == 1, one[None, :], zero[None, :])
    vecs[8] = 0                                        # pi
    vecs[3][rng.integers(0, 2, m) == 1] = 0            # sparse w[3]
    sc = _scalars(9)
    zhi = oc.z_h_inv(sc["k"][1], n, factor)
    got = _run(gpu, n, factor, vecs, sc, zhi)
    want = oc.t_quotient(n, factor, vecs, sc["alpha"], sc["beta"], sc["gamma"], sc["k"], sc["anemoi_g"],
                         sc["anemoi_g_inv"], sc["edwards_a"], zhi)
    assert np.array_equal(got, want)


def test_z_h_inv_definition():
    """oracle_z_h_inv against the big-int definition (helpers.rs:242-252)."""
    n, factor = 16, 6
    k1 = 7
    zh = oc.fr_to_ints(oc.z_h_inv(oc.fr_from_ints([k1])[0], n, factor))
    g = opy.root_of_unity(n * factor)
    for i in range(factor):
        want = pow((pow(k1, n, opy.R) * pow(g, n * i, opy.R) - 1) % opy.R, -1, opy.R)
        assert zh[i] == want


def test_quotient_argument_errors(gpu):
    from uzkge_amd.errors import UzkgeError
    n, factor = 4, 16
    m = n * factor
    vecs = rand_fr_wire(56 * m, 3).reshape(56, m, 4)
    sc = _scalars(1)
    zhi = oc.z_h_inv(sc["k"][1], n, factor)
    dev = torch.from_numpy(vecs.view(np.int64)).to("cuda")
    out = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    ptrs = [dev[i].data_ptr() for i in range(56)]
    args = (sc["alpha"], sc["beta"], sc["gamma"], sc["k"], sc["anemoi_g"], sc["anemoi_g_inv"], sc["edwards_a"], zhi)
    with pytest.raises(UzkgeError):
        gpu.t_quotient_device(n, 17, ptrs, *args, out.data_ptr())           # factor > 16
    bad = list(ptrs); bad[40] = 0
    with pytest.raises(UzkgeError):
        gpu.t_quotient_device(n, factor, bad, *args, out.data_ptr())        # null vector
    with pytest.raises(UzkgeError):
        gpu.t_quotient_device(n, factor, ptrs, *args, ptrs[9])              # output aliases an input
