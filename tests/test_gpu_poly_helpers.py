"""Device polynomial helpers next to the hot path against CPU restatements of
FpPolynomial::eval (field_polynomial.rs:198-209) and z_poly (plonk/helpers.rs:160-220).
Parity here is restatement-vs-restatement plus protocol properties: the reference holds no fixture
for these intermediate values (noted as "parity unpinned" for this row in DESIGN.md)."""
import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import rand_fr, rand_fr_wire

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,batch", [(1, 1), (3, 2), (17, 5), (1024, 2), (1025, 2), (4096, 3), (16384, 9), (16387, 20), (65536, 3), (65537, 2),
                                     (98304, 6), (100000, 2), (262144, 1), (262145, 2)])
def test_poly_eval_batch(gpu, n, batch):
    """Both evaluation paths -- the one-launch kernel for <= 256 blocks of 1024 coefficients (n <= 2^18) and the table-driven one
    beyond -- against the oracle; repeated calls (the one-launch kernel's arrival counters must come back to zero)."""
    c = rand_fr_wire(n * batch, 10 + n).reshape(batch, n, 4)
    x = rand_fr_wire(1, 77)[0]
    want = [oc.poly_eval(c[b], x) for b in range(batch)]
    for rep in range(3):
        got = gpu.poly_eval_batch(c, x)
        for b in range(batch):
            assert np.array_equal(got[b], want[b]), (n, b, rep)


def test_poly_eval_special_points(gpu):
    c = rand_fr_wire(1000, 5).reshape(1, 1000, 4)
    zero = np.zeros(4, dtype=np.uint64)
    one = oc.fr_from_ints([1])[0]
    assert np.array_equal(gpu.poly_eval_batch(c, zero)[0], c[0, 0])            # p(0) = c_0
    total = c[0, 0]
    for j in range(1, 1000):
        total = oc.fr_add(total, c[0, j])
    assert np.array_equal(gpu.poly_eval_batch(c, one)[0], total)               # p(1) = sum c_j


def _domain(n):
    w = opy.root_of_unity(n)
    g, x = [], 1
    for _ in range(n):
        g.append(x)
        x = x * w % opy.R
    return oc.fr_from_ints(g)


@pytest.mark.parametrize("n,n_wires", [(2, 3), (8, 5), (1024, 5), (16384, 5), (5000, 3)])
def test_z_poly_matches_restatement(gpu, n, n_wires):
    rng = np.random.default_rng(n)
    w = rand_fr_wire(n * n_wires, 3 + n).reshape(n_wires, n, 4)
    perm = rng.integers(0, n * n_wires, size=(n_wires, n), dtype=np.uint32)
    group = _domain(n) if (n & (n - 1)) == 0 else rand_fr_wire(n, 4)
    k = oc.fr_from_ints([1, 7, 13, 17, 23][:n_wires])
    beta, gamma = rand_fr_wire(2, 99)
    got = gpu.z_poly(w, perm, group, k, beta, gamma)
    assert np.array_equal(got, oc.z_poly(w, perm, group, k, beta, gamma))
    assert oc.fr_to_ints(got[:1]) == [1]


def test_z_poly_device_entry_point(gpu):
    """uzk_z_poly_device on device-resident inputs gives the same bytes as the host-pointer entry point."""
    import torch
    n, n_wires = 5000, 5
    rng = np.random.default_rng(7)
    w = rand_fr_wire(n * n_wires, 31).reshape(n_wires, n, 4)
    perm = rng.integers(0, n * n_wires, size=(n_wires, n), dtype=np.uint32)
    group = rand_fr_wire(n, 32)
    k = oc.fr_from_ints([1, 7, 13, 17, 23])
    beta, gamma = rand_fr_wire(2, 33)
    dw = torch.from_numpy(w.view(np.int64)).to("cuda")
    dp = torch.from_numpy(perm.view(np.int32)).to("cuda")
    dg = torch.from_numpy(group.view(np.int64)).to("cuda")
    dz = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    gpu.z_poly_device(dw.data_ptr(), dp.data_ptr(), dg.data_ptr(), k, beta, gamma, n, n_wires, dz.data_ptr())
    gpu.sync()
    assert np.array_equal(dz.cpu().numpy().view(np.uint64), oc.z_poly(w, perm, group, k, beta, gamma))


def test_z_poly_closes_for_a_satisfied_permutation(gpu):
    """Protocol property: when the wire values respect the copy constraints (w at position p equals
    w at perm[p]), the grand product over the WHOLE domain is 1, i.e. z[n-1] times the last row's
    numerator/denominator equals 1 -- the identity the verifier relies on."""
    n, n_wires = 256, 5
    rng = np.random.default_rng(1)
    total = n * n_wires
    perm_flat = rng.permutation(total).astype(np.uint32)           # a genuine permutation of wire slots
    # values constant on every cycle of the permutation
    vals = [None] * total
    ints = rand_fr(total, 11)
    for start in range(total):
        if vals[start] is None:
            p = start
            while vals[p] is None:
                vals[p] = ints[start]
                p = int(perm_flat[p])
    w = oc.fr_from_ints(vals).reshape(n_wires, n, 4)
    perm = perm_flat.reshape(n_wires, n)
    group = _domain(n)
    kints = [1, 7, 13, 17, 23]
    k = oc.fr_from_ints(kints)
    b_i, g_i = rand_fr(2, 5)
    z = gpu.z_poly(w, perm, group, k, oc.fr_from_ints([b_i])[0], oc.fr_from_ints([g_i])[0])
    zl = oc.fr_to_ints(z[n - 1:n])[0]
    gi = oc.fr_to_ints(group)
    i = n - 1
    num = den = 1
    for j in range(n_wires):
        f = vals[j * n + i]
        num = num * (f + g_i + b_i * kints[j] * gi[i]) % opy.R
        pv = int(perm[j, i])
        den = den * (f + g_i + b_i * kints[pv // n] * gi[pv % n]) % opy.R
    assert zl * num % opy.R == den % opy.R
