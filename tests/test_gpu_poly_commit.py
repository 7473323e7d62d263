"""The reference's own unit tests for the boundary, restated against the host mirror
(uzkge_amd/poly_commit.py) running on the GPU:
  test_commit, test_homomorphic_poly_com_elem   uzkge/src/poly_commit/kzg_poly_commitment.rs:483-548
  test_fft                                      uzkge/src/poly_commit/field_polynomial.rs:632-719
"""
import os

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import GOLDEN, affine_of, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcs(gpu):
    from uzkge_amd.poly_commit import KZGCommitmentSchemeBN254
    data = open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read()
    s = KZGCommitmentSchemeBN254.from_unchecked_bytes(data)
    yield s
    s.release()


def test_from_unchecked_bytes_matches_oracle_parser(pcs):
    data = open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read()
    pts = opy.parse_srs_g1(data)
    assert pcs.max_degree() == len(pts) - 1 == 2059
    assert np.array_equal(pcs.public_parameter_group_1, oc.points_from_affine(pts))


def test_commit(pcs):
    """test_commit: commit(poly) == sum_i coef_i * SRS_i by a naive scalar-mul loop."""
    from uzkge_amd.poly_commit import FpPolynomial
    coefs = [1, 2, 3] + rand_fr(11, 3)
    poly = FpPolynomial.from_ints(coefs)
    got = affine_of(pcs.commit(poly))
    want = None
    pts = opy.parse_srs_g1(open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read())
    for c, p in zip(coefs, pts):
        want = opy.g1_add(want, opy.g1_mul(p, c))
    assert got == want


def test_homomorphic_poly_com_elem(pcs):
    from uzkge_amd.poly_commit import FpPolynomial
    a, b = rand_fr(20, 1), rand_fr(20, 2)
    ca = pcs.commit(FpPolynomial.from_ints(a))
    cb = pcs.commit(FpPolynomial.from_ints(b))
    cs = pcs.commit(FpPolynomial.from_ints([(x + y) % opy.R for x, y in zip(a, b)]))
    assert affine_of(cs) == opy.g1_add(affine_of(ca), affine_of(cb))
    c5 = pcs.commit(FpPolynomial.from_ints([5 * x % opy.R for x in a]))
    assert affine_of(c5) == opy.g1_mul(affine_of(ca), 5)


def test_commit_degree_error_and_zero_poly(pcs):
    from uzkge_amd import UzkgeError
    from uzkge_amd.poly_commit import FpPolynomial
    with pytest.raises(UzkgeError) as e:
        pcs.commit(FpPolynomial.from_ints(rand_fr(2061, 5)))
    assert e.value.kind == "DegreeError"
    zero = FpPolynomial.from_ints([0, 0, 0, 0])
    assert zero.coefs.shape[0] == 1 and zero.degree() == 0      # trimmed to a single zero
    assert affine_of(pcs.commit(zero)) is None


def test_apply_blind_factors(pcs):
    """C' = C + sum b_i (SRS[i] - SRS[z + i])  (kzg_poly_commitment.rs:299-313)."""
    from uzkge_amd.poly_commit import FpPolynomial, fr_from_int
    pts = opy.parse_srs_g1(open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read())
    poly = FpPolynomial.from_ints(rand_fr(16, 9))
    c = pcs.commit(poly)
    blinds = [11, 22, opy.R - 3]
    z = 2051
    got = affine_of(pcs.apply_blind_factors(c, np.stack([fr_from_int(x) for x in blinds]), z))
    want = affine_of(c)
    for i, bl in enumerate(blinds):
        want = opy.g1_add(want, opy.g1_mul(pts[i], bl))
        want = opy.g1_add(want, opy.g1_mul(pts[z + i], (-bl) % opy.R))
    assert got == want


def _check_fft(poly, n, fft):
    w = opy.root_of_unity(n)
    vals = oc.fr_to_ints(fft)
    return all(vals[i] == poly.eval(pow(w, i, opy.R)) for i in range(n))


def test_fft(gpu):
    """The literal sequence of the reference's test_fft."""
    from uzkge_amd.poly_commit import FpPolynomial
    one, zero = 1, 0
    p = FpPolynomial.from_ints([one]);            assert _check_fft(p, 1, p.fft(1))
    p = FpPolynomial.from_ints([one, one]);       assert _check_fft(p, 2, p.fft(2))
    p = FpPolynomial.from_ints([one, zero]);      assert _check_fft(p, 2, p.fft(2))
    p = FpPolynomial.from_ints([zero, one]);      assert _check_fft(p, 2, p.fft(2))
    p = FpPolynomial.from_ints([zero, one, one])
    f = p.fft(3)
    assert _check_fft(p, 3, f)
    assert FpPolynomial.ifft_with_domain(FpPolynomial.quotient_evaluation_domain(3), f) == p
    for n, dom in ((16, FpPolynomial.evaluation_domain), (32, FpPolynomial.evaluation_domain),
                   (3, FpPolynomial.quotient_evaluation_domain), (48, FpPolynomial.quotient_evaluation_domain)):
        p = FpPolynomial.from_ints(rand_fr(n, 100 + n))
        d = dom(n)
        assert FpPolynomial.ifft_with_domain(d, p.fft_with_domain(d)) == p


def test_coset_fft_roundtrip_on_quotient_domain(gpu):
    """t_poly's shape: n + 3 coefficients on the 6n coset domain and back (helpers.rs:256-266,673)."""
    from uzkge_amd.poly_commit import FpPolynomial, fr_from_int
    n = 64
    p = FpPolynomial.from_ints(rand_fr(n + 3, 8))
    k = 5
    d = FpPolynomial.quotient_evaluation_domain(6 * n)
    ev = p.coset_fft_with_domain(d, fr_from_int(k))
    w = opy.root_of_unity(6 * n)
    assert oc.fr_to_ints(ev[:5]) == [p.eval(k * pow(w, i, opy.R) % opy.R) for i in range(5)]
    back = FpPolynomial.coset_ifft_with_domain(d, ev, fr_from_int(pow(k, -1, opy.R)))
    assert back == p
