"""The pairing oracle (oracle/bn254_pairing.py) anchored on the reference's own parameter file, and -- through it -- the
CPU restatement of the KZG opening (`prove` / `verify`, uzkge/src/poly_commit/kzg_poly_commitment.rs:316-371).

`srs-padding.bin` ends with `public_parameter_group_2` = [H, [tau]H] (kzg_poly_commitment.rs:195-198).  The checks:
  * the file's first G2 element is the standard BN254 G2 generator and both lie on the twist;
  * the reference's own parameter test (`check_public_parameters_generation`, :440-470) on the reference's own file:
    e(srs[i+1], H) == e(srs[i], [tau]H) -- this fails for any error in the Miller loop, the Frobenius constants or the
    final exponentiation, so it pins the pairing restatement without arkworks;
  * bilinearity and non-degeneracy on small multiples;
  * an opening computed by the CPU oracle verifies, and a wrong evaluation does not.
No GPU."""
import os

import numpy as np
import pytest

import bn254_py as opy
import bn254_pairing as pr
from util import GOLDEN


@pytest.fixture(scope="module")
def params():
    blob = open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read()
    return opy.parse_srs_g1(blob), pr.parse_srs_g2(blob)


def test_g2_elements_of_the_reference_file(params):
    g1, g2 = params
    assert len(g2) == 2 and g2[0] == pr.G2_GEN
    assert pr.g2_is_on_curve(g2[0]) and pr.g2_is_on_curve(g2[1]) and g2[1] != g2[0]
    assert pr.g2_mul(g2[0], opy.R) is None and pr.g2_mul(g2[1], opy.R) is None      # both in the order-r subgroup


def test_bilinearity_and_non_degeneracy():
    e = pr.pairing(opy.G1_GEN, pr.G2_GEN)
    assert e != pr.F12_ONE
    assert pr.f12_pow(e, opy.R) == pr.F12_ONE
    a, b = 0x1234567, 0x89ABC
    assert pr.pairing(opy.g1_mul(opy.G1_GEN, a), pr.g2_mul(pr.G2_GEN, b)) == pr.f12_pow(e, a * b)
    assert pr.pairing(None, pr.G2_GEN) == pr.F12_ONE and pr.pairing(opy.G1_GEN, None) == pr.F12_ONE


@pytest.mark.parametrize("i", [0, 1, 1024, 2049, 2051, 2054, 2057])
def test_reference_parameters_are_consecutive_powers(params, i):
    """kzg_poly_commitment.rs:449-458 on the reference's file; 2051 / 2054 / 2057 are the first of the three padding powers
    tau^N, tau^(N+1), tau^(N+2) for N = 4096 / 8192 / 16384 (gen_params/mod.rs:163-173)."""
    g1, g2 = params
    assert pr.pairing_product_is_one([(g1[i + 1], g2[0]), (opy.g1_neg(g1[i]), g2[1])])
    assert not pr.pairing_product_is_one([(g1[i + 1], g2[0]), (opy.g1_neg(g1[i + 1]), g2[1])])


def test_oracle_opening_verifies(params):
    """prove (:316-342): q = (f - f(z)) / (X - z), proof = commit(q); verify (:344-371)."""
    g1, g2 = params
    rng = np.random.default_rng(3)
    f = [int(x) for x in rng.integers(1, 1 << 62, 24)]
    z = 0xABCDEF0123456789
    v = opy.poly_eval(f, z)
    # synthetic division by X - z
    q = [0] * (len(f) - 1)
    acc = 0
    for k in range(len(f) - 1, 0, -1):
        acc = (f[k] + acc * z) % opy.R
        q[k - 1] = acc
    assert (f[0] + acc * z) % opy.R == v
    cm = opy.msm_naive(g1[: len(f)], f)
    proof = opy.msm_naive(g1[: len(q)], q)
    assert pr.kzg_verify(g1[0], g2[0], g2[1], cm, z, v, proof, opy.g1_mul, opy.g1_add)
    assert not pr.kzg_verify(g1[0], g2[0], g2[1], cm, z, (v + 1) % opy.R, proof, opy.g1_mul, opy.g1_add)
    assert not pr.kzg_verify(g1[0], g2[0], g2[1], cm, z + 1, v, proof, opy.g1_mul, opy.g1_add)
