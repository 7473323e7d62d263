"""Pins the CPU oracle (oracle/bn254_py.py, oracle/bn254_oracle.c) against the reference's own
fixture files -- the parameter blobs uzkge embeds (uzkge/src/gen_params/mod.rs:20-65), copied as
data under tests/golden/ -- and cross-checks the two oracle implementations against each other.

Known answers:
  * srs-padding.bin[0] == G = (1, 2)
  * sum_i L_i == G                       over lagrange-srs-{4096,8192,16384}.bin (8192 = zmatchmaking's circuit size,
                                         matchmaking/src/build_cs.rs:68-99)
  * sum_i omega^i L_i == [tau]G          == srs-padding.bin[1]   (pins omega and natural order)
  * MSM(lagrange_srs, NTT(c)) == MSM(monomial_srs, c)  for deg c <= 2050 (NTT + MSM together)
"""
import random

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import load_srs, rand_fr


@pytest.fixture(scope="module")
def mono():
    return load_srs("srs-padding.bin")


@pytest.mark.parametrize("name,n", [("lagrange-srs-4096.bin", 4096), ("lagrange-srs-8192.bin", 8192),
                                    ("lagrange-srs-16384.bin", 16384)])
def test_lagrange_identities(name, n, mono):
    wire, pts = load_srs(name)
    _, mono_pts = mono
    assert len(pts) == n and mono_pts[0] == opy.G1_GEN
    assert all(opy.g1_is_on_curve(p) for p in pts[:64])
    ones = oc.fr_from_ints([1] * n)
    assert oc.jac_to_affine_ints(oc.msm_pippenger(wire, ones, 0, 8)) == opy.G1_GEN
    w = opy.root_of_unity(n)
    ws, x = [], 1
    for _ in range(n):
        ws.append(x)
        x = x * w % opy.R
    assert oc.jac_to_affine_ints(oc.msm_pippenger(wire, oc.fr_from_ints(ws), 0, 8)) == mono_pts[1]


def test_ntt_and_msm_against_reference_srs(mono):
    wire, _ = load_srs("lagrange-srs-4096.bin")
    mono_wire, _ = mono
    c = rand_fr(2051, 1)
    ev = oc.ntt(oc.fr_from_ints(c + [0] * (4096 - 2051)), threads=4)
    lhs = oc.jac_to_affine_ints(oc.msm_pippenger(wire, ev, 0, 8))
    rhs = oc.jac_to_affine_ints(oc.msm_pippenger(mono_wire[:2051], oc.fr_from_ints(c), 0, 8))
    assert lhs == rhs and lhs is not None


def test_srs_file_format(golden_dir):
    """u32 len_g1 | u32 len_g2 | len_g1 x 64 B | len_g2 x 128 B (kzg_poly_commitment.rs:206-264)."""
    import os, struct
    for name, n in (("lagrange-srs-4096.bin", 4096), ("lagrange-srs-8192.bin", 8192), ("lagrange-srs-16384.bin", 16384)):
        data = open(os.path.join(golden_dir, name), "rb").read()
        l1, l2 = struct.unpack_from("<II", data, 0)
        assert (l1, l2) == (n, 0) and len(data) == 8 + 64 * n
    data = open(os.path.join(golden_dir, "srs-padding.bin"), "rb").read()
    l1, l2 = struct.unpack_from("<II", data, 0)
    assert (l1, l2) == (2060, 2) and len(data) == 8 + 64 * 2060 + 128 * 2


def test_field_kats_c_vs_python():
    rng = random.Random(5)
    edge = [0, 1, 2, opy.R - 1, opy.R - 2, opy.FR_R, (1 << 253) % opy.R]
    vals = edge + [rng.randrange(opy.R) for _ in range(100)]
    for a in vals:
        for b in vals[:12]:
            am = np.array(opy.int_to_limbs(a), dtype=np.uint64)
            bm = np.array(opy.int_to_limbs(b), dtype=np.uint64)
            assert opy.limbs_to_int(oc.fr_mul(am, bm)) == opy.mont_mul(a, b, opy.R) == opy.mont_mul_cios32(a, b, opy.R)
            assert opy.limbs_to_int(oc.fr_add(am, bm)) == (a + b) % opy.R
            assert opy.limbs_to_int(oc.fr_sub(am, bm)) == (a - b) % opy.R
            aq, bq = a % opy.P, b % opy.P
            assert opy.limbs_to_int(oc.fq_mul(np.array(opy.int_to_limbs(aq), dtype=np.uint64),
                                              np.array(opy.int_to_limbs(bq), dtype=np.uint64))) == opy.mont_mul(aq, bq, opy.P)
    x = oc.fr_from_ints([12345])[0]
    assert oc.fr_to_ints(oc.fr_mul(x, oc.fr_inv(x))) == [1]


def test_group_law_edge_cases():
    g = oc.points_from_affine([opy.G1_GEN])[0]
    inf = np.zeros(12, dtype=np.uint64)
    gj = np.concatenate([g, oc.fr_from_ints([1], mod=opy.P)[0]])
    assert oc.jac_to_affine_ints(oc.g1_add(gj, inf)) == opy.G1_GEN
    assert oc.jac_to_affine_ints(oc.g1_add(inf, gj)) == opy.G1_GEN
    assert oc.jac_to_affine_ints(oc.g1_add(gj, gj)) == opy.g1_add(opy.G1_GEN, opy.G1_GEN)       # P + P
    neg = oc.points_from_affine([opy.g1_neg(opy.G1_GEN)])[0]
    nj = np.concatenate([neg, oc.fr_from_ints([1], mod=opy.P)[0]])
    assert oc.jac_to_affine_ints(oc.g1_add(gj, nj)) is None                                     # P + (-P)
    k = 0xDEADBEEFCAFEBABE1234
    assert oc.jac_to_affine_ints(oc.g1_mul(g, oc.fr_from_ints([k])[0])) == opy.g1_mul(opy.G1_GEN, k)
    assert oc.jac_to_affine_ints(oc.g1_mul(g, oc.fr_from_ints([opy.R - 1])[0])) == opy.g1_neg(opy.G1_GEN)


@pytest.mark.parametrize("n", [1, 2, 33, 256])
def test_msm_c_pippenger_vs_naive_vs_python(n):
    wire, pts = load_srs("lagrange-srs-4096.bin")
    ints = rand_fr(n, 100 + n)
    ints[0] = 0
    if n > 2:
        ints[1], ints[2] = 1, opy.R - 1
    s = oc.fr_from_ints(ints)
    a = oc.jac_to_affine_ints(oc.msm_naive(wire[:n], s))
    b = oc.jac_to_affine_ints(oc.msm_pippenger(wire[:n], s, 0, 2))
    assert a == b
    if n <= 33:
        assert a == opy.msm_naive(pts[:n], ints) == opy.msm_pippenger(pts[:n], ints, c=5)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6, 16, 32, 48, 96, 1024, 3072])
def test_ntt_c_vs_python_definition(n):
    v = rand_fr(n, 200 + n)
    got = oc.fr_to_ints(oc.ntt(oc.fr_from_ints(v)))
    assert got == opy.ntt(v, n)
    if n <= 96:
        assert got == opy.dft_naive(v, n)
        w = opy.root_of_unity(n)
        assert got == [opy.poly_eval(v, pow(w, i, opy.R)) for i in range(n)]      # reference's check_fft
    assert oc.fr_to_ints(oc.ntt(oc.ntt(oc.fr_from_ints(v)), inverse=True)) == v     # ifft(fft) == id
    assert oc.fr_to_ints(oc.ntt(oc.fr_from_ints(v), inverse=True)) == opy.ntt(v, n, inverse=True)


def test_coset_c_vs_python():
    n, k = 48, 7
    v = rand_fr(n, 9)
    kw = oc.fr_from_ints([k])[0]
    assert oc.fr_to_ints(oc.ntt(oc.mul_var(oc.fr_from_ints(v), kw))) == opy.coset_ntt(v, n, k)
    assert opy.coset_intt(opy.coset_ntt(v, n, k), n, pow(k, -1, opy.R)) == v


def test_domain_support_and_roots():
    for n in (1, 2, 3, 4, 6, 48, 1 << 14, 98304, 1 << 28):
        assert opy.domain_supported(n) and oc.lib.oracle_domain_supported(n)
    for n in (0, 5, 9, 7, 10, 1 << 29):
        assert not opy.domain_supported(n) and not oc.lib.oracle_domain_supported(n)
    for n in (2, 3, 48, 1 << 14, 98304, 1 << 22):
        assert oc.fr_to_ints(oc.root_of_unity(n)) == [opy.root_of_unity(n)]
    # BASELINE.md section 4 quotes omega_{2^22}
    assert opy.root_of_unity(1 << 22) == 12143866164239048021030917283424216263377309185099704096317235600302831912062
