"""MSM parity: HIP Pippenger (through the C ABI) vs the CPU oracle, compared after affine
normalisation (the Jacobian representative is not unique).  Mirrors the reference's
`test_commit` (uzkge/src/poly_commit/kzg_poly_commitment.rs:526-548: commit == naive sum of
coef_i * SRS_i) and `test_homomorphic_poly_com_elem` (:483-514)."""
import numpy as np
import pytest
import torch

import bn254_py as opy
import oracle_c as oc
from util import affine_of, load_srs, rand_fr, rand_fr_wire

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lagrange(gpu):
    wire, pts = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)
    yield srs, wire, pts
    srs.release()


def _check(gpu, srs, wire, scalars, offset=0):
    got = affine_of(gpu.msm(srs, scalars, offset=offset))
    n = scalars.shape[0]
    want = affine_of(oc.msm_pippenger(wire[offset:offset + n], scalars, 0, 8))
    assert got == want


@pytest.mark.parametrize("n", [1, 2, 3, 33, 64, 255, 1024, 4096])
def test_uniform_scalars(gpu, lagrange, n):
    srs, wire, _ = lagrange
    _check(gpu, srs, wire, rand_fr_wire(n, 500 + n))


def test_matches_naive_definition(gpu, lagrange):
    """test_commit's definition: sum of per-term scalar multiplications."""
    srs, wire, _ = lagrange
    s = rand_fr_wire(40, 9)
    assert affine_of(gpu.msm(srs, s)) == affine_of(oc.msm_naive(wire[:40], s))


@pytest.mark.parametrize("kind", ["zeros", "ones", "minus_one", "boolean", "small", "prover_mix", "same"])
def test_special_scalar_sets(gpu, lagrange, kind):
    """Value classes the real prover produces: padding zeros, boolean wires,
    +1 / -1 = r-1 selectors, small values, and a mixture; plus all-equal scalars (one bucket)."""
    srs, wire, _ = lagrange
    n = 4096
    rng = np.random.default_rng(77)
    if kind == "zeros":
        ints = [0] * n
    elif kind == "ones":
        ints = [1] * n
    elif kind == "minus_one":
        ints = [opy.R - 1] * n
    elif kind == "boolean":
        ints = [int(v) for v in rng.integers(0, 2, n)]
    elif kind == "small":
        ints = [int(v) for v in rng.integers(0, 1 << 16, n)]
    elif kind == "same":
        ints = [0x1234567890ABCDEF1234567890ABCDEF] * n
    else:
        u = rand_fr(n, 5)
        cls = rng.integers(0, 10, n)
        ints = [0 if c < 5 else 1 if c < 7 else opy.R - 1 if c < 8 else int(rng.integers(0, 1 << 16)) if c < 9 else u[i]
                for i, c in enumerate(cls)]
    _check(gpu, srs, wire, oc.fr_from_ints(ints))


def test_all_zero_vector_arrives_as_single_zero(gpu, lagrange):
    """from_coefs trims trailing zeros, so an all-zero evaluation vector reaches commit as n = 1
    with scalar 0: must return infinity (z == 0), not an error."""
    srs, _, _ = lagrange
    out = gpu.msm(srs, oc.fr_from_ints([0]))
    assert affine_of(out) is None
    assert not out[8:12].any()
    assert affine_of(gpu.msm(srs, np.zeros((0, 4), dtype=np.uint64))) is None


def test_degree_error(gpu, lagrange):
    """commit: degree + 1 > SRS length -> DegreeError (kzg_poly_commitment.rs:283-285)."""
    from uzkge_amd import UzkgeError
    srs, _, _ = lagrange
    with pytest.raises(UzkgeError) as e:
        gpu.msm(srs, rand_fr_wire(4097, 1))
    assert e.value.kind == "DegreeError"
    with pytest.raises(UzkgeError):
        gpu.msm(srs, rand_fr_wire(10, 1), offset=4090)


def test_offset_window(gpu, lagrange):
    srs, wire, _ = lagrange
    _check(gpu, srs, wire, rand_fr_wire(100, 3), offset=1000)


def test_infinity_bases_are_identity(gpu):
    """The monomial SRS holds identity points between 2051 and n (gen_params/mod.rs:160-173):
    affine (0,0) must act as the identity."""
    wire, _ = load_srs("lagrange-srs-4096.bin")
    pts = wire[:64].copy()
    pts[5] = 0
    pts[17] = 0
    s = rand_fr_wire(64, 21)
    got = affine_of(gpu.msm_raw(pts, s))
    assert got == affine_of(oc.msm_pippenger(pts, s, 0, 1))


def test_repeated_and_opposite_points(gpu):
    """P + P (doubling branch) and P + (-P) (-> infinity branch) inside one bucket."""
    wire, _ = load_srs("lagrange-srs-4096.bin")
    p = wire[7]
    pts = np.stack([p, p, p, p])
    one = oc.fr_from_ints([1, 1, 1, opy.R - 1])
    assert affine_of(gpu.msm_raw(pts, one)) == affine_of(oc.msm_naive(pts, one))
    two = oc.fr_from_ints([5, opy.R - 5])
    assert affine_of(gpu.msm_raw(pts[:2], two)) is None


def test_homomorphism(gpu, lagrange):
    """commit(p + q) == commit(p) + commit(q); commit(5 p) == 5 commit(p)."""
    srs, wire, _ = lagrange
    n = 512
    a, b = rand_fr(n, 31), rand_fr(n, 32)
    ca = gpu.msm(srs, oc.fr_from_ints(a))
    cb = gpu.msm(srs, oc.fr_from_ints(b))
    cab = gpu.msm(srs, oc.fr_from_ints([(x + y) % opy.R for x, y in zip(a, b)]))
    assert affine_of(gpu.g1_fold(np.stack([ca, cb]))) == affine_of(cab)
    c5 = gpu.msm(srs, oc.fr_from_ints([5 * x % opy.R for x in a]))
    assert affine_of(c5) == opy.g1_mul(affine_of(ca), 5)


def test_lagrange_srs_identities(gpu, lagrange):
    """Known answers from the reference's own parameter files: sum_i L_i == G and
    sum_i w^i L_i == [tau]G (srs-padding.bin[1]) -- pins omega and ordering on the GPU path."""
    srs, _, _ = lagrange
    _, mono = load_srs("srs-padding.bin")
    assert affine_of(gpu.msm(srs, oc.fr_from_ints([1] * 4096))) == opy.G1_GEN
    w = opy.root_of_unity(4096)
    ws = oc.fr_from_ints([pow(w, i, opy.R) for i in range(4096)])
    assert affine_of(gpu.msm(srs, ws)) == mono[1]


def test_ntt_then_msm_equals_monomial_commit(gpu, lagrange):
    """MSM(lagrange_srs, NTT(c)) == MSM(monomial_srs, c): NTT and MSM together against
    reference data (the shape of every commit in prover_with_lagrange, prover.rs:132-149)."""
    srs, _, _ = lagrange
    mono_wire, _ = load_srs("srs-padding.bin")
    c = rand_fr(2051, 41)
    ev = gpu.ntt(oc.fr_from_ints(c + [0] * (4096 - 2051)))
    lhs = affine_of(gpu.msm(srs, ev))
    rhs = affine_of(gpu.msm_raw(mono_wire[:2051], oc.fr_from_ints(c)))
    assert lhs == rhs == affine_of(oc.msm_pippenger(mono_wire[:2051], oc.fr_from_ints(c), 0, 8))


def test_prover_size_16384(gpu):
    """The real workload: n = 2^14 over lagrange-srs-16384.bin."""
    wire, _ = load_srs("lagrange-srs-16384.bin")
    srs = gpu.Srs.from_host(wire)
    try:
        s = rand_fr_wire(16384, 14)
        assert affine_of(gpu.msm(srs, s)) == affine_of(oc.msm_pippenger(wire, s, 0, 8))
    finally:
        srs.release()


@pytest.mark.parametrize("c", [6, 9, 13, 16])
def test_window_bits_do_not_change_result(gpu, lagrange, c):
    srs, wire, _ = lagrange
    s = rand_fr_wire(3000, 60 + c)
    gpu.set_msm_window_bits(c)
    try:
        _check(gpu, srs, wire, s)
    finally:
        gpu.set_msm_window_bits(0)


# ---- precomputed-window mode (uzk_srs_precompute): same results, one shared bucket set ----------
@pytest.mark.parametrize("c", [0, 6, 9, 13, 16, 20, 22])
def test_precomputed_mode_matches_oracle(gpu, c):
    """c <= 10: one radix pass, 11..19: two, >= 20: three."""
    wire, _ = load_srs("lagrange-srs-4096.bin")
    pts = wire.copy()
    pts[7] = 0                      # an infinity base must stay the identity in every window table
    srs = gpu.Srs.from_host(pts)
    try:
        srs.precompute(c)
        for n, seed in ((1, 1), (2, 2), (33, 3), (1000, 4), (4096, 5)):
            s = rand_fr_wire(n, 900 + seed + c)
            assert affine_of(gpu.msm(srs, s)) == affine_of(oc.msm_pippenger(pts[:n], s, 0, 8)), (c, n)
        s = rand_fr_wire(100, 77)
        assert affine_of(gpu.msm(srs, s, offset=1000)) == affine_of(oc.msm_pippenger(pts[1000:1100], s, 0, 2))
        ints = [0, 1, opy.R - 1, 2, opy.R - 2] * 200
        sw = oc.fr_from_ints(ints)
        assert affine_of(gpu.msm(srs, sw)) == affine_of(oc.msm_pippenger(pts[:1000], sw, 0, 4))
        assert affine_of(gpu.msm(srs, oc.fr_from_ints([0]))) is None
        # switching the table off gives the same commitment
        gpu.tune("msm_no_precompute", 1)
        try:
            s = rand_fr_wire(4096, 123)
            a = affine_of(gpu.msm(srs, s))
        finally:
            gpu.tune("msm_no_precompute", 0)
        assert a == affine_of(gpu.msm(srs, s))
    finally:
        srs.release()


def test_switches_do_not_change_the_result(gpu):
    """uzk_tune's remaining switches choose a pipeline, never a result: without the window table, with the generic last sort pass,
    with a forced segment-sort instantiation."""
    n = 1 << 20
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 3)
    gpu.synth_scalars(sc.data_ptr(), n, 4)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        ref = gpu.g1_to_affine(gpu.msm_device(srs, sc.data_ptr(), n))
        for key, val, back in (("msm_seg_sort", 0, 1), ("msm_seg_sort", 12, 1), ("msm_no_precompute", 1, 0)):
            gpu.tune(key, val)
            try:
                assert np.array_equal(gpu.g1_to_affine(gpu.msm_device(srs, sc.data_ptr(), n)), ref), (key, val)
            finally:
                gpu.tune(key, back)
        with pytest.raises(Exception):
            gpu.tune("msm_acc_variant", 1)          # the experiment switches of earlier rounds are gone from the ABI
    finally:
        srs.release()


@pytest.mark.parametrize("n,stream_log", [(4096, 10), (5000, 10), (16384, 12), (100000, 14), (100000, 0)])
def test_streamed_host_scalar_msm(gpu, n, stream_log):
    """uzk_msm_g1 on host scalars streams large general-mode MSMs in point chunks that share one bucket set (the chunk's
    bucket sums are added onto the previous ones, one reduction at the end).  With the thresholds lowered the same code runs
    at sizes the oracle checks: uniform chunks, a ragged last chunk, and the graded default schedule."""
    wire, _ = load_srs("lagrange-srs-16384.bin")
    pts = np.concatenate([wire] * ((n + 16383) // 16384))[:n]
    srs = gpu.Srs.from_host(pts)
    s = rand_fr_wire(n, 4200 + n)
    s[::7] = 0
    s[1::11] = oc.fr_from_ints([opy.R - 1])[0]
    try:
        gpu.tune("msm_stream_min_log", 12)
        gpu.tune("msm_stream_log", stream_log)
        got = affine_of(gpu.msm(srs, s))
        gpu.tune("msm_stream_log", -1)                       # upload, then one MSM
        plain = affine_of(gpu.msm(srs, s))
        assert got == plain == affine_of(oc.msm_pippenger(pts, s, 0, 4))
    finally:
        gpu.tune("msm_stream_min_log", 22)
        gpu.tune("msm_stream_log", 0)
        srs.release()
