"""Batched entry points: `batch` independent problems in one launch sequence must equal the same
problems solved one at a time (and the oracle)."""
import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,batch", [(8, 3), (1024, 8), (4096, 5), (1 << 14, 8), (48, 4), (3 << 12, 3), (98304, 2), (1 << 16, 2)])
@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_batch_equals_single(gpu, n, batch, inverse):
    x = rand_fr_wire(n * batch, 70 + n + batch).reshape(batch, n, 4)
    got = gpu.ntt_batch(x, inverse=inverse)
    for b in range(batch):
        assert np.array_equal(got[b], oc.ntt(x[b], inverse=inverse, threads=8)), (n, b)


def test_ntt_batch_coset_roundtrip(gpu):
    n, batch, k = 3 << 10, 4, 11
    kw = oc.fr_from_ints([k])[0]
    kinv = oc.fr_from_ints([pow(k, -1, opy.R)])[0]
    x = rand_fr_wire(n * batch, 5).reshape(batch, n, 4)
    f = gpu.ntt_batch(x, coset_shift=kw)
    for b in range(batch):
        assert np.array_equal(f[b], oc.ntt(oc.mul_var(x[b], kw), threads=4))
    assert np.array_equal(gpu.ntt_batch(f, inverse=True, coset_shift=kinv), x)


@pytest.mark.parametrize("n,batch", [(1, 4), (33, 3), (1000, 8), (4096, 8)])
@pytest.mark.parametrize("pre_c", [-1, 0, 13])
def test_msm_batch_equals_single(gpu, n, batch, pre_c):
    """The prover's shape: several polynomials committed against the same (Lagrange) SRS; -1 = general mode."""
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)
    try:
        if pre_c >= 0:
            srs.precompute(pre_c)
        s = rand_fr_wire(n * batch, 300 + n + batch).reshape(batch, n, 4).copy()
        s[0, : min(n, 5)] = 0                                    # zero-padding / trimmed polynomials
        if batch > 1:
            s[1] = oc.fr_from_ints([1] * n)                      # boolean-like vector
        got = gpu.msm_batch(srs, s)
        for b in range(batch):
            want = affine_of(oc.msm_pippenger(wire[:n], s[b], 0, 4))
            assert affine_of(got[b]) == want, (n, b)
            assert affine_of(gpu.msm(srs, s[b])) == want
    finally:
        srs.release()


def test_msm_batch_prover_shape(gpu):
    """8 commits of 2^14 evaluations over lagrange-srs-16384.bin in one call."""
    wire, _ = load_srs("lagrange-srs-16384.bin")
    srs = gpu.Srs.from_host(wire)
    try:
        s = rand_fr_wire(8 << 14, 1414).reshape(8, 1 << 14, 4)
        got = gpu.msm_batch(srs, s)
        for b in (0, 3, 7):
            assert affine_of(got[b]) == affine_of(oc.msm_pippenger(wire, s[b], 0, 8))
    finally:
        srs.release()
