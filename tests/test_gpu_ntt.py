"""NTT parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact.
Mirrors the reference's own `test_fft` (uzkge/src/poly_commit/field_polynomial.rs:632-719):
fft[i] == eval(group_gen^i) in natural order; ifft(fft(p)) == p at 16, 32, 3, 48."""
import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import rand_fr, rand_fr_wire

pytestmark = pytest.mark.gpu

SIZES_POW2 = [1, 2, 4, 8, 16, 32, 64, 512, 1024, 2048, 4096, 8192, 1 << 14, 1 << 15, 1 << 16, 1 << 17]
SIZES_MIXED = [3, 6, 12, 48, 96, 3 << 10, 3 << 12, 98304]


@pytest.mark.parametrize("n", SIZES_POW2 + SIZES_MIXED)
def test_forward_matches_oracle(gpu, n):
    x = rand_fr_wire(n, 1000 + n)
    got = gpu.ntt(x)
    want = oc.ntt(x, threads=8)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", SIZES_POW2 + SIZES_MIXED)
def test_inverse_matches_oracle_and_roundtrip(gpu, n):
    x = rand_fr_wire(n, 2000 + n)
    inv = gpu.ntt(x, inverse=True)
    assert np.array_equal(inv, oc.ntt(x, inverse=True, threads=8))
    assert np.array_equal(gpu.ntt(gpu.ntt(x), inverse=True), x)


def test_reference_test_fft_cases(gpu):
    """The literal cases of the reference's test_fft: [1], [1,1], [1,0], [0,1] on sizes 1/2 and
    [0,1,1] on the size-3 domain, each checked as fft[i] == poly.eval(group_gen^i)."""
    cases = [([1], 1), ([1, 1], 2), ([1, 0], 2), ([0, 1], 2), ([0, 1, 1], 3)]
    for coefs, n in cases:
        got = oc.fr_to_ints(gpu.ntt(oc.fr_from_ints(coefs)))
        w = opy.root_of_unity(n)
        assert got == [opy.poly_eval(coefs, pow(w, i, opy.R)) for i in range(n)]
        assert opy.from_mont(opy.limbs_to_int(gpu.domain_group_gen(n)), opy.R) == w


def test_short_input_zero_padded_by_caller(gpu):
    """n+3 coefficients into a 6n domain (the t_poly shape, helpers.rs:256-266): the caller shim
    zero-pads; result equals evaluating the short polynomial on the whole domain."""
    n = 16
    coefs = rand_fr(n + 3, 7)
    dom = 6 * n
    got = oc.fr_to_ints(gpu.ntt(oc.fr_from_ints(coefs + [0] * (dom - len(coefs)))))
    w = opy.root_of_unity(dom)
    assert got == [opy.poly_eval(coefs, pow(w, i, opy.R)) for i in range(dom)]


@pytest.mark.parametrize("n", [8, 48, 4096, 3 << 12])
def test_coset_fft_and_ifft(gpu, n):
    """coset_fft_with_domain / coset_ifft_with_domain (field_polynomial.rs:589-607)."""
    k = 7  # any non-trivial shift; the prover uses a fixed quadratic non-residue
    kw = oc.fr_from_ints([k])[0]
    kinv = oc.fr_from_ints([pow(k, -1, opy.R)])[0]
    x = rand_fr_wire(n, 3000 + n)
    fwd = gpu.ntt(x, coset_shift=kw)
    assert np.array_equal(fwd, oc.ntt(oc.mul_var(x, kw), threads=8))
    back = gpu.ntt(fwd, inverse=True, coset_shift=kinv)
    assert np.array_equal(back, x)


def test_unsupported_domain_is_fft_error(gpu):
    from uzkge_amd import UzkgeError
    with pytest.raises(UzkgeError) as e:
        gpu.ntt(rand_fr_wire(5, 1))
    assert e.value.kind == "FFTError"


def test_full_size_2p22(gpu):
    """BASELINE.json config: 2^22 coefficients, forward + inverse round trip bit-exact, forward
    output equal to the CPU oracle on the whole vector and to Horner p(w^i) at sampled points."""
    n = 1 << 22
    x = rand_fr_wire(n, 22)
    f = gpu.ntt(x)
    assert np.array_equal(f, oc.ntt(x, threads=16))
    assert np.array_equal(gpu.ntt(f, inverse=True), x)
    w = oc.root_of_unity(n)
    wi = oc.fr_from_ints([1])[0]
    for i in range(4):
        assert np.array_equal(f[i], oc.poly_eval(x, wi)), i
        wi = oc.fr_mul(wi, w)
