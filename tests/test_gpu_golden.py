"""GPU path against the COMMITTED golden vectors (tests/golden/vectors_v1.npz): no oracle code runs
for the expected values here -- inputs and outputs are data."""
import os

import numpy as np
import pytest

import oracle_c as oc            # only for wire <-> affine conversions of the results
from util import GOLDEN, load_srs, affine_of

pytestmark = pytest.mark.gpu
V = np.load(os.path.join(GOLDEN, "vectors_v1.npz"))


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_field_ops(gpu, field):
    a, b = V[f"{field}_a"], V[f"{field}_b"]
    assert np.array_equal(gpu.field_op(field, 0, a, b), V[f"{field}_mul"])      # assembly product
    assert np.array_equal(gpu.field_op(field, 3, a, b), V[f"{field}_mul"])      # portable product
    assert np.array_equal(gpu.field_op(field, 10, a, b), V[f"{field}_mul"])     # 29-bit-limb product
    assert np.array_equal(gpu.field_op(field, 1, a, b), V[f"{field}_add"])
    assert np.array_equal(gpu.field_op(field, 2, a, b), V[f"{field}_sub"])


def test_g1_additions(gpu):
    want = V["g1_sum"]
    for op in (0, 1):                      # mixed addition and full XYZZ addition
        got = gpu.g1_op(op, V["g1_a"], V["g1_b"])
        for i in range(want.shape[0]):
            assert np.array_equal(oc.points_from_affine([affine_of(got[i])])[0], want[i]), (op, i)


@pytest.mark.parametrize("n", [1, 2, 33, 1024, 4096])
def test_msm(gpu, n):
    srs_wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(srs_wire[:n])
    try:
        for kind in ("uniform", "zero", "one", "rminus1", "boolean_heavy"):
            got = gpu.msm(srs, V[f"msm_{n}_{kind}_scalars"])
            assert np.array_equal(oc.points_from_affine([affine_of(got)]), V[f"msm_{n}_{kind}_affine"]), kind
    finally:
        srs.release()


@pytest.mark.parametrize("n", [1, 2, 4, 16, 48, 1024, 1 << 14])
def test_ntt(gpu, n):
    x = V[f"ntt_{n}_in"]
    assert np.array_equal(gpu.ntt(x), V[f"ntt_{n}_fwd"])
    assert np.array_equal(gpu.ntt(x, inverse=True), V[f"ntt_{n}_inv"])
    assert np.array_equal(gpu.ntt(x, coset_shift=oc.fr_from_ints([7])[0]), V[f"ntt_{n}_coset7"])


def check_against_frozen(V3, small, big):
    """small: name -> array (commitments as Jacobian [k, 12]), big: name -> array; V3: the loaded fixture."""
    import hashlib
    for key in ("cm_w_wsel", "cm_z", "cm_t", "cm_q"):
        assert np.array_equal(oc.points_from_affine([affine_of(j) for j in small[key]]), V3[key]), key
    for key in ("evals", "t_blinds", "q_blinds"):
        assert np.array_equal(np.asarray(small[key]).reshape(V3[key].shape), V3[key]), key
    for key in [f[len("sha256_"):] for f in V3.files if f.startswith("sha256_")]:
        dig = np.frombuffer(hashlib.sha256(np.ascontiguousarray(big[key], dtype=np.uint64).tobytes()).digest(), dtype=np.uint8)
        assert np.array_equal(dig, V3["sha256_" + key]), key


def test_prover_chain_against_frozen_outputs(gpu):
    """The prover-round chain at n = 2^12 against tests/golden/vectors_v3.npz (made by make_vectors_v3.py from the oracle
    chain): commitments over the reference's SRS files, evaluations, blinds, and digests of the large intermediates."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from prover_chain import ChainInputs, ProverChain
    V3 = np.load(os.path.join(GOLDEN, "vectors_v3.npz"))
    n = int(V3["n"][0])
    c = ProverChain(inputs=ChainInputs(n, int(V3["seed"][0])), keep_blinds=True)
    try:
        o = c.run()
        snap = c.snapshot()
        assert not snap["coefs_beyond"].any()
        check_against_frozen(V3, o, snap)
    finally:
        c.release()
