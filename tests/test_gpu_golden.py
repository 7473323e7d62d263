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


def test_prover_chain_against_frozen_outputs(gpu):
    """The prover-round chain at n = 2^12 against tests/golden/vectors_v2.npz (made by make_vectors_v2.py from the oracle
    chain): commitments over the reference's SRS files, evaluations, blinds, and digests of the large intermediates."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from prover_chain import ChainInputs, ProverChain
    V2 = np.load(os.path.join(GOLDEN, "vectors_v2.npz"))
    n = int(V2["n"][0])
    c = ProverChain(inputs=ChainInputs(n, int(V2["seed"][0])))
    try:
        o = c.run()
        m = c.m
        for key in ("cm_w_wsel", "cm_z", "cm_t", "cm_q"):
            assert np.array_equal(oc.points_from_affine([affine_of(j) for j in o[key]]), V2[key]), key
        for key in ("evals_zeta", "z_eval_zeta_omega", "open_evals_zeta", "open_evals_zeta_omega"):
            assert np.array_equal(o[key], V2[key]), key
        assert np.array_equal(np.concatenate([x.reshape(-1, 4) for x in o["t_blinds"]]), V2["t_blinds"])
        assert np.array_equal(np.concatenate([x.reshape(-1, 4) for x in o["q_blinds"]]), V2["q_blinds"])
        host = lambda t: t.cpu().numpy().view(np.uint64)
        big = {"coefs": host(c.d_coefs).reshape(10, m, 4)[:, : n + 3], "coset_evals": host(c.d_coset).reshape(10, m, 4),
               "t_quotient": host(c.d_tq), "t": host(c.d_t), "z_evals": host(c.d_z), "r": host(c.d_r)[: n + 3]}
        for key, arr in big.items():
            dig = np.frombuffer(hashlib.sha256(np.ascontiguousarray(arr, dtype=np.uint64).tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(dig, V2["sha256_" + key]), key
    finally:
        c.release()
