"""Openings produced by the HIP path, verified with the reference's own verifier equation over the reference's own G2
parameters (oracle/bn254_pairing.py; `KZGCommitmentScheme::verify`, uzkge/src/poly_commit/kzg_poly_commitment.rs:344-371):

        e(C - [v]G, H) == e(pi, [tau]H - [z]H),   H, [tau]H = the G2 tail of parameters/srs-padding.bin.

An opening proof is unique given (C, z, v), so a proof that satisfies the equation is THE proof arkworks would have
produced: this pins `batch_prove`'s polynomial half (pcs.rs:107-168: evaluations, linear combination in powers of alpha,
division by X - z, fold mod X^N - 1, FFT(N), Lagrange commit, blind factors) on reference-held data -- SURVEY.md section 8
rows a7 and f4, which no stored fixture of the reference covers -- and, for the device-resident prover chain, everything
between the witness and the opening at zeta * omega (iFFT, hiding, tail-scalar commit, evaluations, opening quotient)."""
import os
import sys

import numpy as np
import pytest

import bn254_py as opy
import bn254_pairing as pr
import oracle_c as oc
from util import GOLDEN, affine_of, rand_fr_wire

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _blob(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


@pytest.fixture(scope="module")
def g2():
    return pr.parse_srs_g2(_blob("srs-padding.bin"))


def _combine(cms, vals, alpha):
    """sum alpha^k C_k, sum alpha^k v_k (the verifier's side of batch_prove's linear combination, pcs.rs:119-131)."""
    c, v, mult = None, 0, 1
    for cm, val in zip(cms, vals):
        c = opy.g1_add(c, opy.g1_mul(cm, mult))
        v = (v + mult * val) % opy.R
        mult = mult * alpha % opy.R
    return c, v


@pytest.mark.parametrize("n", [4096, 8192, 16384])     # 8192: zmatchmaking's circuit size (matchmaking/src/build_cs.rs:68-99)
def test_batch_prove_opening_verifies_under_the_reference_g2(gpu, g2, n):
    from uzkge_amd.poly_commit import FpPolynomial, KZGCommitmentSchemeBN254, batch_prove, fr_from_int, fr_to_int, load_srs_params
    pcs = load_srs_params(_blob("srs-padding.bin"), n)
    lag = KZGCommitmentSchemeBN254.from_unchecked_bytes(_blob(f"lagrange-srs-{n}.bin"))
    try:
        # polynomials of the prover's shape (n + 3 coefficients) that the MONOMIAL parameters can commit to: non-zero where
        # srs-padding.bin holds real powers (0..2050 and the padding powers N..N+2).  Their commitments come from the CPU oracle.
        polys = []
        for k in range(3):
            c = np.zeros((n + 3, 4), dtype=np.uint64)
            c[:2051] = rand_fr_wire(2051, 100 + k)
            c[n:] = rand_fr_wire(3, 200 + k)
            polys.append(c)
        g1 = pcs.public_parameter_group_1
        cms = [affine_of(oc.msm_pippenger(g1, c, 0, 8)) for c in polys]
        z, alpha = 0x1F2E3D4C5B6A79881122334455667788, 0x0123456789ABCDEFFEDCBA9876543210
        cm_q, evals = batch_prove(pcs, lag, [FpPolynomial.from_coefs(c) for c in polys], fr_from_int(z), fr_from_int(alpha))
        vals = [fr_to_int(e) for e in evals]
        assert vals == [opy.poly_eval(oc.fr_to_ints(c), z) for c in polys]
        c_comb, v_comb = _combine(cms, vals, alpha)
        g1_0 = opy.wire_to_affine(g1[0].tobytes())
        proof = affine_of(cm_q)
        assert pr.kzg_verify(g1_0, g2[0], g2[1], c_comb, z, v_comb, proof, opy.g1_mul, opy.g1_add)
        assert not pr.kzg_verify(g1_0, g2[0], g2[1], c_comb, z, (v_comb + 1) % opy.R, proof, opy.g1_mul, opy.g1_add)
    finally:
        pcs.release()
        lag.release()


def test_prover_chain_opening_at_zeta_omega_verifies(gpu, g2):
    """The device-resident chain (tools/prover_chain.py, prover.rs:151-372): its commitments of z, w0, w1, w2, its evaluations
    at zeta * omega and its second opening proof satisfy the verifier's equation -- commitments, evaluations and proof all
    come from the device; only the equation and the G2 elements are the reference's."""
    from prover_chain import ChainInputs, ProverChain, eval_plan
    from uzkge_amd.poly_commit import fr_to_int
    inp = ChainInputs(1 << 14, 7)
    c = ProverChain(inputs=inp)
    try:
        o = c.run()
        plan = eval_plan(True)
        at = {(kind, idx, pt): i for i, (kind, idx, pt) in enumerate(plan)}
        order = [("c", 9), ("c", 0), ("c", 1), ("c", 2)]                       # open_plan: z, w0, w1, w2 (prover.rs:361-362)
        vals = [fr_to_int(o["evals"][at[(k, i, 1)]]) for k, i in order]
        cms = [affine_of(o["cm_z"][0])] + [affine_of(o["cm_w_wsel"][i]) for i in range(3)]
        alpha, point = fr_to_int(inp.alpha_open2), fr_to_int(inp.zeta_omega)
        c_comb, v_comb = _combine(cms, vals, alpha)
        g1_0 = opy.wire_to_affine(inp.mono_wire[0].tobytes())
        proof = affine_of(o["cm_q"][1])
        assert pr.kzg_verify(g1_0, g2[0], g2[1], c_comb, point, v_comb, proof, opy.g1_mul, opy.g1_add)
        assert not pr.kzg_verify(g1_0, g2[0], g2[1], c_comb, point, v_comb, affine_of(o["cm_q"][0]), opy.g1_mul, opy.g1_add)
    finally:
        c.release()


def test_prover_chain_opening_at_zeta_verifies(gpu, g2):
    """The chain's first opening (prover.rs:329-347): sixteen polynomials at zeta -- five wires, four permutation
    polynomials, q_prk3, q_prk4, q_ecc, three wire selectors and r(X) = sum of 43 scalars times polynomials (helpers.rs:681-999).
    The verifier's side is assembled from commitments only: the chain's own (wires, selectors, z, the five t chunks), the
    circuit polynomials' (one batched Lagrange commit of their evaluations, as the indexer does), and
    C_r = sum scalar_k C_k; the evaluation of r is read off the device's r.  The equation holds only if the linear
    combination, all sixteen evaluations, the division by X - zeta, the fold and the blind factors are all right."""
    import prover_chain as pch
    from uzkge_amd import backend as b
    from uzkge_amd.poly_commit import fr_to_int
    inp = pch.ChainInputs(1 << 14, 9)
    c = pch.ProverChain(inputs=inp, precompute=False)
    try:
        o = c.run()
        snap = c.snapshot()
        n = inp.n
        # commitments of the 46 circuit polynomials: evaluations over the domain, MSM over the Lagrange parameters
        table_evals = b.ntt_batch(inp.table_polys)
        table_cms = [affine_of(j) for j in b.msm_batch(c.srs, table_evals)]
        own = [affine_of(j) for j in o["cm_w_wsel"]] + [None, affine_of(o["cm_z"][0])]        # slots 0..7, (8 = pi: never opened), 9
        chunks = [affine_of(j) for j in o["cm_t"]]

        def cm_of(kind, idx):
            return table_cms[idx] if kind == "t" else own[idx] if kind == "c" else chunks[idx]
        rp = pch.r_plan(True)
        assert len(rp) == 43
        cm_r = None
        for (kind, idx), s in zip(rp, inp.r_scalars):
            cm_r = opy.g1_add(cm_r, opy.g1_mul(cm_of(kind, idx), fr_to_int(s)))
        zeta, alpha = fr_to_int(inp.zeta), fr_to_int(inp.alpha_open)
        at = {(kind, idx, pt): i for i, (kind, idx, pt) in enumerate(pch.eval_plan(True))}
        at_zeta, _ = pch.open_plan(True)
        cms, vals = [], []
        for kind, idx in at_zeta:
            if kind == "r":
                cms.append(cm_r)
                vals.append(opy.poly_eval(oc.fr_to_ints(snap["r"]), zeta))
            else:
                cms.append(cm_of(kind, idx))
                vals.append(fr_to_int(o["evals"][at[(kind, idx, 0)]]))
        c_comb, v_comb = _combine(cms, vals, alpha)
        g1_0 = opy.wire_to_affine(inp.mono_wire[0].tobytes())
        assert pr.kzg_verify(g1_0, g2[0], g2[1], c_comb, zeta, v_comb, affine_of(o["cm_q"][0]), opy.g1_mul, opy.g1_add)
        assert not pr.kzg_verify(g1_0, g2[0], g2[1], c_comb, zeta, v_comb, affine_of(o["cm_q"][1]), opy.g1_mul, opy.g1_add)
    finally:
        c.release()
