"""CPU: the oracle's quotient polynomial and grand product held to the reference's VERIFIER formulas (no GPU, no commitments).

For a circuit its witness satisfies (tests/plonk_verifier_oracle.py make_satisfiable: gate, copy constraints, boolean gate,
anemoi rounds and the shuffle gadget all live), the linearisation identity the verifier relies on is plain field arithmetic:

    sum_k scalar_k p_k(zeta)  -  Z_H(zeta) t(zeta)   ==   r_eval_zeta(evaluations)

with the scalars of `r_poly_or_comm` (uzkge/src/plonk/helpers.rs:681-1002) and the right-hand side of `r_eval_zeta` (:1182-1321).
t(X) comes from the oracle's term-by-term restatement of `t_poly`'s loop (oracle/bn254_oracle.c, helpers.rs:284-656) through
the coset iFFT, z(X) from the oracle's `z_poly` (:160-220): two readings of different reference functions that must agree at a
random point.  This pins the ORACLE (the checker of the GPU quotient kernel) on the reference's verifier; the GPU chain itself is
held to the same equations, plus commitments and pairings, in tests/test_gpu_plonk_verifier.py."""
import os
import sys

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
import plonk_verifier_oracle as pv

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
R = opy.R


def _pad(wire, length):
    out = np.zeros((length, 4), dtype=np.uint64)
    out[: wire.shape[0]] = wire
    return out


def _identity_sides(inp, shuffle):
    import prover_chain as pch
    n, m = inp.n, inp.m
    ch = pv._challenge_ints(inp)
    k = pv._ints(inp.k)
    omega = pv._ints(inp.group_gen)[0]
    group = oc.fr_from_ints([pow(omega, i, R) for i in range(n)])
    # the proof's polynomials in coefficient form (no hiding: any multiple of Z_H added to them leaves the identity intact)
    evals9 = [inp.w_evals[i] for i in range(5)] + [inp.wsel_evals[i] for i in range(3)] + [inp.pi_evals]
    polys = [oc.ntt(np.ascontiguousarray(e), inverse=True) for e in evals9]
    z_evals = oc.z_poly(inp.w_evals, inp.perm, group, inp.k, inp.beta, inp.gamma)
    polys.append(oc.ntt(z_evals, inverse=True))
    cos = np.stack([oc.ntt(oc.mul_var(_pad(p, m), inp.k[1])) for p in polys])
    tables = np.stack([oc.ntt(oc.mul_var(_pad(inp.table_polys[i], m), inp.k[1])) for i in range(inp.table_polys.shape[0])])
    vecs = np.concatenate([cos, tables])
    if not shuffle:
        for slot in list(range(5, 8)) + list(range(31, 56)):
            vecs[slot] = 0
    tq = oc.t_quotient(n, 6, vecs, inp.alpha, inp.beta, inp.gamma, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a, inp.z_h_inv)
    t = oc.mul_var(oc.ntt(tq, inverse=True), inp.k1_inv)
    t_tail_nonzero = bool(np.any(t[5 * n + 11:]))
    ev_at = lambda wire, point: pv._ints(oc.poly_eval(np.ascontiguousarray(wire), point))[0]
    zeta_w, zeta_omega_w = inp.zeta, inp.zeta_omega
    tp = inp.table_polys
    ev = {"w": [ev_at(polys[i], zeta_w) for i in range(5)], "s": [ev_at(tp[pch.T_S + i], zeta_w) for i in range(4)],
          "prk3": ev_at(tp[pch.T_QPRK + 2], zeta_w), "prk4": ev_at(tp[pch.T_QPRK + 3], zeta_w), "z_omega": ev_at(polys[9], zeta_omega_w),
          "w_omega": [ev_at(polys[i], zeta_omega_w) for i in range(3)]}
    if shuffle:
        ev["q_ecc"] = ev_at(tp[pch.T_QECC], zeta_w)
        ev["wsel"] = [ev_at(polys[5 + i], zeta_w) for i in range(3)]
    scalars = pv.r_scalars(ch, k, n, ev, shuffle)
    plan = pch.r_plan(shuffle)
    lhs = 0
    for (kind, idx), s in zip(plan, scalars):
        if kind == "k":
            continue                                   # the five chunk terms add up to -Z_H(zeta) t(zeta): the blinds telescope
        p = tp[idx] if kind == "t" else polys[idx]
        lhs = (lhs + s * ev_at(p, zeta_w)) % R
    zh, _ = pv.first_lagrange_poly(ch["zeta"], n)
    lhs = (lhs - zh * ev_at(t, zeta_w)) % R
    pi = pv._ints(inp.pi_evals[:8])
    pi_eval = pv.eval_pi_poly({i: v for i, v in enumerate(pi)}, ch["zeta"], zh, omega, n)
    assert pi_eval == ev_at(polys[8], zeta_w)          # eval_pi_poly (helpers.rs:1135-1165) == the interpolated PI(X) at zeta
    return lhs, pv.r_eval_zeta(ch, n, ev, pi_eval, shuffle), t_tail_nonzero, sum(1 for s in scalars if s)


@pytest.mark.parametrize("shuffle", [True, False])
def test_oracle_quotient_satisfies_the_verifier_identity(shuffle):
    import prover_chain as pch
    inp = pv.make_satisfiable(pch.ChainInputs(4096, 33), seed=6)
    lhs, rhs, tail, live = _identity_sides(inp, shuffle)
    assert live == (43 if shuffle else 19)             # every polynomial of r(X) takes part with a non-zero scalar
    assert not tail
    assert lhs == rhs


def test_a_wrong_witness_value_breaks_the_identity():
    import prover_chain as pch
    inp = pv.make_satisfiable(pch.ChainInputs(4096, 33), seed=6)
    inp.w_evals[3, 100] = oc.fr_from_ints([(pv._ints(inp.w_evals[3, 100:101])[0] + 1) % R])[0]
    lhs, rhs, tail, _ = _identity_sides(inp, True)
    assert tail and lhs != rhs
