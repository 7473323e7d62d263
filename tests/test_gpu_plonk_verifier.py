"""The device-resident prover chain held to the reference's VERIFIER equations, for a circuit its witness really satisfies.

tests/test_gpu_kzg_pairing.py pins the opening flow: there r(X) uses seeded stand-in scalars and r's evaluation is read off the
device.  Here the circuit is satisfiable (tests/plonk_verifier_oracle.py make_satisfiable), r(X)'s 43 scalars are the reference's
formulas of the proof's evaluations (helpers.rs:681-1002), and the value the opening at zeta must hit is the one the VERIFIER
derives from those evaluations alone (`r_eval_zeta`, helpers.rs:1182-1321; PI(zeta): `eval_pi_poly`, :1135-1165).  The pairing
equation (kzg_poly_commitment.rs:344-371, the reference's G2 parameters) then holds only if
     t(zeta) Z_H(zeta) = [gate + permutation + L1 + boolean + anemoi + shuffle terms](zeta)
i.e. only if the quotient kernel (`t_poly`, helpers.rs:223-678), the grand product (`z_poly`, :160-220), `split_t_and_commit`
(:1323-1408), the evaluations and the linear combination are all right -- SURVEY.md section 8 rows f2, f4, a8, which no stored
fixture of the reference covers.  A witness with ONE wrong value must fail the same check."""
import os
import sys

import numpy as np
import pytest

import bn254_py as opy
import bn254_pairing as pr
import oracle_c as oc
import plonk_verifier_oracle as pv
from util import GOLDEN, affine_of

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _run_and_verify(inp, shuffle=True):
    import prover_chain as pch
    from uzkge_amd import backend as b
    g2 = pr.parse_srs_g2(open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read())
    n = inp.n
    ch = pv._challenge_ints(inp)
    k = pv._ints(inp.k)
    plan = pch.eval_plan(shuffle)
    at = {(kind, idx, pt): i for i, (kind, idx, pt) in enumerate(plan)}

    def evals_of(rows):
        v = pv._ints(rows)
        ev = {"w": [v[at[("c", i, 0)]] for i in range(5)], "s": [v[at[("t", pch.T_S + i, 0)]] for i in range(4)],
              "prk3": v[at[("t", pch.T_QPRK + 2, 0)]], "prk4": v[at[("t", pch.T_QPRK + 3, 0)]], "z_omega": v[at[("c", 9, 1)]],
              "w_omega": [v[at[("c", i, 1)]] for i in range(3)]}
        if shuffle:
            ev["q_ecc"] = v[at[("t", pch.T_QECC, 0)]]
            ev["wsel"] = [v[at[("c", 5 + i, 0)]] for i in range(3)]
        return ev
    c = pch.ProverChain(inputs=inp, precompute=False, shuffle=shuffle)
    c.r_scalar_hook = lambda rows: oc.fr_from_ints(pv.r_scalars(ch, k, n, evals_of(rows), shuffle))
    try:
        o = c.run()
        ev = evals_of(o["evals"])
        scalars = pv.r_scalars(ch, k, n, ev, shuffle)
        # the verifier's commitments: the circuit's (one batched Lagrange commit of its polynomials' evaluations) and the proof's
        table_cms = [affine_of(j) for j in b.msm_batch(c.srs, b.ntt_batch(inp.table_polys))]
        own = [affine_of(j) for j in o["cm_w_wsel"]] + [None, affine_of(o["cm_z"][0])]
        chunks = [affine_of(j) for j in o["cm_t"]]
        cm_of = lambda kind, idx: table_cms[idx] if kind == "t" else own[idx] if kind == "c" else chunks[idx]
        rp = pch.r_plan(shuffle)
        assert len(rp) == len(scalars)
        cm_r = None                                                             # r_commitment (helpers.rs:1082-1133)
        for (kind, idx), s in zip(rp, scalars):
            cm_r = opy.g1_add(cm_r, opy.g1_mul(cm_of(kind, idx), s))
        zh, _ = pv.first_lagrange_poly(ch["zeta"], n)
        pi = pv._ints(inp.pi_evals[:8])
        pi_eval = pv.eval_pi_poly({i: v for i, v in enumerate(pi)}, ch["zeta"], zh, pv._ints(inp.group_gen)[0], n)
        v_r = pv.r_eval_zeta(ch, n, ev, pi_eval, shuffle)                       # the VERIFIER's value, from the evaluations alone
        at_zeta, _ = pch.open_plan(shuffle)
        vals_by = dict(zip([p for p in at_zeta if p[0] != "r"], ev["w"] + ev["s"] + [ev["prk3"], ev["prk4"]] +
                           ([ev["q_ecc"]] + ev["wsel"] if shuffle else [])))
        cms, vals = [], []
        for kind, idx in at_zeta:
            cms.append(cm_r if kind == "r" else cm_of(kind, idx))
            vals.append(v_r if kind == "r" else vals_by[(kind, idx)])
        alpha_open = pv._ints(inp.alpha_open)[0]
        comb_c, comb_v, mult = None, 0, 1
        for cm, val in zip(cms, vals):                                          # pcs.batch (pcs.rs:170-200): powers of the batching challenge
            comb_c = opy.g1_add(comb_c, opy.g1_mul(cm, mult))
            comb_v = (comb_v + mult * val) % opy.R
            mult = mult * alpha_open % opy.R
        g1_0 = opy.wire_to_affine(inp.mono_wire[0].tobytes())
        ok = pr.kzg_verify(g1_0, g2[0], g2[1], comb_c, ch["zeta"], comb_v, affine_of(o["cm_q"][0]), opy.g1_mul, opy.g1_add)
        t_tail = c.snapshot()["t"][5 * n + 11:]
        return ok, bool(np.any(t_tail))
    finally:
        c.release()


@pytest.mark.parametrize("n,shuffle", [(1 << 14, True), (1 << 14, False), (1 << 13, False)])     # 2^13 without the shuffle terms: zmatchmaking's shape
def test_chain_satisfies_the_verifier_equations(gpu, n, shuffle):
    import prover_chain as pch
    inp = pv.make_satisfiable(pch.ChainInputs(n, 21), seed=4)
    ok, tail = _run_and_verify(inp, shuffle)
    assert not tail          # the numerator is divisible by Z_H: t has 5n + 11 coefficients and nothing beyond
    assert ok


@pytest.mark.parametrize("what", ["wire", "wire_selector"])
def test_one_wrong_witness_value_fails_the_verifier_equations(gpu, what):
    import prover_chain as pch
    inp = pv.make_satisfiable(pch.ChainInputs(1 << 14, 21), seed=4)
    if what == "wire":               # breaks row 777's gate, an anemoi relation and a copy constraint
        inp.w_evals[2, 777] = oc.fr_from_ints([(pv._ints(inp.w_evals[2, 777:778])[0] + 1) % opy.R])[0]
    else:                            # row 21 carries the shuffle gadget: the other table column no longer matches the wires
        bit = pv._ints(inp.wsel_evals[0, 21:22])[0]
        assert bit in (0, 1) and pv._ints(inp.wsel_evals[2, 21:22])[0] in (1, opy.R - 1)
        inp.wsel_evals[0, 21] = oc.fr_from_ints([1 - bit])[0]
    # the prover itself notices: t no longer has the 5n + 11 coefficients of a satisfied circuit, and round 3 refuses to go on
    # (the reference aborts at this point: apply_blind_factors indexes past its n + 3 SRS powers, kzg_poly_commitment.rs:299-313)
    from uzkge_amd import UzkgeError
    from uzkge_amd import _native as N
    with pytest.raises(UzkgeError) as e:
        _run_and_verify(inp, True)
    assert e.value.code == N.UZK_ERR_COMMITMENT and "does not satisfy" in str(e.value)
    # forced past that check (t read as its first 5n + 11 coefficients), the proof comes out and the verifier's equations reject it
    inp.satisfiable = False
    ok, tail = _run_and_verify(inp, True)
    assert tail              # the division by Z_H leaves a remainder: the 6n-point interpolation fills the top coefficients
    assert not ok


class _FiatShamir:
    """The prover's side of the reference's transcript (prover.rs:151-372 appends what verifier.rs:166-222 re-derives), driving the
    chain's challenges; tests/plonk_golden_verifier.py holds the transcript itself, pinned on the reference's golden proof."""

    def __init__(self, vk, pi, n_cards, plan, shuffle=True):
        import plonk_golden_verifier as gv
        from uzkge_amd.poly_commit import fr_from_int
        self.wire, self.plan, self.vk, self.ch = fr_from_int, plan, vk, {}
        t = gv.Transcript(b"Plonk shuffle Proof")
        t.append_u64(n_cards)
        t.append_message(b"PLONK")
        t.append_u64(vk["cs_size"])
        t.append_message(opy.R.to_bytes(32, "big"))
        for c in vk["cm_q"] + vk["cm_s"]:
            t.append_commitment(c)
        t.append_challenge(vk["root"])
        for k in vk["k"]:
            t.append_challenge(k)
        for v in pi:
            t.append_challenge(v)
        self.t = t

    def beta_gamma(self, cms):
        for j in cms:
            self.t.append_commitment(affine_of(j))
        self.ch["beta"] = self.t.challenge()
        self.t.append_single_byte(0x01)
        self.ch["gamma"] = self.t.challenge()
        return self.wire(self.ch["beta"]), self.wire(self.ch["gamma"])

    def alpha(self, cm_z):
        self.t.append_commitment(affine_of(cm_z[0]))
        self.ch["alpha"] = self.t.challenge()
        return self.wire(self.ch["alpha"])

    def zeta(self, cm_t):
        for j in cm_t:
            self.t.append_commitment(affine_of(j))
        self.ch["zeta"] = self.t.challenge()
        return self.wire(self.ch["zeta"])

    def after_evaluations(self, rows, zeta_w, zeta_omega_w):
        import prover_chain as pch
        v = pv._ints(rows)
        at = {(kind, idx, pt): i for i, (kind, idx, pt) in enumerate(self.plan)}
        order = [("c", i, 0) for i in range(5)] + [("t", pch.T_S + i, 0) for i in range(4)] + [("c", 5 + i, 0) for i in range(3)] + \
            [("t", pch.T_QPRK + 2, 0), ("t", pch.T_QPRK + 3, 0), ("c", 9, 1), ("t", pch.T_QECC, 0)] + [("c", i, 1) for i in range(3)]
        for key in order:                                       # prover.rs:275-298
            self.t.append_challenge(v[at[key]])
        self.ch["u"] = self.t.challenge()
        out = []
        for point in (zeta_w, zeta_omega_w):                    # batch_prove -> init_pcs_batch_eval_transcript + alpha (pcs.rs:107-118)
            self.t.append_message(b"New PCS-Batch-Eval Protocol")
            self.t.append_message(opy.R.to_bytes(32, "big"))
            self.t.append_u64(self.vk["cs_size"] + 2)
            self.t.append_challenge(pv._ints(point)[0])
            out.append(self.wire(self.t.challenge()))
        return out


def test_a_whole_proof_from_the_chain_is_accepted_by_the_golden_proof_verifier(gpu):
    """Prove -> verify, non-interactively: the chain's challenges come from the reference's transcript, its commitments,
    evaluations and opening proofs are packed as a PlonkProof, the circuit's commitments as a verifier key, and
    tests/plonk_golden_verifier.py -- the verifier restatement that accepts the REFERENCE's golden proof -- must accept it
    (and reject it with one evaluation changed).  The circuit has anemoi rounds and the quintic selector live, which the golden
    circuit has not."""
    import plonk_golden_verifier as gv
    import prover_chain as pch
    from uzkge_amd import backend as b
    n = 1 << 14
    inp = pv.make_satisfiable(pch.ChainInputs(n, 21), seed=4)
    c = pch.ProverChain(inputs=inp, precompute=False)
    try:
        table_cms = [affine_of(j) for j in b.msm_batch(c.srs, b.ntt_batch(inp.table_polys))]
        omega = pv._ints(inp.group_gen)[0]
        ninv = pow(n, -1, opy.R)
        g = pv._ints(inp.anemoi_g)[0]
        vk = {"cm_q": table_cms[pch.T_Q:pch.T_Q + 9], "cm_s": table_cms[pch.T_S:pch.T_S + 5], "cm_qb": table_cms[pch.T_QB],
              "cm_prk": table_cms[pch.T_QPRK:pch.T_QPRK + 4], "cm_q_ecc": table_cms[pch.T_QECC],
              "cm_shuffle_generator": table_cms[pch.T_QG:pch.T_QG + 12], "cm_shuffle_public_key": table_cms[pch.T_QPK:pch.T_QPK + 12],
              "anemoi_g": g, "anemoi_g_inv": pow(g, -1, opy.R), "k": pv._ints(inp.k), "edwards_a": pv._ints(inp.edwards_a)[0],
              "root": omega, "cs_size": n, "pi_root_powers": [pow(omega, j, opy.R) for j in range(8)],
              "pi_lagrange": [pow(omega, j, opy.R) * ninv % opy.R for j in range(8)]}
        pi = pv._ints(inp.pi_evals[:8])
        plan = pch.eval_plan(True)
        fs = _FiatShamir(vk, pi, 52, plan)
        c.fs = fs
        k = vk["k"]
        at = {(kind, idx, pt): i for i, (kind, idx, pt) in enumerate(plan)}

        def evals_of(rows):
            v = pv._ints(rows)
            return {"w": [v[at[("c", i, 0)]] for i in range(5)], "s": [v[at[("t", pch.T_S + i, 0)]] for i in range(4)],
                    "prk3": v[at[("t", pch.T_QPRK + 2, 0)]], "prk4": v[at[("t", pch.T_QPRK + 3, 0)]], "z_omega": v[at[("c", 9, 1)]],
                    "w_omega": [v[at[("c", i, 1)]] for i in range(3)], "q_ecc": v[at[("t", pch.T_QECC, 0)]],
                    "wsel": [v[at[("c", 5 + i, 0)]] for i in range(3)]}
        chd = lambda: {"alpha": fs.ch["alpha"], "beta": fs.ch["beta"], "gamma": fs.ch["gamma"], "zeta": fs.ch["zeta"], "anemoi_g": g,
                       "edwards_a": vk["edwards_a"]}
        c.r_scalar_hook = lambda rows: oc.fr_from_ints(pv.r_scalars(chd(), k, n, evals_of(rows), True))
        o = c.run()
        ev = evals_of(o["evals"])
        proof = {"cm_w": [affine_of(j) for j in o["cm_w_wsel"][:5]], "cm_wsel": [affine_of(j) for j in o["cm_w_wsel"][5:8]],
                 "cm_t": [affine_of(j) for j in o["cm_t"]], "cm_z": affine_of(o["cm_z"][0]), "prk3": ev["prk3"], "prk4": ev["prk4"],
                 "w": ev["w"], "w_omega": ev["w_omega"], "z_omega": ev["z_omega"], "s": ev["s"], "q_ecc": ev["q_ecc"], "wsel": ev["wsel"],
                 "open_zeta": affine_of(o["cm_q"][0]), "open_zeta_omega": affine_of(o["cm_q"][1])}
        assert ev["prk3"] != 0 and vk["cm_q"][7] is not None            # anemoi rounds and the quintic selector are live here
        assert gv.verify(vk, proof, pi, n_cards=52)
        raw = gv.proof_to_bytes(proof)                                  # the reference's proof format (indexer.rs:539-590): 1632 bytes
        assert len(raw) == 1632 and gv.verify(vk, gv.proof_from_bytes(raw), pi, n_cards=52)
        bad = dict(proof); bad["s"] = list(proof["s"]); bad["s"][1] = (bad["s"][1] + 1) % opy.R
        assert not gv.verify(vk, bad, pi, n_cards=52)
    finally:
        c.release()
