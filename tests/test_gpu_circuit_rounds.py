"""uzk_circuit_* / uzk_prover_* / uzk_prove_round1..5 (include/uzkge_gpu.h): circuit residency that follows the reference's
parameter lifecycle, and the five rounds as the ONE implementation every host language drives.

The reference overwrites `q_shuffle_public_key_polys` / `_coset_evals` in place once per game
(`refresh_prover_params_public_key`, shuffle/src/gen_params/params.rs:57-129, called from shuffle/src/sdk.rs:143-157) and then
proves with the same `PlonkProverParams` (sdk.rs:196-214).  A device-resident circuit has to follow: here a satisfiable circuit is
proven, its twelve public-key tables are replaced (`uzk_circuit_refresh_tables`: the refresh loop on the device; and
`uzk_circuit_update_tables`: the upload form), proven again, and BOTH proofs must be accepted by the verifier restatement that
accepts the reference's golden proof (tests/plonk_golden_verifier.py) -- each under its own key, neither under the other's."""
import os
import sys

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
import plonk_verifier_oracle as pv
from util import GOLDEN, affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _vk_of(b, c, inp, table_cms):
    import prover_chain as pch
    n = inp.n
    omega = pv._ints(inp.group_gen)[0]
    ninv = pow(n, -1, opy.R)
    g = pv._ints(inp.anemoi_g)[0]
    return {"cm_q": table_cms[pch.T_Q:pch.T_Q + 9], "cm_s": table_cms[pch.T_S:pch.T_S + 5], "cm_qb": table_cms[pch.T_QB],
            "cm_prk": table_cms[pch.T_QPRK:pch.T_QPRK + 4], "cm_q_ecc": table_cms[pch.T_QECC],
            "cm_shuffle_generator": table_cms[pch.T_QG:pch.T_QG + 12], "cm_shuffle_public_key": table_cms[pch.T_QPK:pch.T_QPK + 12],
            "anemoi_g": g, "anemoi_g_inv": pow(g, -1, opy.R), "k": pv._ints(inp.k), "edwards_a": pv._ints(inp.edwards_a)[0],
            "root": omega, "cs_size": n, "pi_root_powers": [pow(omega, j, opy.R) for j in range(8)],
            "pi_lagrange": [pow(omega, j, opy.R) * ninv % opy.R for j in range(8)]}


def _prove(c, inp, vk):
    """One non-interactive proof from the chain (challenges from the reference's transcript), packed as the reference's PlonkProof."""
    import prover_chain as pch
    from test_gpu_plonk_verifier import _FiatShamir
    n = inp.n
    plan = pch.eval_plan(True)
    pi = pv._ints(inp.pi_evals[:8])
    fs = _FiatShamir(vk, pi, 52, plan)
    c.fs = fs
    at = {(kind, idx, pt): i for i, (kind, idx, pt) in enumerate(plan)}

    def evals_of(rows):
        v = pv._ints(rows)
        return {"w": [v[at[("c", i, 0)]] for i in range(5)], "s": [v[at[("t", pch.T_S + i, 0)]] for i in range(4)],
                "prk3": v[at[("t", pch.T_QPRK + 2, 0)]], "prk4": v[at[("t", pch.T_QPRK + 3, 0)]], "z_omega": v[at[("c", 9, 1)]],
                "w_omega": [v[at[("c", i, 1)]] for i in range(3)], "q_ecc": v[at[("t", pch.T_QECC, 0)]],
                "wsel": [v[at[("c", 5 + i, 0)]] for i in range(3)]}
    chd = lambda: {"alpha": fs.ch["alpha"], "beta": fs.ch["beta"], "gamma": fs.ch["gamma"], "zeta": fs.ch["zeta"], "anemoi_g": vk["anemoi_g"],
                   "edwards_a": vk["edwards_a"]}
    c.r_scalar_hook = lambda rows: oc.fr_from_ints(pv.r_scalars(chd(), vk["k"], n, evals_of(rows), True))
    o = c.run()
    ev = evals_of(o["evals"])
    return {"cm_w": [affine_of(j) for j in o["cm_w_wsel"][:5]], "cm_wsel": [affine_of(j) for j in o["cm_w_wsel"][5:8]],
            "cm_t": [affine_of(j) for j in o["cm_t"]], "cm_z": affine_of(o["cm_z"][0]), "prk3": ev["prk3"], "prk4": ev["prk4"],
            "w": ev["w"], "w_omega": ev["w_omega"], "z_omega": ev["z_omega"], "s": ev["s"], "q_ecc": ev["q_ecc"], "wsel": ev["wsel"],
            "open_zeta": affine_of(o["cm_q"][0]), "open_zeta_omega": affine_of(o["cm_q"][1])}


def test_proofs_across_a_public_key_refresh_are_each_accepted_under_their_own_key(gpu):
    import plonk_golden_verifier as gv
    import prover_chain as pch
    from uzkge_amd import backend as b
    n = 1 << 14
    inp = pv.make_satisfiable(pch.ChainInputs(n, 21), seed=4)
    pi = pv._ints(inp.pi_evals[:8])
    c = pch.ProverChain(inputs=inp, precompute=False)
    try:
        table_cms = [affine_of(j) for j in b.msm_batch(c.srs, b.ntt_batch(inp.table_polys))]
        vk_a = _vk_of(b, c, inp, table_cms)
        proof_a = _prove(c, inp, vk_a)
        assert gv.verify(vk_a, proof_a, pi, n_cards=52)
        # ---- the next game's joint key: twelve new public-key selector vectors.  The witness stays the same, so the new vectors
        # agree with the old ones on the rows that carry the shuffle gadget (q_ecc = 1: there the wires were solved against them)
        # and are fresh everywhere else -- different polynomials, different commitments, different coset tables.
        q_ecc = pv._ints(oc.ntt(np.ascontiguousarray(inp.table_polys[pch.T_QECC])))
        gadget = np.array([v == 1 for v in q_ecc])
        assert 500 < gadget.sum() < n // 8
        pk_evals = np.stack([oc.ntt(np.ascontiguousarray(inp.table_polys[pch.T_QPK + t])) for t in range(12)])
        fresh = rand_fr_wire(12 * n, 77).reshape(12, n, 4)
        pk_evals_b = np.where(gadget[None, :, None], pk_evals, fresh)
        # the refresh loop on the device (params.rs:88-121): iFFT -> coset FFT(6n) -> Lagrange commit, installed in the circuit
        cms, polys_b, lens_b, coset_b = c.circuit.refresh_tables(b.CS_QPK, pk_evals_b, want_polys=True, want_coset=True)
        for t in range(12):
            assert np.array_equal(polys_b[t], oc.ntt(np.ascontiguousarray(pk_evals_b[t]), inverse=True)), t
            assert int(lens_b[t]) == b.trimmed_len(polys_b[t])
            assert affine_of(cms[t]) == oc.jac_to_affine_ints(oc.msm_pippenger(inp.lagrange_wire, np.ascontiguousarray(pk_evals_b[t]), 0, 4)), t
        pad = np.zeros((6 * n, 4), dtype=np.uint64); pad[:n] = polys_b[5]
        assert np.array_equal(coset_b[5], oc.ntt(oc.mul_var(pad, inp.k[1]), threads=4))
        vk_b = dict(vk_a, cm_shuffle_public_key=[affine_of(j) for j in cms])
        assert vk_b["cm_shuffle_public_key"] != vk_a["cm_shuffle_public_key"]
        proof_b = _prove(c, inp, vk_b)
        assert gv.verify(vk_b, proof_b, pi, n_cards=52)            # the new tables are what the prover used
        assert not gv.verify(vk_a, proof_b, pi, n_cards=52)        # ... and not the stale ones
        assert not gv.verify(vk_b, proof_a, pi, n_cards=52)
        assert proof_b["cm_w"] == proof_a["cm_w"] and proof_b["cm_t"] != proof_a["cm_t"]     # same witness, another quotient
        # ---- the upload form (what rust/uzkge-glue/gpu_prover.rs calls when the verifier key's cm_shuffle_public_key_vec changed):
        # back to key A from its coefficient forms; the proof is again A's, byte for byte
        c.circuit.update_tables(b.CS_QPK, [inp.table_polys[pch.T_QPK + t] for t in range(12)])
        proof_a2 = _prove(c, inp, vk_a)
        assert proof_a2 == proof_a
    finally:
        c.release()


def _round_inputs(inp, lanes):
    """Per-lane inputs for a lockstep batch: lane 0 is `inp` itself, the others differ in witness, blinds and challenges."""
    import prover_chain as pch
    out = []
    for lane in range(lanes):
        x = pch.ChainInputs(inp.n, inp.seed + 1000 * lane) if lane else inp
        out.append(x)
    return out


def _run_rounds(b, circuit, prover, lanes_in, shuffle=True, with_wsel=True):
    import prover_chain as pch
    B = len(lanes_in)
    n = lanes_in[0].n
    cat = lambda f: np.concatenate([np.ascontiguousarray(f(x), dtype=np.uint64).reshape(-1, 4) for x in lanes_in])
    hiding = list(pch.HIDE_W) + ([pch.HIDE_WSEL] * 3 if with_wsel else [])
    bl = cat(lambda x: np.concatenate([x.blinds_w, x.blinds_wsel]) if with_wsel else x.blinds_w)
    o = {}
    o["cm1"] = prover.round1(circuit, cat(lambda x: x.w_evals).reshape(B, 5 * n, 4), cat(lambda x: x.wsel_evals).reshape(B, 3 * n, 4) if with_wsel else None,
                             np.arange(8, dtype=np.uint32), cat(lambda x: x.pi_evals[:8]).reshape(B, 8, 4), hiding, bl)
    o["cm_z"] = prover.round2(cat(lambda x: x.beta), cat(lambda x: x.gamma), cat(lambda x: x.blinds_z))
    o["cm_t"] = prover.round3(cat(lambda x: x.alpha), cat(lambda x: x.t_rands))
    o["evals"] = prover.round4(cat(lambda x: x.zeta), shuffle)
    nr = len(pch.r_plan(shuffle))
    o["cm_q"] = prover.round5(cat(lambda x: x.r_scalars[:nr]), cat(lambda x: x.alpha_open), cat(lambda x: x.alpha_open2))
    return o


def _circuit_of(b, inp, shuffle=True, precompute=False, synthetic=True):
    """synthetic: random table polynomials no witness satisfies (tools/prover_chain.py) -- round 3 reads t as its expected length"""
    import prover_chain as pch
    return b.Circuit(inp.n, inp.lagrange_wire, inp.bases[inp.n:], inp.perm, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a,
                     [inp.table_polys[i] for i in range(pch.N_TABLES)], shuffle=shuffle, precompute=precompute, synthetic=synthetic)


@pytest.mark.parametrize("precompute", [0, 1, 11])
def test_lockstep_batch_equals_single_proofs(gpu, precompute):
    """uzk_prover_create(n, 3): three different witnesses / blinds / challenges over one circuit advance in lockstep (commits of
    24 vectors, transforms of 30); every lane must equal the proof a batch-1 prover makes of it.  precompute = 1: the batch commits
    over the 15-bit window table (one shared bucket set per vector, general pipeline), the single proofs over the 8-bit one (one
    workgroup per window): two MSM pipelines, two tables, the same points."""
    import prover_chain as pch
    b = gpu
    n = 1 << 12
    inp = pch.ChainInputs(n, 31)
    lanes = _round_inputs(inp, 3)
    cir = _circuit_of(b, inp, precompute=precompute)
    p3, p1 = b.Prover(n, 3), b.Prover(n, 1, shared=False)
    try:
        o3 = _run_rounds(b, cir, p3, lanes)
        snap3 = {w: [p3.download(w, lane) for lane in range(3)] for w in (b.PB_COEFS, b.PB_T, b.PB_R, b.PB_Q)}
        for lane, x in enumerate(lanes):
            o1 = _run_rounds(b, cir, p1, [x])
            for key, per in (("cm1", 8), ("cm_z", 1), ("cm_t", 5), ("cm_q", 2)):
                assert [affine_of(j) for j in o3[key][lane * per:(lane + 1) * per]] == [affine_of(j) for j in o1[key]], (lane, key)
            assert np.array_equal(o3["evals"][lane * 19:(lane + 1) * 19], o1["evals"]), lane
            for w in snap3:
                assert np.array_equal(snap3[w][lane], p1.download(w)), (lane, w)
    finally:
        p3.destroy(); p1.destroy(); cir.release()


def test_tables_are_copy_on_write_for_a_proof_in_flight(gpu):
    """A refresh between round 1 and round 5 of a proof does not change that proof (it keeps the tables it started with); the next
    proof sees the new tables."""
    import prover_chain as pch
    b = gpu
    n = 1 << 12
    inp = pch.ChainInputs(n, 41)
    cir = _circuit_of(b, inp)
    pr = b.Prover(n, 1)
    try:
        base = _run_rounds(b, cir, pr, [inp])
        new_pk = [rand_fr_wire(n, 900 + t) for t in range(12)]
        # same proof again, tables swapped after round 2
        hiding = list(pch.HIDE_W) + [pch.HIDE_WSEL] * 3
        pr.round1(cir, inp.w_evals.reshape(1, 5 * n, 4), inp.wsel_evals.reshape(1, 3 * n, 4), np.arange(8, dtype=np.uint32), inp.pi_evals[:8].reshape(1, 8, 4),
                  hiding, np.concatenate([inp.blinds_w, inp.blinds_wsel]))
        pr.round2(inp.beta, inp.gamma, inp.blinds_z)
        old_ptr = cir.table(b.CS_QPK, coset=True)[0]
        cir.update_tables(b.CS_QPK, new_pk)
        assert cir.table(b.CS_QPK, coset=True)[0] != old_ptr
        cm_t = pr.round3(inp.alpha, inp.t_rands)
        ev = pr.round4(inp.zeta)
        cm_q = pr.round5(inp.r_scalars, inp.alpha_open, inp.alpha_open2)
        assert [affine_of(j) for j in cm_t] == [affine_of(j) for j in base["cm_t"]]
        assert np.array_equal(ev, base["evals"]) and [affine_of(j) for j in cm_q] == [affine_of(j) for j in base["cm_q"]]
        # the next proof runs on the new tables: its quotient differs, its round-1 / round-2 commitments do not
        nxt = _run_rounds(b, cir, pr, [inp])
        assert [affine_of(j) for j in nxt["cm1"]] == [affine_of(j) for j in base["cm1"]]
        assert [affine_of(j) for j in nxt["cm_t"]] != [affine_of(j) for j in base["cm_t"]]
        # and equals a circuit created with the new tables from the start
        inp2 = pch.ChainInputs(n, 41)
        for t in range(12):
            inp2.table_polys[pch.T_QPK + t] = new_pk[t]
        cir2 = _circuit_of(b, inp2)
        try:
            fresh = _run_rounds(b, cir2, pr, [inp2])
            for key in ("cm_t", "cm_q"):
                assert [affine_of(j) for j in nxt[key]] == [affine_of(j) for j in fresh[key]], key
            assert np.array_equal(nxt["evals"], fresh["evals"])
        finally:
            cir2.release()
    finally:
        pr.destroy(); cir.release()


def test_circuit_without_wire_selectors_and_with_built_coset_quotient(gpu):
    """zmatchmaking's shape as the reference builds it: no "shuffle" feature, no wire selectors (prover.rs:177-192 is cfg'd out) --
    five commitments in round 1, seven polynomial slots.  Also: slot 20 left to the library is k[1] * g_m^i."""
    import prover_chain as pch
    b = gpu
    n = 1 << 12
    inp = pch.ChainInputs(n, 51)
    polys = [inp.table_polys[i] for i in range(pch.N_TABLES)]
    polys[pch.T_CQ] = None
    cir = b.Circuit(n, inp.lagrange_wire, inp.bases[n:], inp.perm, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a, polys, shuffle=False, synthetic=True)
    pr = b.Prover(n, 1)
    try:
        m = 6 * n
        ptr, ln = cir.table(b.CS_COSET_QUOTIENT, coset=True)
        got = b.dev_download(ptr, (m, 4))
        gm = pv._ints(b.domain_group_gen(m))[0]
        k1 = pv._ints(inp.k[1:2])[0]
        idx = [0, 1, 2, 5, m // 3, m - 1]
        assert pv._ints(got[idx]) == [k1 * pow(gm, i, opy.R) % opy.R for i in idx]
        assert cir.table(b.CS_COSET_QUOTIENT)[1] == 2                       # the polynomial X
        o5 = _run_rounds(b, cir, pr, [inp], shuffle=False, with_wsel=False)
        o8 = _run_rounds(b, cir, pr, [inp], shuffle=False, with_wsel=True)
        assert o5["cm1"].shape[0] == 5 and o8["cm1"].shape[0] == 8
        for key in ("cm_z", "cm_t", "cm_q"):                                # without the feature the selectors enter nothing after round 1
            assert [affine_of(j) for j in o5[key]] == [affine_of(j) for j in o8[key]], key
        assert [affine_of(j) for j in o5["cm1"]] == [affine_of(j) for j in o8["cm1"][:5]]
        assert np.array_equal(o5["evals"], o8["evals"]) and o5["evals"].shape[0] == 15
        # three lanes in lockstep without wire selectors, on a prover whose idle slots hold what proofs WITH selectors left there:
        # round 3 transforms all ten slots of every lane in one launch sequence, seven of them in use
        lanes = _round_inputs(inp, 3)
        p3 = b.Prover(n, 3)
        try:
            _run_rounds(b, cir, p3, lanes, shuffle=False, with_wsel=True)
            o3 = _run_rounds(b, cir, p3, lanes, shuffle=False, with_wsel=False)
            for lane, x in enumerate(lanes):
                o1 = _run_rounds(b, cir, pr, [x], shuffle=False, with_wsel=False)
                for key, per in (("cm1", 5), ("cm_z", 1), ("cm_t", 5), ("cm_q", 2)):
                    assert [affine_of(j) for j in o3[key][lane * per:(lane + 1) * per]] == [affine_of(j) for j in o1[key]], (lane, key)
                assert np.array_equal(o3["evals"][lane * 15:(lane + 1) * 15], o1["evals"]), lane
        finally:
            p3.destroy()
    finally:
        pr.destroy(); cir.release()


def test_public_input_polynomial_and_argument_errors(gpu):
    """pi_poly (helpers.rs:111-131): the online values land on their constraint indices, a repeated index takes its FIRST value
    (find_position); rounds out of order, a foreign group generator, a prover of another size are argument errors."""
    import prover_chain as pch
    from uzkge_amd import UzkgeError
    from uzkge_amd import _native as N
    b = gpu
    n = 1 << 12
    inp = pch.ChainInputs(n, 61)
    cir = _circuit_of(b, inp)
    pr = b.Prover(n, 1, shared=False)
    try:
        idx = np.array([7, 3, 4000, 3, 19], dtype=np.uint32)
        val = rand_fr_wire(5, 62)
        hiding = list(pch.HIDE_W) + [pch.HIDE_WSEL] * 3
        args = (inp.w_evals.reshape(1, 5 * n, 4), inp.wsel_evals.reshape(1, 3 * n, 4))
        pr.round1(cir, *args, idx, val.reshape(1, 5, 4), hiding, np.concatenate([inp.blinds_w, inp.blinds_wsel]))
        evals = np.zeros((n, 4), dtype=np.uint64)
        for j in (4, 2, 1, 0):                                              # index 3 appears twice: position 1 wins over position 3
            evals[idx[j]] = val[j]
        coefs = pr.download(b.PB_COEFS).reshape(10, 6 * n, 4)
        assert np.array_equal(coefs[8, :n], oc.ntt(evals, inverse=True)) and not coefs[8, n:].any()
        with pytest.raises(UzkgeError) as e:                                # round 3 before round 2
            pr.round3(inp.alpha, inp.t_rands)
        assert e.value.code == N.UZK_ERR_PARAMETER
        pr.round2(inp.beta, inp.gamma, inp.blinds_z)                        # ... which did not disturb the proof in flight
        with pytest.raises(UzkgeError):
            pr.round1(cir, *args, np.array([n], dtype=np.uint32), val[:1].reshape(1, 1, 4), hiding, np.concatenate([inp.blinds_w, inp.blinds_wsel]))
        with pytest.raises(UzkgeError):                                     # hiding degrees the split cannot fold (sum 8)
            pr.round1(cir, *args, idx, val.reshape(1, 5, 4), [2, 2, 2, 1, 1, 2, 2, 2], np.concatenate([inp.blinds_w, inp.blinds_wsel]))
        other = b.Prover(2 * n, 1)
        try:
            with pytest.raises(UzkgeError) as e:
                other.round1(cir, np.zeros((1, 10 * n, 4), dtype=np.uint64), np.zeros((1, 6 * n, 4), dtype=np.uint64), idx, val.reshape(1, 5, 4), hiding,
                             np.concatenate([inp.blinds_w, inp.blinds_wsel]))
            assert "the circuit has n" in str(e.value)
        finally:
            other.destroy()
        with pytest.raises(UzkgeError) as e:                                # another root of unity than the library's: FFTError, nothing built
            b.Circuit(n, inp.lagrange_wire, inp.bases[n:], inp.perm, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a,
                      [inp.table_polys[i] for i in range(pch.N_TABLES)], group_gen=np.asarray(oc.fr_mul(b.domain_group_gen(n), b.domain_group_gen(n))).reshape(-1)[:4])
        assert e.value.code == N.UZK_ERR_FFT
        with pytest.raises(UzkgeError):
            cir.update_tables(40, [inp.table_polys[0]] * 7)                 # slots 40 .. 47 of 46
    finally:
        pr.destroy(); cir.release()


@pytest.mark.parametrize("n", [4096, 16384])
def test_refresh_tables_on_the_reference_srs(gpu, n):
    """The refresh / indexer loop as one device call (`uzk_circuit_refresh_tables`) on the reference's own parameter files: for
    selector vectors whose polynomials have degree <= 2050 the Lagrange commitment the device returns equals the MONOMIAL
    commit over srs-padding.bin (the commit closure's two branches, indexer.rs:284-299) -- NTT, trimming and MSM pinned together
    on reference-held data."""
    import prover_chain as pch
    from uzkge_amd import poly_commit as pc
    b = gpu
    inp = pch.ChainInputs(n, 71)
    cir = _circuit_of(b, inp)
    try:
        deg = 2051
        coefs = np.zeros((12, n, 4), dtype=np.uint64)
        coefs[:, :deg] = rand_fr_wire(12 * deg, 72).reshape(12, deg, 4)
        coefs[3] = 0                                                        # a selector that is identically zero (prk3 of the golden circuit)
        coefs[4, 1:] = 0                                                    # a constant
        evals = np.stack([oc.ntt(np.ascontiguousarray(coefs[t])) for t in range(12)])
        cms, polys, lens, _ = cir.refresh_tables(b.CS_QG, evals)
        mono = inp.mono_wire[:deg]
        for t in range(12):
            assert np.array_equal(polys[t], coefs[t]), t
            want_len = 0 if t == 3 else 1 if t == 4 else deg
            assert int(lens[t]) == want_len and cir.table(b.CS_QG + t)[1] == want_len
            want = oc.jac_to_affine_ints(oc.msm_pippenger(mono, np.ascontiguousarray(coefs[t, :deg]), 0, 4))
            assert affine_of(cms[t]) == want, t
        assert affine_of(cms[3]) is None
    finally:
        cir.release()


def test_preprocess_tables_without_a_circuit(gpu):
    """uzk_preprocess_tables: the indexer's per-table loop (indexer.rs:316-470) as one call over a plain registered Lagrange SRS --
    iFFT, trimmed lengths, coset FFT over the 6n domain, Lagrange commitments -- against the oracle."""
    b = gpu
    n = 4096
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = b.Srs.from_host(wire)
    try:
        evals = rand_fr_wire(5 * n, 81).reshape(5, n, 4)
        low = np.zeros((n, 4), dtype=np.uint64); low[:100] = rand_fr_wire(100, 82)
        evals[2] = oc.ntt(low)                                   # a polynomial of 100 coefficients
        k1 = rand_fr_wire(1, 83)[0]
        polys, lens, coset, cms = b.preprocess_tables_device(srs, evals, k1=k1, want_coset=True)
        for t in range(5):
            want = oc.ntt(np.ascontiguousarray(evals[t]), inverse=True)
            assert np.array_equal(polys[t], want), t
            assert int(lens[t]) == (100 if t == 2 else n)
            pad = np.zeros((6 * n, 4), dtype=np.uint64); pad[:n] = want
            assert np.array_equal(coset[t], oc.ntt(oc.mul_var(pad, k1), threads=4)), t
            assert affine_of(cms[t]) == oc.jac_to_affine_ints(oc.msm_pippenger(wire, np.ascontiguousarray(evals[t]), 0, 4)), t
    finally:
        srs.release()


def test_preprocess_tables_in_the_shapes_of_the_indexers_steps(gpu):
    """What the Rust hook in uzkge/src/plonk/indexer.rs does (rust/uzkge-glue/gpu.rs preprocess_tables): one call per step with
    5, 9, 1, 4, 1 and 12 tables -- boolean selector vectors (qb, q_ecc: a handful of ones), an all-zero vector (an unused
    selector trims to the zero polynomial) -- and, when verifier parameters were handed in, no commitments and no SRS handle."""
    b = gpu
    n = 4096
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = b.Srs.from_host(wire)
    k1 = rand_fr_wire(1, 91)[0]
    one = np.frombuffer(opy.to_mont(1, opy.R).to_bytes(32, "little"), dtype=np.uint64)
    try:
        for step, count in enumerate((5, 9, 1, 4, 1, 12)):
            evals = rand_fr_wire(count * n, 100 + step).reshape(count, n, 4)
            if count == 1:                                   # indicator vectors: ones on a few rows
                evals[:] = 0
                evals[0, [3, 17, 1000, n - 1]] = one
            if count == 9:
                evals[4] = 0                                 # the zero polynomial
            polys, lens, coset, cms = b.preprocess_tables_device(srs, evals, k1=k1, want_coset=True)
            polys2, lens2, coset2, none = b.preprocess_tables_device(None, evals, k1=k1, want_coset=True, want_commit=False)
            assert none is None and np.array_equal(polys, polys2) and np.array_equal(lens, lens2) and np.array_equal(coset, coset2)
            for t in range(count):
                want = oc.ntt(np.ascontiguousarray(evals[t]), inverse=True)
                assert np.array_equal(polys[t], want), (step, t)
                nz = np.nonzero(want.any(axis=1))[0]
                assert int(lens[t]) == (int(nz[-1]) + 1 if nz.size else 0), (step, t)
                pad = np.zeros((6 * n, 4), dtype=np.uint64); pad[:n] = want
                assert np.array_equal(coset[t], oc.ntt(oc.mul_var(pad, k1), threads=4)), (step, t)
                assert affine_of(cms[t]) == oc.jac_to_affine_ints(oc.msm_pippenger(wire, np.ascontiguousarray(evals[t]), 0, 4)), (step, t)
    finally:
        srs.release()


def test_provers_on_several_contexts_while_the_tables_are_being_swapped(gpu):
    """Three prover threads (one context and one prover each, one shared circuit) prove in a loop while the main thread keeps
    swapping the twelve public-key tables between two sets: every proof must be, as a whole, the proof of set A or the proof of
    set B (the snapshot a proof takes at round 1 is never torn, nothing is freed under a running kernel)."""
    import threading
    import prover_chain as pch
    b = gpu
    n = 1 << 12
    inp = pch.ChainInputs(n, 91)
    cir = _circuit_of(b, inp)
    set_a = [np.ascontiguousarray(inp.table_polys[pch.T_QPK + t]) for t in range(12)]
    set_b = [rand_fr_wire(n, 950 + t) for t in range(12)]
    pr0 = b.Prover(n, 1)
    want = {}
    try:
        for name, tables in (("b", set_b), ("a", set_a)):
            cir.update_tables(b.CS_QPK, tables)
            o = _run_rounds(b, cir, pr0, [inp])
            want[name] = ([affine_of(j) for j in o["cm_t"]], [affine_of(j) for j in o["cm_q"]], o["evals"].tobytes())
        assert want["a"] != want["b"]
        stop, seen, errors = threading.Event(), [], []

        def worker():
            try:
                ctx = b.ctx_create()
                b.ctx_set_current(ctx)
                pr = b.Prover(n, 1)
                try:
                    while not stop.is_set():
                        o = _run_rounds(b, cir, pr, [inp])
                        seen.append(([affine_of(j) for j in o["cm_t"]], [affine_of(j) for j in o["cm_q"]], o["evals"].tobytes()))
                finally:
                    pr.destroy()
                    b.ctx_set_current(0)
                    b.ctx_destroy(ctx)
            except Exception as e:        # surfaced by the main thread
                errors.append(e)
        threads = [threading.Thread(target=worker) for _ in range(3)]
        for t in threads:
            t.start()
        for i in range(40):
            cir.update_tables(b.CS_QPK, set_b if i % 2 == 0 else set_a)
        stop.set()
        for t in threads:
            t.join()
        assert not errors, errors
        assert len(seen) >= 6
        kinds = {"a" if s == want["a"] else "b" if s == want["b"] else "torn" for s in seen}
        assert kinds <= {"a", "b"}, kinds
    finally:
        pr0.destroy(); cir.release()


@pytest.mark.parametrize("n", [64, 1024])
def test_small_circuits_are_self_consistent(gpu, n):
    """Sizes below the reference's parameter files (its smallest Lagrange SRS has 4096 bases): the one-workgroup NTT, the 3 * 2^k
    path of the small quotient domain, the tiny-n MSM windows.  No oracle chain exists at these sizes (it reads the reference's
    SRS files), so the rounds are held to themselves: a lockstep pair over the window tables equals two single proofs without
    tables, the evaluations equal the oracle's Horner values of the coefficient polynomials read back from the device."""
    import prover_chain as pch
    b = gpu
    big = pch.ChainInputs(4096, 17)
    rng = np.random.default_rng(n)

    def lane(seed):
        x = pch.ChainInputs.__new__(pch.ChainInputs)
        x.n, x.m, x.seed = n, 6 * n, seed
        x.w_evals = rand_fr_wire(5 * n, seed).reshape(5, n, 4)
        x.wsel_evals = rand_fr_wire(3 * n, seed + 1).reshape(3, n, 4)
        x.pi_evals = np.zeros((n, 4), dtype=np.uint64); x.pi_evals[:8] = rand_fr_wire(8, seed + 2)
        sc = rand_fr_wire(16, seed + 3)
        x.beta, x.gamma, x.alpha, x.zeta, x.alpha_open, x.alpha_open2 = sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]
        x.blinds_w = rand_fr_wire(15, seed + 4).reshape(5, 3, 4); x.blinds_w[3:, 2] = 0
        x.blinds_wsel = rand_fr_wire(9, seed + 5).reshape(3, 3, 4); x.blinds_wsel[:, 2] = 0
        x.blinds_z, x.t_rands, x.r_scalars = rand_fr_wire(3, seed + 6), rand_fr_wire(5, seed + 7), rand_fr_wire(43, seed + 8)
        return x
    lanes = [lane(100), lane(200)]
    bases = np.ascontiguousarray(big.lagrange_wire[: n + 6])                    # any n + 6 valid points serve as commit bases here
    perm = rng.permutation(5 * n).astype(np.uint32)
    k, polys = rand_fr_wire(5, 9), [rand_fr_wire(n, 300 + i) for i in range(pch.N_TABLES)]
    g = rand_fr_wire(1, 10)[0]
    ginv = oc.fr_inv(g)
    mk = lambda pre: b.Circuit(n, bases[:n], bases[n:], perm, k, g, ginv, rand_fr_wire(1, 11)[0], polys, precompute=pre, synthetic=True)
    c0, c1 = mk(0), mk(1)
    p1, p2 = b.Prover(n, 1), b.Prover(n, 2)
    try:
        o2 = _run_rounds(b, c1, p2, lanes)
        coefs2 = [p2.download(b.PB_COEFS, lane_i).reshape(10, 6 * n, 4) for lane_i in range(2)]
        for i, x in enumerate(lanes):
            o1 = _run_rounds(b, c0, p1, [x])
            for key, per in (("cm1", 8), ("cm_z", 1), ("cm_t", 5), ("cm_q", 2)):
                assert [affine_of(j) for j in o2[key][i * per:(i + 1) * per]] == [affine_of(j) for j in o1[key]], (n, i, key)
            assert np.array_equal(o2["evals"][i * 19:(i + 1) * 19], o1["evals"])
            # w0(zeta) and z(zeta omega) by the oracle's Horner over the coefficients the device holds
            w0 = np.ascontiguousarray(coefs2[i][0, : n + 3])
            assert np.array_equal(oc.poly_eval(w0, x.zeta).reshape(4), o2["evals"][i * 19])
            zw = oc.fr_mul(x.zeta, b.domain_group_gen(n))
            assert np.array_equal(oc.poly_eval(np.ascontiguousarray(coefs2[i][9, : n + 3]), zw).reshape(4), o2["evals"][i * 19 + 11])
    finally:
        p1.destroy(); p2.destroy(); c0.release(); c1.release()
