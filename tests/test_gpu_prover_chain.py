"""A checked prover-round chain: every device-resident step one 52-card proof issues (tools/prover_chain.py, the
stand-in for BASELINE config #4 -- uzkge/src/plonk/prover.rs:88-394 cannot run here: no Rust toolchain), with each
commitment, evaluation vector and intermediate polynomial compared against the same chain on the CPU oracle.
Circuit tables and witness are synthetic (random elements of the real shapes: n = 2^14, 6n = 98304); the SRS files are
the reference's (`lagrange-srs-16384.bin`, `srs-padding.bin`), so every commitment is an MSM over reference bases."""
import os
import sys

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import affine_of

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def _add_blinds(coefs, blinds, n):
    """hide_polynomial on canonical ints (helpers.rs:139-158)."""
    c = list(coefs) + [0] * (n + len(blinds) - len(coefs))
    for i, bl in enumerate(blinds):
        c[i] = (c[i] + bl) % opy.R
        c[n + i] = (c[n + i] - bl) % opy.R
    return c


def _commit_with_blinds(lagrange_wire, mono_pts, evals_wire, blinds_ints, n):
    """prover.rs:136-142: lagrange_pcs.commit(evals) then apply_blind_factors(blinds, n) (kzg_poly_commitment.rs:299-313)."""
    cm = affine_of(oc.msm_pippenger(lagrange_wire, evals_wire, 0, 8))
    for i, bl in enumerate(blinds_ints):
        cm = opy.g1_add(cm, opy.g1_mul(mono_pts[i], bl))
        cm = opy.g1_add(cm, opy.g1_mul(mono_pts[n + i], (-bl) % opy.R))
    return cm


def _pad(wire, length):
    out = np.zeros((length, 4), dtype=np.uint64)
    out[: wire.shape[0]] = wire
    return out


def test_prover_round_chain_matches_oracle_chain(gpu):
    from prover_chain import HIDE, ProverChain
    c = ProverChain(n=1 << 14, seed=11)
    try:
        o = c.run()
        n, m = c.n, c.m
        mono_pts = {i: opy.wire_to_affine(c.mono_wire[i].tobytes()) for i in list(range(3)) + list(range(n, n + 3))}
        ints = oc.fr_to_ints
        # ---- round 1: coefficient polynomials (hidden) and the eight commitments
        evals9 = [c.w_evals[i] for i in range(5)] + [c.wsel_evals[i] for i in range(3)] + [c.pi_evals]
        blinds9 = [ints(c.blinds_w[i]) for i in range(5)] + [ints(c.blinds_wsel[i]) for i in range(3)] + [[]]
        polys = []                                              # hidden coefficient polynomials, canonical ints
        dev_coefs = _host(c.d_coefs).reshape(10, m, 4)
        for i in range(9):
            co = _add_blinds(ints(oc.ntt(evals9[i], inverse=True)), blinds9[i], n)
            polys.append(co)
            assert ints(dev_coefs[i, : n + 3]) == (co + [0] * 3)[: n + 3], f"polynomial {i}"
            assert not dev_coefs[i, n + 3:].any()
        for i in range(8):
            assert affine_of(o["cm_w_wsel"][i]) == _commit_with_blinds(c.lagrange_wire, mono_pts, evals9[i], blinds9[i], n), f"commitment {i}"
        # ---- round 2: z
        group = oc.fr_from_ints([pow(pc_int(c.group_gen), i, opy.R) for i in range(n)])
        z_evals = oc.z_poly(c.w_evals, c.perm, group, c.k, c.beta, c.gamma)
        assert np.array_equal(_host(c.d_z), z_evals)
        z_co = _add_blinds(ints(oc.ntt(z_evals, inverse=True)), ints(c.blinds_z), n)
        polys.append(z_co)
        assert ints(dev_coefs[9, : n + 3]) == z_co[: n + 3]
        assert affine_of(o["cm_z"][0]) == _commit_with_blinds(c.lagrange_wire, mono_pts, z_evals, ints(c.blinds_z), n)
        # ---- round 3: coset evaluations, quotient, t
        cos = np.stack([oc.ntt(oc.mul_var(_pad(oc.fr_from_ints(p), m), c.k[1])) for p in polys])
        assert np.array_equal(_host(c.d_coset).reshape(10, m, 4), cos)
        vecs = np.concatenate([cos, c.tables])                  # slot order UZK_TQ_*: 10 fresh vectors then the 46 tables
        want_tq = oc.t_quotient(n, 6, vecs, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv)
        assert np.array_equal(_host(c.d_tq), want_tq)
        t = oc.mul_var(oc.ntt(want_tq, inverse=True), c.k1_inv)
        assert np.array_equal(_host(c.d_t), t)
        t_int = ints(t)
        prev, rands = 0, ints(c.t_rands)
        for i in range(5):
            chunk = t_int[i * n:(i + 1) * n] + [rands[i]] if i < 4 else t_int[4 * n:5 * n + 2]
            chunk[0] = (chunk[0] - prev) % opy.R
            prev = rands[i]
            fold = chunk[:n]
            blinds = [(-x) % opy.R for x in chunk[n:]]
            for j, bl in enumerate(blinds):
                fold[j] = (fold[j] - bl) % opy.R
            assert ints(o["t_blinds"][i]) == blinds
            ev = oc.ntt(oc.fr_from_ints(fold))
            assert affine_of(o["cm_t"][i]) == _commit_with_blinds(c.lagrange_wire, mono_pts, ev, blinds, n), f"t chunk {i}"
            polys.append(chunk)
        # ---- round 4: evaluations
        for j in range(10):
            assert np.array_equal(o["evals_zeta"][j], oc.poly_eval(oc.fr_from_ints(polys[j]), c.zeta)), f"evaluation {j}"
        assert np.array_equal(o["z_eval_zeta_omega"][0], oc.poly_eval(oc.fr_from_ints(polys[9]), c.zeta_omega))
        # ---- round 5: r(X), openings
        order = [9, 10, 11, 12, 13, 14, 0, 1, 2, 3, 4, 5]
        rs = ints(c.r_scalars)
        r = [0] * (n + 3)
        for s_k, idx in zip(rs, order):
            for j, v in enumerate(polys[idx][: n + 3]):
                r[j] = (r[j] + s_k * v) % opy.R
        assert ints(_host(c.d_r)[: n + 3]) == r
        stack = np.stack([_pad(oc.fr_from_ints(p), n + 8) for p in polys] + [_pad(oc.fr_from_ints(r), n + 8)])
        for which, (pset, point) in enumerate(((stack, c.zeta), (stack[9:10], c.zeta_omega))):
            q, ev, rem_zero = oc.open_quotient(pset, point, c.alpha_open)
            assert rem_zero
            assert np.array_equal(o["open_evals_zeta" if which == 0 else "open_evals_zeta_omega"], ev)
            qi = ints(q)
            assert not any(qi[n + 2:]) and qi[n + 1] != 0
            blinds = [(-x) % opy.R for x in qi[n:n + 2]]
            fold = qi[:n]
            for j, bl in enumerate(blinds):
                fold[j] = (fold[j] - bl) % opy.R
            assert ints(o["q_blinds"][which]) == blinds
            ev_q = oc.ntt(oc.fr_from_ints(fold))
            assert affine_of(o["cm_q"][which]) == _commit_with_blinds(c.lagrange_wire, mono_pts, ev_q, blinds, n), f"opening {which}"
    finally:
        c.release()


def pc_int(row):
    from uzkge_amd.poly_commit import fr_to_int
    return fr_to_int(row)


def test_quotient_without_shuffle_vectors(gpu):
    """A circuit without the "shuffle" feature (zmatchmaking, helpers.rs:437 #[cfg(feature = "shuffle")]): the 28 vectors
    of terms 12..18 are passed as NULL; the result equals the full formula with those vectors zero."""
    from prover_chain import ProverChain
    c = ProverChain(n=4096, seed=5, shuffle=False, precompute=False)
    try:
        c.run()
        n, m = c.n, c.m
        assert sum(1 for p in c.tq_ptrs if not p) == 28
        cos = _host(c.d_coset).reshape(10, m, 4).copy()
        vecs = np.concatenate([cos, c.tables])
        for slot in list(range(5, 8)) + list(range(31, 56)):
            vecs[slot] = 0
        want = oc.t_quotient(n, 6, vecs, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv)
        assert np.array_equal(_host(c.d_tq), want)
        # a partial set of shuffle vectors is an argument error, not a silent zero
        from uzkge_amd import UzkgeError
        bad = list(c.tq_ptrs)
        bad[5] = c.d_coset.data_ptr()
        with pytest.raises(UzkgeError):
            gpu.t_quotient_device(n, 6, bad, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv,
                                  c.d_tq.data_ptr())
    finally:
        c.release()


def test_lincomb_and_hide_primitives(gpu):
    import torch
    n = 1000
    from util import rand_fr_wire
    polys = [rand_fr_wire(ln, 40 + i) for i, ln in enumerate((n, n - 7, 3, n + 5))]
    scal = rand_fr_wire(4, 50)
    d = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
    out = torch.empty((n + 5, 4), dtype=torch.int64, device="cuda")
    gpu.poly_lincomb_device([t.data_ptr() for t in d], [p.shape[0] for p in polys], scal, out.data_ptr(), n + 5)
    want = [0] * (n + 5)
    si = oc.fr_to_ints(scal)
    for s_k, p in zip(si, polys):
        for j, v in enumerate(oc.fr_to_ints(p)):
            want[j] = (want[j] + s_k * v) % opy.R
    assert oc.fr_to_ints(_host(out)) == want
    co = rand_fr_wire(n + 3, 60); co[n:] = 0
    bl = rand_fr_wire(3, 61)
    dco = torch.from_numpy(co.view(np.int64)).cuda()
    gpu.hide_polynomial_device(dco.data_ptr(), n + 3, bl, n)
    assert oc.fr_to_ints(_host(dco)) == _add_blinds(oc.fr_to_ints(co[:n]), oc.fr_to_ints(bl), n)
