"""The prover-round chain of tools/prover_chain.py on the CPU oracle (test infrastructure): from a `ChainInputs` object
to every commitment (affine wire), evaluation vector and intermediate polynomial the device chain produces.  Used live by
tests/test_gpu_prover_chain.py (n = 2^14) and by tests/golden/make_vectors_v2.py to freeze a small case (n = 2^12).

Follows uzkge/src/plonk/prover.rs:151-372 step by step: iFFT + hide_polynomial (helpers.rs:139-158), Lagrange commits with
blind factors (prover.rs:136-142, kzg_poly_commitment.rs:299-313), z_poly (helpers.rs:160-220), the quotient loop
(helpers.rs:284-656), split_t_and_commit (helpers.rs:1323-1408), evaluations, an r_poly-shaped combination, batch_prove
(pcs.rs:107-168)."""
import numpy as np

import bn254_py as opy
import oracle_c as oc
from util import affine_of

ints = oc.fr_to_ints


def add_blinds(coefs, blinds, n):
    """hide_polynomial on canonical ints (helpers.rs:139-158)."""
    c = list(coefs) + [0] * (n + len(blinds) - len(coefs))
    for i, bl in enumerate(blinds):
        c[i] = (c[i] + bl) % opy.R
        c[n + i] = (c[n + i] - bl) % opy.R
    return c


def commit_with_blinds(lagrange_wire, mono_pts, evals_wire, blinds_ints, n):
    """lagrange_pcs.commit(evals) then apply_blind_factors(blinds, n)."""
    cm = affine_of(oc.msm_pippenger(lagrange_wire, evals_wire, 0, 8))
    for i, bl in enumerate(blinds_ints):
        cm = opy.g1_add(cm, opy.g1_mul(mono_pts[i], bl))
        cm = opy.g1_add(cm, opy.g1_mul(mono_pts[n + i], (-bl) % opy.R))
    return cm


def pad(wire, length):
    out = np.zeros((length, 4), dtype=np.uint64)
    out[: wire.shape[0]] = wire
    return out


def aff_wire(pts):
    return oc.points_from_affine(pts)


def oracle_chain(c, shuffle=True):
    """c: ChainInputs.  Returns a dict of expected values (numpy arrays, wire format; commitments as affine [k, 8])."""
    from uzkge_amd.poly_commit import fr_to_int
    n, m = c.n, c.m
    out = {}
    mono_pts = {i: opy.wire_to_affine(c.mono_wire[i].tobytes()) for i in list(range(3)) + list(range(n, n + 3))}
    # ---- round 1
    evals9 = [c.w_evals[i] for i in range(5)] + [c.wsel_evals[i] for i in range(3)] + [c.pi_evals]
    blinds9 = [ints(c.blinds_w[i]) for i in range(5)] + [ints(c.blinds_wsel[i]) for i in range(3)] + [[]]
    polys = [add_blinds(ints(oc.ntt(evals9[i], inverse=True)), blinds9[i], n) for i in range(9)]
    out["cm_w_wsel"] = aff_wire([commit_with_blinds(c.lagrange_wire, mono_pts, evals9[i], blinds9[i], n) for i in range(8)])
    # ---- round 2
    g = fr_to_int(c.group_gen)
    group = oc.fr_from_ints([pow(g, i, opy.R) for i in range(n)])
    z_evals = oc.z_poly(c.w_evals, c.perm, group, c.k, c.beta, c.gamma)
    out["z_evals"] = z_evals
    polys.append(add_blinds(ints(oc.ntt(z_evals, inverse=True)), ints(c.blinds_z), n))
    out["cm_z"] = aff_wire([commit_with_blinds(c.lagrange_wire, mono_pts, z_evals, ints(c.blinds_z), n)])
    out["coefs"] = np.stack([pad(oc.fr_from_ints(p), n + 3) for p in polys])
    # ---- round 3
    cos = np.stack([oc.ntt(oc.mul_var(pad(oc.fr_from_ints(p), m), c.k[1])) for p in polys])
    out["coset_evals"] = cos
    vecs = np.concatenate([cos, c.tables])
    if not shuffle:
        for slot in list(range(5, 8)) + list(range(31, 56)):
            vecs[slot] = 0
    tq = oc.t_quotient(n, 6, vecs, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv)
    out["t_quotient"] = tq
    t = oc.mul_var(oc.ntt(tq, inverse=True), c.k1_inv)
    out["t"] = t
    t_int = ints(t)
    prev, rands = 0, ints(c.t_rands)
    cm_t, t_blinds = [], []
    for i in range(5):
        chunk = t_int[i * n:(i + 1) * n] + [rands[i]] if i < 4 else t_int[4 * n:5 * n + 2]
        chunk[0] = (chunk[0] - prev) % opy.R
        prev = rands[i]
        fold = chunk[:n]
        blinds = [(-x) % opy.R for x in chunk[n:]]
        for j, bl in enumerate(blinds):
            fold[j] = (fold[j] - bl) % opy.R
        t_blinds.append(oc.fr_from_ints(blinds))
        cm_t.append(commit_with_blinds(c.lagrange_wire, mono_pts, oc.ntt(oc.fr_from_ints(fold)), blinds, n))
        polys.append(chunk)
    out["cm_t"] = aff_wire(cm_t)
    out["t_blinds"] = t_blinds
    # ---- round 4
    out["evals_zeta"] = np.stack([oc.poly_eval(oc.fr_from_ints(polys[j]), c.zeta) for j in range(10)])
    out["z_eval_zeta_omega"] = oc.poly_eval(oc.fr_from_ints(polys[9]), c.zeta_omega).reshape(1, 4)
    # ---- round 5
    order = [9, 10, 11, 12, 13, 14, 0, 1, 2, 3, 4, 5]
    r = [0] * (n + 3)
    for s_k, idx in zip(ints(c.r_scalars), order):
        for j, v in enumerate(polys[idx][: n + 3]):
            r[j] = (r[j] + s_k * v) % opy.R
    out["r"] = oc.fr_from_ints(r)
    stack = np.stack([pad(oc.fr_from_ints(p), n + 8) for p in polys] + [pad(oc.fr_from_ints(r), n + 8)])
    cm_q, q_blinds, open_evals = [], [], []
    for pset, point in ((stack, c.zeta), (stack[9:10], c.zeta_omega)):
        q, ev, rem_zero = oc.open_quotient(pset, point, c.alpha_open)
        assert rem_zero
        qi = ints(q)
        assert not any(qi[n + 2:]) and qi[n + 1] != 0          # degree n + 1: max_power_of_2 = n, two blinds
        blinds = [(-x) % opy.R for x in qi[n:n + 2]]
        fold = qi[:n]
        for j, bl in enumerate(blinds):
            fold[j] = (fold[j] - bl) % opy.R
        open_evals.append(ev)
        q_blinds.append(oc.fr_from_ints(blinds))
        cm_q.append(commit_with_blinds(c.lagrange_wire, mono_pts, oc.ntt(oc.fr_from_ints(fold)), blinds, n))
    out["cm_q"] = aff_wire(cm_q)
    out["q_blinds"] = q_blinds
    out["open_evals_zeta"], out["open_evals_zeta_omega"] = open_evals
    return out
