"""The prover-round chain of tools/prover_chain.py on the CPU oracle (test infrastructure): from a `ChainInputs` object
to every commitment (affine wire), evaluation and intermediate polynomial the device chain produces.  Used live by
tests/test_gpu_prover_chain.py (n = 2^14 and 2^13) and by tests/golden/make_vectors_v3.py to freeze a small case (n = 2^12).

Follows uzkge/src/plonk/prover.rs:151-372 step by step: iFFT + hide_polynomial (helpers.rs:139-158), Lagrange commits with
blind factors (prover.rs:136-142, kzg_poly_commitment.rs:299-313), z_poly (helpers.rs:160-220), the quotient loop
(helpers.rs:284-656), split_t_and_commit with chunk = n + 2 (helpers.rs:1323-1408), the 19 evaluations of prover.rs:246-273,
an r_poly-shaped combination of 43 polynomials, the two batch_prove openings (pcs.rs:107-168).  Which polynomial goes where
comes from the same plans the device chain reads (tools/prover_chain.py: eval_plan, r_plan, open_plan) -- the arithmetic is
the oracle's own."""
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


def max_power_of_2(degree):
    """pcs.rs:139-145 / helpers.rs:1367-1373."""
    for i in range(degree, -1, -1):
        if i & (i - 1) == 0:
            return i
    return degree


def fold_and_commit(c, mono_pts, coefs_ints, degree):
    """The Lagrange branch shared by batch_prove (pcs.rs:137-166) and split_t_and_commit (helpers.rs:1366-1394)."""
    n = c.n
    npow = max_power_of_2(degree)
    assert npow == n
    blinds = [(-x) % opy.R for x in coefs_ints[npow:]]
    fold = list(coefs_ints[:npow]) + [0] * (npow - len(coefs_ints[:npow]))
    for j, bl in enumerate(blinds):
        fold[j] = (fold[j] - bl) % opy.R
    return commit_with_blinds(c.lagrange_wire, mono_pts, oc.ntt(oc.fr_from_ints(fold)), blinds, npow), blinds


def oracle_chain(c, shuffle=True):
    """c: ChainInputs.  Returns a dict of expected values (numpy arrays, wire format; commitments as affine [k, 8])."""
    from prover_chain import eval_plan, open_plan, r_plan
    from uzkge_amd.poly_commit import fr_to_int
    n, m = c.n, c.m
    out = {}
    mono_pts = {i: opy.wire_to_affine(c.mono_wire[i].tobytes()) for i in list(range(3)) + list(range(n, n + 3))}
    # ---- setup: the circuit's coset tables from its coefficient polynomials (coset FFT over the 6n domain)
    tables = np.stack([oc.ntt(oc.mul_var(pad(c.table_polys[i], m), c.k[1]), threads=4) for i in range(c.table_polys.shape[0])])
    out["tables"] = tables
    tpolys = [ints(c.table_polys[i]) for i in range(c.table_polys.shape[0])]
    # ---- round 1
    evals9 = [c.w_evals[i] for i in range(5)] + [c.wsel_evals[i] for i in range(3)] + [c.pi_evals]
    from prover_chain import HIDE_W, HIDE_WSEL
    blinds9 = [ints(c.blinds_w[i])[: HIDE_W[i]] for i in range(5)] + [ints(c.blinds_wsel[i])[:HIDE_WSEL] for i in range(3)] + [[]]
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
    vecs = np.concatenate([cos, tables])
    if not shuffle:
        for slot in list(range(5, 8)) + list(range(31, 56)):
            vecs[slot] = 0
    tq = oc.t_quotient(n, 6, vecs, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv)
    out["t_quotient"] = tq
    t = oc.mul_var(oc.ntt(tq, inverse=True), c.k1_inv)
    out["t"] = t
    t_int = ints(t)[: c.t_len]
    # split_t_and_commit (helpers.rs:1335-1394) with the reference's argument n + 2
    chunk_size, prev, rands = n + 2, 0, ints(c.t_rands)
    cm_t, t_blinds, chunks = [], [], []
    for i in range(5):
        start = i * chunk_size
        end = c.t_len if i == 4 else (i + 1) * chunk_size
        coefs = t_int[start:min(c.t_len, end)] if start < c.t_len else []
        if i != 4:
            coefs = coefs + [0] * (chunk_size + 1 - len(coefs))
            coefs[chunk_size] = (coefs[chunk_size] + rands[i]) % opy.R
            coefs[0] = (coefs[0] - prev) % opy.R
        elif not coefs:
            coefs = [(-prev) % opy.R]
        else:
            coefs[0] = (coefs[0] - prev) % opy.R
        prev = rands[i]
        cm, blinds = fold_and_commit(c, mono_pts, coefs, len(coefs))            # degree = coefs.len() (helpers.rs:1367)
        cm_t.append(cm)
        t_blinds.append(oc.fr_from_ints(blinds + [0] * (3 - len(blinds))))
        chunks.append(coefs)
    out["cm_t"] = aff_wire(cm_t)
    out["t_blinds"] = np.stack(t_blinds)
    out["chunks"] = np.stack([pad(oc.fr_from_ints(ch), n + 8) for ch in chunks])

    def poly_of(kind, idx):
        return {"c": lambda: polys[idx], "t": lambda: tpolys[idx], "k": lambda: chunks[idx], "r": lambda: r}[kind]()
    # ---- round 4
    points = (c.zeta, c.zeta_omega)
    out["evals"] = np.stack([oc.poly_eval(oc.fr_from_ints(poly_of(kind, idx)), points[pt]) for kind, idx, pt in eval_plan(shuffle)])
    # ---- round 5
    r = [0] * (n + 3)
    for s_k, (kind, idx) in zip(ints(c.r_scalars), r_plan(shuffle)):
        for j, v in enumerate(poly_of(kind, idx)[: n + 3]):
            r[j] = (r[j] + s_k * v) % opy.R
    out["r"] = oc.fr_from_ints(r)
    cm_q, q_blinds, quotients = [], [], []
    for plan, point, alpha in zip(open_plan(shuffle), points, (c.alpha_open, c.alpha_open2)):
        stack = np.stack([pad(oc.fr_from_ints(poly_of(kind, idx)), n + 3) for kind, idx in plan])
        q, _, rem_zero = oc.open_quotient(stack, point, alpha)
        assert rem_zero
        qi = ints(q)
        assert qi[n + 2] == 0 and qi[n + 1] != 0                                 # degree n + 1
        cm, blinds = fold_and_commit(c, mono_pts, qi[: n + 2], n + 1)            # degree = q.degree() (pcs.rs:138)
        cm_q.append(cm)
        q_blinds.append(oc.fr_from_ints(blinds + [0] * (3 - len(blinds))))
        quotients.append(pad(q[: n + 2], n + 8))
    out["cm_q"] = aff_wire(cm_q)
    out["q_blinds"] = np.stack(q_blinds)
    out["quotients"] = np.stack(quotients)
    return out
