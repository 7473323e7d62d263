#!/usr/bin/env python3
"""One proof's worth of hot-path work, chained on the device the way a GPU-resident `prover_with_lagrange`
(uzkge/src/plonk/prover.rs:88-394) issues it -- the stand-in for BASELINE config #4 (the Rust prover itself cannot run
here: no toolchain).  The CALL MIX is the reference's: which polynomials are transformed, committed, evaluated, combined
and opened, in which batches and at which lengths.  The circuit and the witness are synthetic (random field elements of
the real shapes: n constraints, quotient domain 6n, 5 wires, 3 wire selectors, 46 per-circuit polynomials); the SRS files
are the reference's own.  Fiat-Shamir challenges, the prover's random blinds and r_poly's O(1) scalars are given
(seeded), where the Rust draws / derives them; only those scalars, the commitments and the evaluations cross PCIe.

  setup     per-circuit polynomials -> their coset evaluations over the 6n domain (the indexer's loop, indexer.rs:316-470)
  round 1   iFFT(n) x9 (5 wires, 3 wire selectors, pi) into 6n-slots, hide, 8 commits with blinds      prover.rs:151-192
  round 2   z_poly grand product, iFFT(n), hide, commit                                         prover.rs:199-209, helpers.rs:160-220
  round 3   coset FFT(6n) x10, quotient kernel, coset iFFT(6n)                                  helpers.rs:223-678
            split_t_and_commit with chunk = n + 2: split, fold, FFT(n), 5 commits with blinds   helpers.rs:1323-1408
  round 4   15 evaluations at zeta, 4 at zeta * omega                                           prover.rs:246-273
  round 5   r_poly: 43 polynomials x scalars; batch_prove of 16 polynomials at zeta and 4 at zeta * omega:
            quotient, fold, FFT(n), commit with blinds                                          helpers.rs:681-1090, pcs.rs:107-168

Every commit is `lagrange_pcs.commit(evals)` + `apply_blind_factors` (prover.rs:132-142) as ONE batched MSM: the registered
bases are lagrange[0..n) || srs[0..3) || srs[n..n+3), the scalars evals || b || -b (uzk_msm_g1_batch_tail_device).

Everything goes through the C ABI alone (uzk_dev_alloc / uzk_dev_copy for the buffers): no torch, no HIP binding --
tests/cpp/prover_rounds.cpp and rust/uzkge-glue/gpu_prover.rs issue the same calls in the same order.
tests/test_gpu_prover_chain.py checks every commitment, evaluation and intermediate polynomial against the CPU oracle chain.
As a script: timing of the whole chain (python tools/prover_chain.py [--reps 5])."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from uzkge_amd import backend as b
from uzkge_amd import poly_commit as pc

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
N_WIRES, N_WSEL, N_TABLES = 5, 3, 46
# hiding degrees: TurboCS::get_hiding_degree (constraint_system/turbo/mod.rs:366-373) gives wires 0..2 three blinds and wires
# 3, 4 two; wire selectors 2 (prover.rs:186), z 3 (prover.rs:204).  The batched hide takes one degree per call, so every
# polynomial carries three blind slots and the unused third one is zero -- adding a zero blind is the identity.
HIDE_W, HIDE_WSEL, HIDE_Z = (3, 3, 3, 2, 2), 2, 3
# slots of the 46 per-circuit polynomials (= UZK_TQ_Q .. UZK_TQ_QECC minus UZK_TQ_Q): q (9), s (5), l1, qb, q_prk (4),
# coset_quotient, q_shuffle_public_key (12), q_shuffle_generator (12), q_ecc
T_Q, T_S, T_L1, T_QB, T_QPRK, T_CQ, T_QPK, T_QG, T_QECC = 0, 9, 14, 15, 16, 20, 21, 33, 45


def eval_plan(shuffle: bool):
    """Round 4 (prover.rs:246-273): (kind, index, point) with point 0 = zeta, 1 = zeta * omega, in the reference's order of
    computation.  kind: 'c' = a polynomial of this proof (slot of d_coefs: w0..4, wsel0..2, pi, z), 't' = a circuit polynomial."""
    plan = [("c", i, 0) for i in range(5)] + [("t", T_S + i, 0) for i in range(4)] + [("t", T_QPRK + 2, 0), ("t", T_QPRK + 3, 0)]
    plan += [("c", 9, 1)] + [("c", i, 1) for i in range(3)]
    if shuffle:
        plan += [("t", T_QECC, 0)] + [("c", 5 + i, 0) for i in range(3)]
    return plan


def r_plan(shuffle: bool):
    """r_poly's polynomials (helpers.rs:681-999) in the order the scalars multiply them: q (9), z, the last s, qb, q_prk1,
    q_prk2, [q_pk (12), q_g (12)], the t chunks (5).  ('t', slot) | ('c', slot) | ('k', chunk)."""
    plan = [("t", T_Q + i) for i in range(9)] + [("c", 9), ("t", T_S + 4), ("t", T_QB), ("t", T_QPRK), ("t", T_QPRK + 1)]
    if shuffle:
        plan += [("t", T_QPK + i) for i in range(12)] + [("t", T_QG + i) for i in range(12)]
    return plan + [("k", i) for i in range(5)]


def open_plan(shuffle: bool):
    """polys_to_open at zeta (prover.rs:329-347): w (5), s (4), q_prk3, q_prk4, [q_ecc, w_sel (3)], r; at zeta * omega: z, w0..2."""
    at_zeta = [("c", i) for i in range(5)] + [("t", T_S + i) for i in range(4)] + [("t", T_QPRK + 2), ("t", T_QPRK + 3)]
    if shuffle:
        at_zeta += [("t", T_QECC)] + [("c", 5 + i) for i in range(3)]
    at_zeta += [("r", 0)]
    return at_zeta, [("c", 9), ("c", 0), ("c", 1), ("c", 2)]


class ChainInputs:
    """Everything a chain run consumes, as host arrays (no GPU needed): the reference's SRS files, the synthetic circuit
    and witness, the seeded challenges and blinds.  tests/chain_oracle.py computes the expected outputs from the same object."""

    def __init__(self, n: int = 1 << 14, seed: int = 2024):
        self.n, self.m, self.seed = n, 6 * n, seed
        m = self.m
        rng = np.random.default_rng(seed)

        def fr(*shape):
            a = rng.integers(0, 1 << 63, size=shape + (4,), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape + (4,), dtype=np.uint64)
            a[..., 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
            return a
        # ---- parameters (reference files) and the combined commit bases
        self.lagrange_wire = pc.parse_srs_g1_wire(open(os.path.join(GOLDEN, f"lagrange-srs-{n}.bin"), "rb").read())
        self.mono_wire = pc.srs_params_wire(open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read(), n)
        self.bases = np.concatenate([self.lagrange_wire, self.mono_wire[:3], self.mono_wire[n:n + 3]])      # n + 6 points
        # ---- synthetic circuit: witness evaluations, wire selectors, public input, permutation, per-circuit polynomials
        self.w_evals = fr(N_WIRES, n)
        self.wsel_evals = fr(N_WSEL, n)
        self.pi_evals = np.zeros((n, 4), dtype=np.uint64); self.pi_evals[:8] = fr(8)
        self.perm = rng.permutation(N_WIRES * n).astype(np.uint32).reshape(N_WIRES, n)
        self.k = fr(N_WIRES)
        self.group_gen = b.domain_group_gen(n)
        self.table_polys = fr(N_TABLES, n)     # coefficient form (prover_params.{q,s,..}_polys); the chain derives the coset tables
        # challenges / blinds (seeded stand-ins for the transcript and the prover's rng)
        sc = fr(16)
        self.beta, self.gamma, self.alpha, self.zeta, self.alpha_open = sc[0], sc[1], sc[2], sc[3], sc[4]
        self.anemoi_g, self.edwards_a, self.alpha_open2 = sc[5], sc[6], sc[7]
        self.blinds_w = fr(N_WIRES, 3); self.blinds_wsel = fr(N_WSEL, 3); self.blinds_z = fr(HIDE_Z)
        for i, hd in enumerate(HIDE_W):
            self.blinds_w[i, hd:] = 0
        self.blinds_wsel[:, HIDE_WSEL:] = 0
        self.t_rands = fr(5)
        self.r_scalars = fr(43)
        # 1 / Z_H on the coset: 1 / (k1^n * g_m^(n i) - 1), i < 6 (helpers.rs:242-252) -- O(1) host arithmetic
        k1 = pc.fr_to_int(self.k[1]); gm = pc.fr_to_int(b.domain_group_gen(m)); R = pc.FR_MODULUS
        self.z_h_inv = np.stack([pc.fr_from_int(pow((pow(k1, n, R) * pow(gm, n * i, R) - 1) % R, -1, R)) for i in range(6)])
        self.k1_inv = pc.fr_from_int(pow(k1, -1, R))
        self.anemoi_g_inv = pc.fr_from_int(pow(pc.fr_to_int(self.anemoi_g), -1, R))
        self.zeta_omega = pc.fr_from_int(pc.fr_to_int(self.zeta) * pc.fr_to_int(self.group_gen) % R)
        # deg t = deg z + sum deg w_j - n = (n + 2) + 3 (n + 2) + 2 (n + 1) - n = 5n + 10 (the permutation term dominates)
        self.t_len = 5 * n + 11


class _Buf:
    """`count` field elements of device memory from uzk_dev_alloc."""

    def __init__(self, count: int, zero: bool = False):
        self.count = count
        self.ptr = b.dev_alloc(count * 32)
        if zero:
            b.dev_memset(self.ptr, 0, count * 32)

    def at(self, elem: int) -> int:
        return self.ptr + 32 * elem

    def host(self, count: int = None, offset: int = 0) -> np.ndarray:
        return b.dev_download(self.at(offset), (self.count - offset if count is None else count, 4))

    def free(self):
        if self.ptr:
            b.dev_free(self.ptr)
            self.ptr = 0


class ProverChain:
    def __init__(self, n: int = 1 << 14, seed: int = 2024, shuffle: bool = True, precompute: bool = True, inputs: ChainInputs = None,
                 keep_blinds: bool = False):
        b.init(0)
        self.keep_blinds = keep_blinds       # tests: also fetch the fold blinds (a synchronisation per fold; timing runs leave it off)
        inp = inputs if inputs is not None else ChainInputs(n, seed)
        self.__dict__.update(inp.__dict__)               # the inputs' fields are read as attributes of the chain
        self.inputs, self.shuffle = inp, shuffle
        n, m = self.n, self.m
        self.srs = b.Srs.from_host(self.bases)
        b.tune("msm_no_precompute", 0)
        if precompute:
            self.srs.precompute(0)            # static SRS: window table (same commitments, shorter calls)
        self.cs = n + 8                       # stride of the chunk / quotient arrays
        # ---- device residency
        self.d_evals = _Buf(9 * n)            # w0..w4, wsel0..2, pi
        b.dev_upload(self.d_evals.ptr, np.concatenate([self.w_evals.reshape(-1, 4), self.wsel_evals.reshape(-1, 4), self.pi_evals]))
        self.d_perm = b.dev_alloc(N_WIRES * n * 4)
        b.dev_upload(self.d_perm, self.perm)
        self.d_coefs = _Buf(10 * m, zero=True)           # 10 polynomials of this proof, 6n slots each (zero beyond n + 3)
        self.d_coset = _Buf(10 * m)
        self.d_tq, self.d_t, self.d_z = _Buf(m), _Buf(m), _Buf(n)
        self.d_chunks = _Buf(5 * self.cs)
        self.d_fold = _Buf(5 * n)
        self.d_tail = _Buf(5 * 6)
        self.d_q = _Buf(2 * self.cs)
        self.d_r = _Buf(self.cs)
        self.h_lens = b.host_alloc(4 * 8)     # pinned result words of the asynchronous trimmed-length checks: t, q at zeta, q at zeta omega
        # the circuit's polynomials and their coset evaluations (the indexer's work, once per circuit): zero-padded copy into
        # 6n-slots, one batched coset FFT in place
        self.d_tpolys = _Buf(N_TABLES * n)
        b.dev_upload(self.d_tpolys.ptr, self.table_polys.reshape(-1, 4))
        self.d_tables = _Buf(N_TABLES * m, zero=True)
        b.dev_copy2d(self.d_tables.ptr, m * 32, self.d_tpolys.ptr, n * 32, n * 32, N_TABLES)
        b.ntt_batch_device(self.d_tables.ptr, self.d_tables.ptr, m, N_TABLES, coset_shift=self.k[1], sync=True)
        # the powers group[i] = omega^i on the device: forward NTT of X (coefficient 1 at index 1)
        x = np.zeros((n, 4), dtype=np.uint64)
        x[1] = pc.fr_from_int(1)
        self.d_group = _Buf(n)
        b.dev_upload(self.d_group.ptr, x)
        b.ntt_device(self.d_group.ptr, self.d_group.ptr, n, sync=True)
        self.out = {}

    # the device address and length of a polynomial named by a plan entry
    def _poly(self, kind, idx):
        n, m = self.n, self.m
        if kind == "c":
            return self.d_coefs.at(idx * m), n + 3
        if kind == "t":
            return self.d_tpolys.at(idx * n), n
        if kind == "k":
            return self.d_chunks.at(idx * self.cs), int(self.chunk_lens[idx])
        return self.d_r.ptr, n + 3

    @staticmethod
    def _tails(blinds_list):
        tail = np.zeros((len(blinds_list), 6, 4), dtype=np.uint64)
        for i, bl in enumerate(blinds_list):
            bl = np.asarray(bl, dtype=np.uint64).reshape(-1, 4)
            tail[i, : bl.shape[0]] = bl
            tail[i, 3:3 + bl.shape[0]] = pc.fr_neg(bl)
        return tail

    def run(self):
        n, m, o, cs = self.n, self.m, self.out, self.cs
        coefs = self.d_coefs
        # ---- round 1: iFFT of the nine evaluation vectors straight into their 6n-slots, hide, commit wires and wire selectors
        b.ntt_batch_strided_device(self.d_evals.ptr, n, coefs.ptr, m, n, 9, inverse=True)
        b.hide_polynomial_batch_device(coefs.ptr, m, n, np.concatenate([self.blinds_w, self.blinds_wsel]), n)
        o["cm_w_wsel"] = b.msm_batch_tail_device(self.srs, self.d_evals.ptr, n, n, 8, self._tails(list(self.blinds_w) + list(self.blinds_wsel)), 6)
        # Fiat-Shamir: the challenges are the caller's (the Rust prover draws them from its transcript after each round,
        # prover.rs:194-244,300-302).  Seeded stand-ins for the timing runs; `self.fs` (tests/test_gpu_plonk_verifier.py) derives them
        # from the commitments and evaluations the way the reference's transcript does.
        fs = getattr(self, "fs", None)
        if fs is not None:
            self.beta, self.gamma = fs.beta_gamma(o["cm_w_wsel"])
        # ---- round 2: permutation grand product
        b.z_poly_device(self.d_evals.ptr, self.d_perm, self.d_group.ptr, self.k, self.beta, self.gamma, n, N_WIRES, self.d_z.ptr)
        b.ntt_batch_strided_device(self.d_z.ptr, n, coefs.at(9 * m), m, n, 1, inverse=True)
        b.hide_polynomial_batch_device(coefs.at(9 * m), m, n, self.blinds_z.reshape(1, 3, 4), n)
        o["cm_z"] = b.msm_batch_tail_device(self.srs, self.d_z.ptr, n, n, 1, self._tails([self.blinds_z]), 6)
        if fs is not None:
            self.alpha = fs.alpha(o["cm_z"])
        # ---- round 3: quotient polynomial
        b.ntt_batch_device(coefs.ptr, self.d_coset.ptr, m, 10, coset_shift=self.k[1])
        cos = [self.d_coset.at(i * m) for i in range(10)]
        tab = [self.d_tables.at(i * m) for i in range(N_TABLES)]
        ptrs = cos[:5] + [cos[5 + i] if self.shuffle else 0 for i in range(3)] + [cos[8], cos[9]]
        ptrs += tab[:21]                                                     # q (9), s (5), l1, qb, q_prk (4), coset_quotient
        ptrs += [tab[21 + i] if self.shuffle else 0 for i in range(25)]      # q_pk (12), q_g (12), q_ecc
        self.tq_ptrs = ptrs
        b.t_quotient_device(n, 6, ptrs, self.alpha, self.beta, self.gamma, self.k, self.anemoi_g, self.anemoi_g_inv, self.edwards_a,
                            self.z_h_inv, self.d_tq.ptr, sync=False)
        b.ntt_device(self.d_tq.ptr, self.d_t.ptr, m, inverse=True, coset_shift=self.k1_inv)
        # split_t_and_commit (helpers.rs:1323-1408, chunk = n + 2): split with the random blinds, fold mod X^n - 1, FFT(n), commit
        # from_coefs trims t (field_polynomial.rs:86-90) and its coefs.len() drives the split (helpers.rs:1333): go on with the length a
        # well-formed proof has (deg t = 5n + 10) while the device measures the trimmed one into pinned memory; compare after the
        # commit has synchronised, redo the split with the measured length if they ever differ
        lens_view = np.ctypeslib.as_array(ctypes.cast(self.h_lens, ctypes.POINTER(ctypes.c_uint64)), shape=(4,))
        b.poly_trimmed_len_async_device(self.d_t.ptr, m, [self.t_len], self.h_lens)

        def split_and_commit(t_len):
            self.chunk_lens = b.split_t_device(self.d_t.ptr, t_len, n + 2, self.t_rands, self.d_chunks.ptr, cs)
            assert [pc.max_power_of_2(int(v)) for v in self.chunk_lens] == [n] * 5      # degree = coefs.len() (helpers.rs:1367)
            o["t_blinds"] = b.fold_blinds_batch_device(self.d_chunks.ptr, cs, self.chunk_lens, n, self.d_fold.ptr, n, self.d_tail.ptr, 6,
                                                       want_blinds=self.keep_blinds)
            b.ntt_batch_device(self.d_fold.ptr, self.d_fold.ptr, n, 5)
            o["cm_t"] = b.msm_batch_tail_device(self.srs, self.d_fold.ptr, n, n, 5, self.d_tail.ptr, 6)
        split_and_commit(self.t_len)
        if int(lens_view[0]) != self.t_len:
            split_and_commit(int(lens_view[0]))
        if fs is not None:
            self.zeta = fs.zeta(o["cm_t"])
            self.zeta_omega = pc.fr_from_int(pc.fr_to_int(self.zeta) * pc.fr_to_int(self.group_gen) % pc.FR_MODULUS)
        # ---- round 4: the evaluations of prover.rs:246-273 in one launch
        plan = eval_plan(self.shuffle)
        pl = [self._poly(kind, idx) for kind, idx, _ in plan]
        o["evals"] = b.poly_eval_ptrs_device([p for p, _ in pl], [ln for _, ln in pl], [pt for _, _, pt in plan], np.stack([self.zeta, self.zeta_omega]))
        # ---- round 5: r(X) = sum of scalars * polynomials (r_poly's shape), then the two openings
        rp = [self._poly(kind, idx) for kind, idx in r_plan(self.shuffle)]
        # r_poly's scalars are O(1) formulas of the evaluations and the challenges (helpers.rs:681-1002) and stay with the caller, as in
        # the reference: the timing runs pass seeded stand-ins, tests/test_gpu_plonk_verifier.py derives them from round 4's evaluations
        if fs is not None:
            self.alpha_open, self.alpha_open2 = fs.after_evaluations(o["evals"], self.zeta, self.zeta_omega)
        r_scalars = self.r_scalar_hook(o["evals"]) if getattr(self, "r_scalar_hook", None) is not None else self.r_scalars
        b.poly_lincomb_device([p for p, _ in rp], [ln for _, ln in rp], r_scalars[: len(rp)], self.d_r.ptr, n + 3)
        at_zeta, at_zeta_omega = open_plan(self.shuffle)
        for j, (plan_j, point, alpha) in enumerate(((at_zeta, self.zeta, self.alpha_open), (at_zeta_omega, self.zeta_omega, self.alpha_open2))):
            op = [self._poly(kind, idx) for kind, idx in plan_j]
            b.open_quotient_ptrs_device([p for p, _ in op], [ln for _, ln in op], point, alpha, self.d_q.at(j * cs), cs)
        # degree = q.degree() (pcs.rs:138): n + 1 for n + 3 coefficients divided by X - z, so max_power_of_2 = n and two blinds;
        # expected lengths first, the device's measurement checked after the commit
        b.poly_trimmed_len_async_device(self.d_q.ptr, cs, [n + 3, n + 3], self.h_lens + 8)

        def fold_and_commit(q_lens):
            assert [pc.max_power_of_2(int(v) - 1) for v in q_lens] == [n, n]
            o["q_blinds"] = b.fold_blinds_batch_device(self.d_q.ptr, cs, q_lens, n, self.d_fold.ptr, n, self.d_tail.ptr, 6,
                                                       want_blinds=self.keep_blinds)
            b.ntt_batch_device(self.d_fold.ptr, self.d_fold.ptr, n, 2)
            o["cm_q"] = b.msm_batch_tail_device(self.srs, self.d_fold.ptr, n, n, 2, self.d_tail.ptr, 6)
        fold_and_commit([n + 2, n + 2])
        if [int(lens_view[1]), int(lens_view[2])] != [n + 2, n + 2]:
            fold_and_commit([int(lens_view[1]), int(lens_view[2])])
        return o

    def snapshot(self):
        """The device-resident intermediates of the last run as host arrays, named as tests/chain_oracle.py names them."""
        n, m, cs = self.n, self.m, self.cs
        b.sync()
        coefs = self.d_coefs.host().reshape(10, m, 4)
        return {"coefs": coefs[:, : n + 3], "coefs_beyond": coefs[:, n + 3:], "coset_evals": self.d_coset.host().reshape(10, m, 4),
                "t_quotient": self.d_tq.host(), "t": self.d_t.host(), "z_evals": self.d_z.host(), "r": self.d_r.host(n + 3),
                "chunks": self.d_chunks.host().reshape(5, cs, 4), "quotients": self.d_q.host().reshape(2, cs, 4),
                "tables": self.d_tables.host().reshape(N_TABLES, m, 4)}

    def release(self):
        self.srs.release()
        for v in list(self.__dict__.values()):
            if isinstance(v, _Buf):
                v.free()
        if self.d_perm:
            b.dev_free(self.d_perm)
            self.d_perm = 0
        if self.h_lens:
            b.host_free(self.h_lens)
            self.h_lens = 0


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=14)
    a = ap.parse_args()
    res = {}
    for name, kw in (("window_table", dict(precompute=True)), ("general", dict(precompute=False)), ("no_shuffle_terms", dict(shuffle=False))):
        c = ProverChain(n=1 << a.log_n, **kw)
        c.run(); b.sync()
        t = time.perf_counter()
        for _ in range(a.reps):
            c.run()
        b.sync()
        res[name + "_ms_per_proof_chain"] = round((time.perf_counter() - t) / a.reps * 1e3, 3)
        b.profile_reset(); b.profile_enable(True); c.run(); b.sync(); b.profile_enable(False)
        tab = b.profile_table()
        res[name + "_kernel_ms"] = round(sum(ms for k, (cnt, ms) in tab.items() if not k.startswith("host_")), 3)
        res[name + "_kernels"] = {k: [cnt, round(ms, 3)] for k, (cnt, ms) in sorted(tab.items())}
        c.release()
    print(json.dumps(res))
