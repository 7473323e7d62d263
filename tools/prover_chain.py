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

Everything goes through the C ABI alone: uzk_circuit_create makes the circuit resident, uzk_prover_create the proof's buffers,
and uzk_prove_round1..5 run the five rounds -- the implementation tests/cpp/prover_rounds.cpp and rust/uzkge-glue/gpu_prover.rs
drive as well (no torch, no HIP binding).
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


class _View:
    """A prover buffer seen from Python: device address, element count, download."""

    def __init__(self, ptr: int, count: int):
        self.ptr, self.count = ptr, count

    def at(self, elem: int) -> int:
        return self.ptr + 32 * elem

    def host(self, count: int = None, offset: int = 0) -> np.ndarray:
        return b.dev_download(self.at(offset), (self.count - offset if count is None else count, 4))


class ProverChain:
    """One proof through uzk_prove_round1..5 (include/uzkge_gpu.h): the circuit is a uzk_circuit handle, the proof's buffers a
    uzk_prover; this class only supplies what the Rust prover supplies -- witness, blinds, challenges, r_poly's scalars."""

    def __init__(self, n: int = 1 << 14, seed: int = 2024, shuffle: bool = True, precompute: bool = True, inputs: ChainInputs = None,
                 keep_blinds: bool = False):
        b.init(0)
        self.keep_blinds = keep_blinds       # tests: also fetch the fold blinds (read back from the tail buffer after rounds 3 and 5)
        inp = inputs if inputs is not None else ChainInputs(n, seed)
        self.__dict__.update(inp.__dict__)               # the inputs' fields are read as attributes of the chain
        self.inputs, self.shuffle = inp, shuffle
        n = self.n
        b.tune("msm_no_precompute", 0)
        polys = [self.table_polys[i] for i in range(N_TABLES)]
        if getattr(inp, "satisfiable", False):
            polys[T_CQ] = None               # coset_quotient: the library builds it (the random circuits keep their arbitrary slot 20)
        # a random circuit is satisfied by nothing: its t fills all 6n coefficients, and round 3 would refuse it (as the reference
        # aborts on it).  The timing / parity chains mark it synthetic: t is taken as its first t_len coefficients
        # (tests/chain_oracle.py does the same); circuits made satisfiable (tests/plonk_verifier_oracle.py) run the real check.
        self.circuit = b.Circuit(n, self.lagrange_wire, self.bases[n:], self.perm, self.k, self.anemoi_g, self.anemoi_g_inv, self.edwards_a,
                                 polys, shuffle=shuffle, precompute=precompute, synthetic=not getattr(inp, "satisfiable", False))
        self.prover = b.Prover(n, 1, shared=False)       # its buffers are read back (snapshot, tq_ptrs)
        self.cs = n + 8
        self._srs = None
        self.witness = np.ascontiguousarray(self.w_evals.reshape(1, N_WIRES * n, 4))
        self.wsel = np.ascontiguousarray(self.wsel_evals.reshape(1, N_WSEL * n, 4))
        self.pi_index = np.arange(8, dtype=np.uint32)    # ChainInputs puts its eight public inputs on the first eight constraints
        self.hiding = list(HIDE_W) + [HIDE_WSEL] * N_WSEL
        self.d_coset = _View(*self.prover.buffer(b.PB_COSET))
        self.d_tq = _View(*self.prover.buffer(b.PB_TQ))
        self.out = {}

    @property
    def srs(self):
        """The Lagrange bases || blind bases as a plain SRS handle (tests commit the circuit's own polynomials with it)."""
        if self._srs is None:
            self._srs = b.Srs.from_host(self.bases)
        return self._srs

    @property
    def tq_ptrs(self):
        """The 56 vector addresses round 3 hands the quotient kernel (UZK_TQ_* order), rebuilt from the public accessors."""
        m = self.m
        cos = [self.d_coset.at(i * m) for i in range(10)]
        tab = [self.circuit.table(i, coset=True)[0] if i < self.circuit.n_slots else 0 for i in range(N_TABLES)]
        return cos[:5] + [cos[5 + i] if self.shuffle else 0 for i in range(3)] + [cos[8], cos[9]] + tab

    def _blinds_from_tail(self, count):
        tail = self.prover.download(b.PB_TAIL)[: count * 6].reshape(count, 6, 4)
        return np.ascontiguousarray(tail[:, :3])

    def run(self):
        o, pr = self.out, self.prover
        # ---- round 1.  Fiat-Shamir: the challenges are the caller's (the Rust prover draws them from its transcript after each
        # round, prover.rs:194-244,300-302).  Seeded stand-ins for the timing runs; `self.fs` (tests/test_gpu_plonk_verifier.py)
        # derives them from the commitments and evaluations the way the reference's transcript does.
        o["cm_w_wsel"] = pr.round1(self.circuit, self.witness, self.wsel, self.pi_index, self.pi_evals[:8].reshape(1, 8, 4), self.hiding,
                                   np.concatenate([self.blinds_w, self.blinds_wsel]))
        fs = getattr(self, "fs", None)
        if fs is not None:
            self.beta, self.gamma = fs.beta_gamma(o["cm_w_wsel"])
        # ---- round 2
        o["cm_z"] = pr.round2(self.beta, self.gamma, self.blinds_z)
        if fs is not None:
            self.alpha = fs.alpha(o["cm_z"])
        # ---- round 3
        o["cm_t"] = pr.round3(self.alpha, self.t_rands)
        if self.keep_blinds:
            o["t_blinds"] = self._blinds_from_tail(5)
        if fs is not None:
            self.zeta = fs.zeta(o["cm_t"])
            self.zeta_omega = pc.fr_from_int(pc.fr_to_int(self.zeta) * pc.fr_to_int(self.group_gen) % pc.FR_MODULUS)
        # ---- round 4
        o["evals"] = pr.round4(self.zeta, self.shuffle)
        # ---- round 5.  r_poly's scalars are O(1) formulas of the evaluations and the challenges (helpers.rs:681-1002) and stay
        # with the caller, as in the reference: the timing runs pass seeded stand-ins, tests/test_gpu_plonk_verifier.py derives
        # them from round 4's evaluations
        if fs is not None:
            self.alpha_open, self.alpha_open2 = fs.after_evaluations(o["evals"], self.zeta, self.zeta_omega)
        r_scalars = self.r_scalar_hook(o["evals"]) if getattr(self, "r_scalar_hook", None) is not None else self.r_scalars
        o["cm_q"] = pr.round5(r_scalars[: len(r_plan(self.shuffle))], self.alpha_open, self.alpha_open2)
        if self.keep_blinds:
            o["q_blinds"] = self._blinds_from_tail(2)
        return o

    def snapshot(self):
        """The device-resident intermediates of the last run as host arrays, named as tests/chain_oracle.py names them."""
        n, m, cs, pr = self.n, self.m, self.cs, self.prover
        b.sync()
        coefs = pr.download(b.PB_COEFS).reshape(10, m, 4)
        tables = np.stack([b.dev_download(self.circuit.table(i, coset=True)[0], (m, 4)) for i in range(self.circuit.n_slots)])
        return {"coefs": coefs[:, : n + 3], "coefs_beyond": coefs[:, n + 3:], "coset_evals": pr.download(b.PB_COSET).reshape(10, m, 4),
                "t_quotient": pr.download(b.PB_TQ), "t": pr.download(b.PB_T), "z_evals": pr.download(b.PB_EVALS).reshape(10, n, 4)[9],
                "r": pr.download(b.PB_R)[: n + 3], "chunks": pr.download(b.PB_CHUNKS).reshape(5, cs, 4),
                "quotients": pr.download(b.PB_Q).reshape(2, cs, 4), "tables": tables}

    def release(self):
        if self._srs is not None:
            self._srs.release()
            self._srs = None
        self.prover.destroy()
        self.circuit.release()


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
