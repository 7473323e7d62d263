#!/usr/bin/env python3
"""One proof's worth of hot-path work, chained on the device the way a GPU-resident `prover_with_lagrange`
(uzkge/src/plonk/prover.rs:88-394) would issue it -- a stand-in for BASELINE config #4 (the Rust prover itself cannot
run here: no toolchain).  Circuit tables and the witness are synthetic (random field elements of the real shapes:
n = 2^14 constraints, quotient domain 6n = 98304, 5 wires, 3 wire selectors, 46 per-circuit coset tables); the SRS files
are the reference's own.  Fiat-Shamir challenges and the prover's random blinds are given (seeded), where the Rust
draws them; only those scalars, the commitments and the evaluations cross PCIe.

  round 1   iFFT(n) x9 (pi, 5 wires, 3 wire selectors), hide, 8 commits          prover.rs:151-192
  round 2   z_poly grand product, iFFT(n), hide, commit                           prover.rs:199-209, helpers.rs:160-220
  round 3   coset FFT(6n) x10, quotient kernel, coset iFFT(6n)                    helpers.rs:223-678
            split t into 5 chunks: fold, FFT(n), commit, blinds                   helpers.rs:1323-1408
  round 4   evaluations at zeta (and z at zeta * omega)                           prover.rs:246-273
  round 5   r_poly-shaped linear combination, two batch_prove openings            helpers.rs:1030, pcs.rs:107-168

Every commit is `MSM(lagrange SRS, evaluations) + blind factors` (prover.rs:132-149); the blind factors ride
in the same MSM: the registered bases are lagrange[0..n) || srs[0..3) || srs[n..n+3), the scalars evals || b || -b.

tests/test_gpu_prover_chain.py runs this and checks every commitment, evaluation vector and intermediate polynomial
against the CPU oracle chain.  As a script: timing of the whole chain (python tools/prover_chain.py [--reps 5])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
try:
    import torch
except ImportError:      # ChainInputs alone needs no torch
    torch = None

from uzkge_amd import backend as b
from uzkge_amd import poly_commit as pc

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
N_WIRES, N_WSEL, N_TABLES = 5, 3, 46
HIDE = {"w": 2, "wsel": 2, "z": 3}            # hiding degrees (prover.rs:166,186,204)


def _dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def _host(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


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
        # ---- synthetic circuit: witness evaluations, wire selectors, public input, permutation, tables
        self.w_evals = fr(N_WIRES, n)
        self.wsel_evals = fr(N_WSEL, n)
        self.pi_evals = np.zeros((n, 4), dtype=np.uint64); self.pi_evals[:8] = fr(8)
        self.perm = rng.permutation(N_WIRES * n).astype(np.uint32).reshape(N_WIRES, n)
        self.k = fr(N_WIRES)
        self.group_gen = b.domain_group_gen(n)
        self.tables = fr(N_TABLES, m)          # q (9), s (5), l1, qb, q_prk (4), coset_quotient, q_pk (12), q_g (12), q_ecc
        # challenges / blinds (seeded stand-ins for the transcript and the prover's rng)
        sc = fr(16)
        self.beta, self.gamma, self.alpha, self.zeta, self.alpha_open = sc[0], sc[1], sc[2], sc[3], sc[4]
        self.anemoi_g, self.edwards_a = sc[5], sc[6]
        self.blinds_w = fr(N_WIRES, HIDE["w"]); self.blinds_wsel = fr(N_WSEL, HIDE["wsel"]); self.blinds_z = fr(HIDE["z"])
        self.t_rands = fr(5)
        self.r_scalars = fr(12)
        # 1 / Z_H on the coset: 1 / (k1^n * g_m^(n i) - 1), i < 6 (helpers.rs:242-252) -- O(1) host arithmetic
        k1 = pc.fr_to_int(self.k[1]); gm = pc.fr_to_int(b.domain_group_gen(m)); R = pc.FR_MODULUS
        self.z_h_inv = np.stack([pc.fr_from_int(pow((pow(k1, n, R) * pow(gm, n * i, R) - 1) % R, -1, R)) for i in range(6)])
        self.k1_inv = pc.fr_from_int(pow(k1, -1, R))
        self.anemoi_g_inv = pc.fr_from_int(pow(pc.fr_to_int(self.anemoi_g), -1, R))
        self.zeta_omega = pc.fr_from_int(pc.fr_to_int(self.zeta) * pc.fr_to_int(self.group_gen) % R)


class ProverChain:
    def __init__(self, n: int = 1 << 14, seed: int = 2024, shuffle: bool = True, precompute: bool = True, inputs: ChainInputs = None):
        b.init(0)
        inp = inputs if inputs is not None else ChainInputs(n, seed)
        self.__dict__.update(inp.__dict__)               # the inputs' fields are read as attributes of the chain
        self.inputs, self.shuffle = inp, shuffle
        n, m = self.n, self.m
        self.srs = b.Srs.from_host(self.bases)
        b.tune("msm_no_precompute", 0)
        if precompute:
            self.srs.precompute(0)            # static SRS: window table (same commitments, shorter calls)
        # ---- device residency
        self.d_evals = _dev(np.concatenate([self.w_evals.reshape(-1, 4), self.wsel_evals.reshape(-1, 4), self.pi_evals]))   # [9n]
        self.d_perm = torch.from_numpy(self.perm.view(np.int32)).cuda()
        self.d_tables = _dev(self.tables.reshape(-1, 4))
        self.d_coefs = torch.zeros((10 * m, 4), dtype=torch.int64, device="cuda")      # 10 polynomials, 6n slots each
        self.d_tmp = torch.empty((10 * n, 4), dtype=torch.int64, device="cuda")
        self.d_coset = torch.empty((10 * m, 4), dtype=torch.int64, device="cuda")
        self.d_tq = torch.empty((m, 4), dtype=torch.int64, device="cuda")
        self.d_t = torch.empty((m, 4), dtype=torch.int64, device="cuda")
        self.d_z = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        self.d_sc = torch.zeros((8 * (n + 6), 4), dtype=torch.int64, device="cuda")    # commit scalars: evals || b || -b
        self.d_chunks = torch.zeros((5 * (n + 8), 4), dtype=torch.int64, device="cuda")
        self.d_fold = torch.empty((5 * n, 4), dtype=torch.int64, device="cuda")
        self.d_q = torch.empty((2 * (n + 8), 4), dtype=torch.int64, device="cuda")
        self.d_r = torch.empty((n + 8, 4), dtype=torch.int64, device="cuda")
        self.d_open = torch.zeros((16 * (n + 8), 4), dtype=torch.int64, device="cuda")
        # the powers group[i] = omega^i on the device: forward NTT of X (coefficient 1 at index 1)
        x = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        x[1] = _dev(pc.fr_from_int(1).reshape(1, 4))[0]
        self.d_group = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        b.ntt_device(x.data_ptr(), self.d_group.data_ptr(), n, sync=True)
        self.out = {}

    # commit `count` evaluation vectors (device, n each, stride n) with their blinds: one batched MSM over n + 6 bases
    def _commit(self, d_evals_ptr: int, count: int, blinds_list):
        n = self.n
        sc = self.d_sc[: count * (n + 6)].view(count, n + 6, 4)
        view = self._as_tensor(d_evals_ptr, count * n).view(count, n, 4)
        sc[:, :n] = view
        tail = np.zeros((count, 6, 4), dtype=np.uint64)
        for i, bl in enumerate(blinds_list):
            bl = np.asarray(bl, dtype=np.uint64).reshape(-1, 4)
            tail[i, : bl.shape[0]] = bl
            tail[i, 3:3 + bl.shape[0]] = pc.fr_neg(bl)
        sc[:, n:] = _dev(tail.reshape(-1, 4)).view(count, 6, 4)
        torch.cuda.synchronize()
        return b.msm_batch_device(self.srs, sc.data_ptr(), n + 6, count)

    def _as_tensor(self, ptr: int, rows: int) -> torch.Tensor:
        for t in (self.d_evals, self.d_z, self.d_fold, self.d_tmp):
            base = t.data_ptr()
            if base <= ptr < base + t.numel() * 8:
                off = (ptr - base) // 32
                return t[off:off + rows]
        raise ValueError("pointer outside the chain's buffers")

    def run(self):
        n, m, o = self.n, self.m, self.out
        coefs = self.d_coefs.view(10, m, 4)              # order = UZK_TQ slots: w0..w4, wsel0..2, pi, z
        # ---- round 1: iFFT of the nine evaluation vectors, hide, commit wires and wire selectors
        b.ntt_batch_device(self.d_evals.data_ptr(), self.d_tmp.data_ptr(), n, 9, inverse=True, sync=True)
        coefs[:, n:n + 8] = 0                            # the slots the blinds are added into (a repeated run starts clean)
        coefs[:9, :n] = self.d_tmp[: 9 * n].view(9, n, 4)
        torch.cuda.synchronize()
        for i in range(N_WIRES):
            b.hide_polynomial_device(coefs[i].data_ptr(), m, self.blinds_w[i], n)
        for i in range(N_WSEL):
            b.hide_polynomial_device(coefs[5 + i].data_ptr(), m, self.blinds_wsel[i], n)
        o["cm_w_wsel"] = self._commit(self.d_evals.data_ptr(), 8, list(self.blinds_w) + list(self.blinds_wsel))
        # ---- round 2: permutation grand product
        b.z_poly_device(self.d_evals.data_ptr(), self.d_perm.data_ptr(), self.d_group.data_ptr(), self.k, self.beta, self.gamma, n, N_WIRES,
                        self.d_z.data_ptr())
        b.ntt_device(self.d_z.data_ptr(), self.d_tmp.data_ptr(), n, inverse=True, sync=True)
        coefs[9, :n] = self.d_tmp[:n]
        torch.cuda.synchronize()
        b.hide_polynomial_device(coefs[9].data_ptr(), m, self.blinds_z, n)
        o["cm_z"] = self._commit(self.d_z.data_ptr(), 1, [self.blinds_z])
        # ---- round 3: quotient polynomial
        b.ntt_batch_device(self.d_coefs.data_ptr(), self.d_coset.data_ptr(), m, 10, coset_shift=self.k[1])
        cos = self.d_coset.view(10, m, 4)
        tab = self.d_tables.view(N_TABLES, m, 4)
        ptrs = [cos[i].data_ptr() for i in range(5)]
        ptrs += [cos[5 + i].data_ptr() if self.shuffle else 0 for i in range(3)]
        ptrs += [cos[8].data_ptr(), cos[9].data_ptr()]
        ptrs += [tab[i].data_ptr() for i in range(21)]                       # q (9), s (5), l1, qb, q_prk (4), coset_quotient
        ptrs += [tab[21 + i].data_ptr() if self.shuffle else 0 for i in range(25)]   # q_pk (12), q_g (12), q_ecc
        self.tq_ptrs = ptrs
        b.t_quotient_device(n, 6, ptrs, self.alpha, self.beta, self.gamma, self.k, self.anemoi_g, self.anemoi_g_inv, self.edwards_a,
                            self.z_h_inv, self.d_tq.data_ptr(), sync=False)
        b.ntt_device(self.d_tq.data_ptr(), self.d_t.data_ptr(), m, inverse=True, coset_shift=self.k1_inv, sync=True)
        # split t (taken as 5n + 2 coefficients) into five chunks with the random blinds of helpers.rs:1353-1363
        ch = self.d_chunks.view(5, n + 8, 4)
        ch.zero_()
        for i in range(5):
            ln = n if i < 4 else n + 2
            ch[i, :ln] = self.d_t[i * n:i * n + ln]
        torch.cuda.synchronize()
        heads = _host(torch.stack([self.d_t[i * n] for i in range(5)]))
        prev = np.zeros((1, 4), dtype=np.uint64)
        fix = np.zeros((5, 2, 4), dtype=np.uint64)
        for i in range(5):
            fix[i, 0] = pc.fr_add_rows(heads[i:i + 1], pc.fr_neg(prev))[0]          # coefs[0] -= prev
            if i < 4:
                fix[i, 1] = self.t_rands[i]                                          # coefs[n] (zero so far) += rand_i
            prev = self.t_rands[i:i + 1]
        dfix = _dev(fix.reshape(-1, 4)).view(5, 2, 4)
        for i in range(5):
            ch[i, 0] = dfix[i, 0]
            if i < 4:
                ch[i, n] = dfix[i, 1]
        torch.cuda.synchronize()
        t_blinds = []
        for i in range(5):
            ln = n + 1 if i < 4 else n + 2
            t_blinds.append(b.fold_blinds_device(ch[i].data_ptr(), ln, n, self.d_fold.data_ptr() + i * n * 32))
        b.ntt_batch_device(self.d_fold.data_ptr(), self.d_fold.data_ptr(), n, 5, sync=True)
        o["cm_t"] = self._commit(self.d_fold.data_ptr(), 5, t_blinds)
        # ---- round 4: evaluations at zeta (all ten polynomials) and z at zeta * omega
        o["evals_zeta"] = b.poly_eval_batch_device(self.d_coefs.data_ptr(), m, 10, self.zeta)
        o["z_eval_zeta_omega"] = b.poly_eval_batch_device(coefs[9].data_ptr(), m, 1, self.zeta_omega)
        # ---- round 5: r(X) = sum of scalars * polynomials (r_poly's shape), then the two openings
        polys = [coefs[9].data_ptr()] + [ch[i].data_ptr() for i in range(5)] + [coefs[i].data_ptr() for i in range(6)]
        lens = [n + 3] + [n + 2] * 5 + [n + 3] * 6
        b.poly_lincomb_device(polys, lens, self.r_scalars, self.d_r.data_ptr(), n + 3)
        b.sync()                                         # asynchronous call; torch copies below run on torch's stream
        op = self.d_open.view(16, n + 8, 4)
        op.zero_()
        for j in range(10):
            op[j, :n + 3] = coefs[j, :n + 3]
        for j in range(5):
            op[10 + j] = ch[j]
        op[15, :n + 3] = self.d_r[:n + 3]
        torch.cuda.synchronize()
        q = self.d_q.view(2, n + 8, 4)
        o["open_evals_zeta"] = b.open_quotient_device(self.d_open.data_ptr(), n + 8, 16, self.zeta, self.alpha_open, q[0].data_ptr())
        o["open_evals_zeta_omega"] = b.open_quotient_device(op[9].data_ptr(), n + 8, 1, self.zeta_omega, self.alpha_open, q[1].data_ptr())
        q_blinds = []
        for j in range(2):
            # q has degree n + 1 (polynomials of n + 3 coefficients divided by X - z): max_power_of_2 = n, two blinds
            q_blinds.append(b.fold_blinds_device(q[j].data_ptr(), n + 2, n, self.d_fold.data_ptr() + j * n * 32))
        b.ntt_batch_device(self.d_fold.data_ptr(), self.d_fold.data_ptr(), n, 2, sync=True)
        o["cm_q"] = self._commit(self.d_fold.data_ptr(), 2, q_blinds)
        o["t_blinds"], o["q_blinds"] = t_blinds, q_blinds
        return o

    def release(self):
        self.srs.release()


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    res = {}
    for name, kw in (("window_table", dict(precompute=True)), ("general", dict(precompute=False)), ("no_shuffle_terms", dict(shuffle=False))):
        c = ProverChain(**kw)
        c.run(); b.sync()
        t = time.perf_counter()
        for _ in range(a.reps):
            c.run()
        b.sync(); torch.cuda.synchronize()
        res[name + "_ms_per_proof_chain"] = round((time.perf_counter() - t) / a.reps * 1e3, 3)
        b.profile_reset(); b.profile_enable(True); c.run(); b.sync(); b.profile_enable(False)
        tab = b.profile_table()
        res[name + "_kernel_ms"] = round(sum(ms for k, (cnt, ms) in tab.items() if not k.startswith("host_")), 3)
        res[name + "_kernels"] = {k: [cnt, round(ms, 3)] for k, (cnt, ms) in sorted(tab.items())}
        c.release()
    print(json.dumps(res))
