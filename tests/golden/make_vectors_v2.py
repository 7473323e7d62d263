#!/usr/bin/env python3
"""Generates tests/golden/vectors_v2.npz: the expected outputs of one prover-round chain (tools/prover_chain.py) at
n = 2^12 -- ChainInputs(4096, seed 7): the reference's lagrange-srs-4096.bin / srs-padding.bin, synthetic circuit -- from the
CPU oracle chain (tests/chain_oracle.py): the 16 commitments (affine, wire format), the evaluation vectors, the blinds of
the folds, and SHA-256 digests of the large intermediate vectors (coefficients, coset evaluations, quotient, t).
tests/test_golden_vectors.py (CPU) holds the oracle to the file, tests/test_gpu_golden.py the GPU chain -- without running
the oracle.  Run from the repo root:  python tests/golden/make_vectors_v2.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
N, SEED = 4096, 7
BIG = ("coefs", "coset_evals", "t_quotient", "t", "z_evals", "r")


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).digest(), dtype=np.uint8).copy()


def expected():
    from chain_oracle import oracle_chain
    from prover_chain import ChainInputs
    o = oracle_chain(ChainInputs(N, SEED))
    out = {"n": np.array([N]), "seed": np.array([SEED])}
    for k in ("cm_w_wsel", "cm_z", "cm_t", "cm_q", "evals_zeta", "z_eval_zeta_omega", "open_evals_zeta", "open_evals_zeta_omega"):
        out[k] = np.asarray(o[k], dtype=np.uint64)
    out["t_blinds"] = np.concatenate([b.reshape(-1, 4) for b in o["t_blinds"]])
    out["q_blinds"] = np.concatenate([b.reshape(-1, 4) for b in o["q_blinds"]])
    for k in BIG:
        out["sha256_" + k] = digest(o[k])
    return out


if __name__ == "__main__":
    out = expected()
    np.savez_compressed(os.path.join(HERE, "vectors_v2.npz"), **out)
    print("wrote vectors_v2.npz:", {k: v.shape for k, v in out.items()})
