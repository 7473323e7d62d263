#!/usr/bin/env python3
"""Generates tests/golden/vectors_v3.npz: the expected outputs of one prover-round chain (tools/prover_chain.py, the
reference's call mix: 9 + 1 iFFT, 8 + 1 + 5 + 2 commits with blinds, ten coset FFT(6n), quotient, split_t with chunk n + 2,
19 evaluations, r_poly over 43 polynomials, openings of 16 and 4 polynomials) at n = 2^12 -- ChainInputs(4096, seed 7): the
reference's lagrange-srs-4096.bin / srs-padding.bin, synthetic circuit -- from the CPU oracle chain (tests/chain_oracle.py):
the 16 commitments (affine, wire format), the evaluations, the blinds of the folds, and SHA-256 digests of the large
intermediate vectors.  tests/test_golden_vectors.py (CPU) holds the oracle to the file, tests/test_gpu_golden.py the Python
chain and tests/test_gpu_cpp_mirror.py the C++ driver -- without running the oracle.
Run from the repo root:  python tests/golden/make_vectors_v3.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
N, SEED = 4096, 7
SMALL = ("cm_w_wsel", "cm_z", "cm_t", "cm_q", "evals", "t_blinds", "q_blinds")
BIG = ("coefs", "coset_evals", "t_quotient", "t", "z_evals", "r", "chunks", "quotients", "tables")


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).digest(), dtype=np.uint8).copy()


def expected():
    from chain_oracle import oracle_chain
    from prover_chain import ChainInputs
    o = oracle_chain(ChainInputs(N, SEED))
    out = {"n": np.array([N]), "seed": np.array([SEED])}
    for k in SMALL:
        out[k] = np.asarray(o[k], dtype=np.uint64)
    for k in BIG:
        out["sha256_" + k] = digest(o[k])
    return out


if __name__ == "__main__":
    out = expected()
    np.savez_compressed(os.path.join(HERE, "vectors_v3.npz"), **out)
    print("wrote vectors_v3.npz:", {k: v.shape for k, v in out.items()})
