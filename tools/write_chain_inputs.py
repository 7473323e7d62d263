#!/usr/bin/env python3
"""Writes the input files of tests/cpp/prover_rounds for ChainInputs(n, seed) into a directory.
usage: python tools/write_chain_inputs.py <dir> [log_n=14] [seed=11] [precompute=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from prover_chain import ChainInputs
from test_gpu_cpp_mirror import _write_inputs
d = sys.argv[1]
os.makedirs(d, exist_ok=True)
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 1       # uzk_circuit_desc.precompute: 0, 1 (automatic) or a window width
_write_inputs(ChainInputs(1 << log_n, seed), d, precompute=pre)
print("wrote", d)
