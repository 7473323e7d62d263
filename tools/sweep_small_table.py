#!/usr/bin/env python3
"""The prover's commits (n = 2^14 + 6 tail scalars, batches of 8 / 1 / 5 / 2) against the window width of the SRS table
(uzk_srs_precompute): per-call time and the sum over one proof's four commits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np
from uzkge_amd import backend as b
from prover_chain import ChainInputs
inp = ChainInputs(1 << 14, 11)
b.init(0)
n = inp.n
d_s = b.dev_alloc(8 * n * 32); b.synth_scalars(d_s, 8 * n, 5)
tail = np.zeros((8, 6, 4), dtype=np.uint64); tail[:, :, 0] = 7
ref = {}
for c in (0, 7, 8, 9, 10, 11):
    srs = b.Srs.from_host(inp.bases)
    if c: srs.precompute(c)
    row = []
    tot = 0.0
    for batch in (8, 1, 5, 2):
        r = b.msm_batch_tail_device(srs, d_s, n, n, batch, tail[:batch], 6)
        aff = np.stack([b.g1_to_affine(x) for x in r])
        if batch in ref: assert np.array_equal(aff, ref[batch]), (c, batch)
        else: ref[batch] = aff
        best = 1e9
        for rnd in range(3):
            t = time.perf_counter()
            for _ in range(20): b.msm_batch_tail_device(srs, d_s, n, n, batch, tail[:batch], 6)
            best = min(best, (time.perf_counter() - t) / 20)
        row.append(f"b{batch}: {best * 1e6:6.1f}")
        tot += best
    print(f"table c={c or 'none':>4}: " + "  ".join(row) + f"   sum {tot * 1e6:7.1f} us", flush=True)
    srs.release()
