#!/usr/bin/env python3
"""Randomised cross-check of the GPU NTT against the CPU oracle (test infrastructure): random supported sizes
(2^k and 3*2^k), forward / inverse, coset shifts, batches, both arithmetic variants, both twiddle products."""
import os, sys, random
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import bn254_py as opy, oracle_c as oc
from util import rand_fr_wire
from uzkge_amd import backend as b

b.init(0)
rng = random.Random(int(os.environ.get("SEED", "1")))
cases = int(os.environ.get("CASES", "120"))
bad = 0
for t in range(cases):
    k = rng.randrange(0, 25)            # up to 2^24 and 3 * 2^22: two-pass plans of 2^17 .. 2^21, their three-pass form, 8 + 8 + 7 / 8
    n = (1 << k) if rng.random() < 0.6 else 3 * (1 << min(k, 22))
    batch = rng.choice([1, 1, 2, 5]) if n <= (1 << 21) else 1
    inv = rng.random() < 0.5
    shift = rand_fr_wire(1, rng.randrange(1 << 30))[0] if rng.random() < 0.4 else None
    b.tune("ntt_tile", rng.choice([0, 0, 1024, 2048]))
    b.tune("ntt_two_pass", rng.choice([1, 1, 0]))
    x = rand_fr_wire(n * batch, rng.randrange(1 << 30)).reshape(batch, n, 4)
    got = b.ntt_batch(x, inverse=inv, coset_shift=shift)
    for j in range(batch):
        v = x[j]
        if shift is not None and not inv: v = oc.mul_var(v, shift)
        w = oc.ntt(v, inverse=inv, threads=int(os.environ.get("THREADS", "16")))
        if shift is not None and inv:
            w = oc.mul_var(w, shift)      # the ABI post-scales by shift^j on the inverse (caller passes k^-1)
        if not np.array_equal(got[j], w):
            bad += 1; print(f"MISMATCH case {t}: n={n} batch={batch} inv={inv} shift={shift is not None}", flush=True); break
b.tune("ntt_tile", 0)
b.tune("ntt_two_pass", 1)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
