#!/usr/bin/env python3
"""Randomised cross-check of the GPU MSM against the CPU oracle (test infrastructure): many small and
medium cases with adversarial structure -- repeated and opposite points (degenerate additions -> the
exception kernel), points at infinity, tiny / huge / boolean scalars, every window size."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bn254_py as opy, oracle_c as oc
from util import rand_fr_wire, load_srs
from uzkge_amd import backend as b

b.init(0)
srs_wire, _ = load_srs("lagrange-srs-16384.bin")
rng = random.Random(int(os.environ.get("SEED", "1")))
nrng = np.random.default_rng(rng.randrange(1 << 30))
cases = int(os.environ.get("CASES", "150"))
one = oc.fr_from_ints([1])[0]; rm1 = oc.fr_from_ints([opy.R - 1])[0]
bad = 0
for t in range(cases):
    n = rng.choice([1, 2, 3, 17, 64, 255, 1000, 4096, 9000, 16384, 16384, 40000, 1 << 16, 100000, 1 << 17, 1 << 18])
    if int(os.environ.get("MAX_N", "0")): n = min(n, int(os.environ["MAX_N"]))
    idx = nrng.integers(0, 16384, n)                                # sizes above 2^14 draw with repetition: the general pipeline with duplicate bases
    mode = rng.randrange(5)
    if mode == 1: idx[:] = idx[0]                                  # one point repeated
    elif mode == 2: idx = idx[nrng.integers(0, max(1, n // 8), n)]  # few distinct points
    pts = srs_wire[idx].copy()
    if mode == 3:                                                   # half the points negated copies
        half = n // 2
        pts[half:2 * half] = pts[:half]
        for i in range(half, 2 * half):
            y = opy.limbs_to_int(pts[i, 4:8])
            pts[i, 4:8] = np.array(opy.int_to_limbs((opy.P - y) % opy.P), dtype=np.uint64) if y else 0
    if mode == 4 and n > 2: pts[nrng.integers(0, n, n // 4)] = 0    # points at infinity
    sc = rand_fr_wire(n, rng.randrange(1 << 30))
    smode = rng.randrange(5)
    if smode == 1: sc[nrng.integers(0, 2, n) == 1] = one
    elif smode == 2: sc[nrng.integers(0, 3, n) == 1] = rm1
    elif smode == 3: sc[:] = sc[0]                                  # identical scalars (with repeated points: doublings)
    elif smode == 4: sc[nrng.integers(0, 2, n) == 1] = 0
    c = rng.choice([0, 0, 0, 4, 5, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18])
    b.set_msm_window_bits(c)
    b.tune("msm_seg_sort", rng.choice([1, 1, 1, 0, 10, 11, 12, 13, 14, 15]))      # the segment sort kernels (any instantiation) or the generic pass
    b.tune("msm_small", rng.choice([1, 1, 1, 0]))                   # small pipeline or the general one
    if rng.random() < 0.3:                                           # batched entry point: the same vector three times over
        h = b.Srs.from_host(pts)
        if rng.random() < 0.5: h.precompute(rng.choice([0, 5, 8, 10]))
        outs = b.msm_batch(h, np.stack([sc, sc, sc]))
        h.release()
        got = outs[rng.randrange(3)]
    else:
        got = b.msm_raw(pts, sc)
    want = oc.msm_pippenger(pts, sc, 0, 8)
    if oc.jac_to_affine_ints(got) != oc.jac_to_affine_ints(want):
        bad += 1
        print(f"MISMATCH case {t}: n={n} mode={mode} smode={smode} c={c}", flush=True)
b.tune("msm_seg_sort", 1)
b.set_msm_window_bits(0); b.tune("msm_small", 1)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
