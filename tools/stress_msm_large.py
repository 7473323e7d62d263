#!/usr/bin/env python3
"""Randomised checks of the general MSM pipeline at sizes the CPU oracle is too slow for (2^17 .. 2^22, odd sizes included):
P_i = (i + 1) Q makes MSM(P, s) = (sum_i s_i (i + 1)) Q, a closed form for any scalar set.  Scalar sets: uniform, the prover-like
mix, and uniform with planted runs of one value (long segments / bins for the sort kernels); random window widths and the round's
kernel switches; single vectors and batches of two (second vector = first, permuted halves are not needed: both must match)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bn254_py as opy, oracle_c as oc
from util import affine_of, weighted_index_sum
from uzkge_amd import backend as b

b.init(0)
rng = random.Random(int(os.environ.get("SEED", "1")))
cases = int(os.environ.get("CASES", "40"))
nmax = 1 << 22
pts = torch.empty((nmax, 8), dtype=torch.int64, device="cuda")
sc = torch.empty((2 * nmax, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
seed_int = 0x51AB5EED
b.synth_points_arith(pts.data_ptr(), nmax, oc.fr_from_ints([seed_int])[0])
srs = b.Srs.from_device(pts.data_ptr(), nmax)
q = opy.g1_mul(opy.G1_GEN, seed_int)
bad = 0
for t in range(cases):
    lg = rng.choice([17, 18, 19, 20, 20, 21, 22])
    n = (1 << lg) if rng.random() < 0.5 else rng.randrange((1 << (lg - 1)) + 1, 1 << lg)
    kind = rng.randrange(3)
    if kind == 1: b.synth_scalars_mix(sc.data_ptr(), n, rng.randrange(1 << 30))
    else: b.synth_scalars(sc.data_ptr(), n, rng.randrange(1 << 30))
    host = sc[:n].cpu().numpy().view(np.uint64).reshape(-1, 4).copy()
    if kind == 2:                                                    # planted runs of single values
        at = 0
        for _ in range(rng.randrange(1, 6)):
            ln = rng.choice([500, 3000, 12000, 30000, 70000])
            v = rng.choice([1, opy.R - 1, rng.randrange(1, 1 << 17), rng.randrange(1, opy.R)])
            if at + ln >= n: break
            host[at:at + ln] = oc.fr_from_ints([v])[0]
            at += ln + rng.randrange(0, 5000)
        sc[:n].copy_(torch.from_numpy(host.view(np.int64)).reshape(n, 4))
    batch = 2 if (rng.random() < 0.25 and lg <= 20) else 1
    if batch == 2: sc[n:2 * n].copy_(sc[:n])
    torch.cuda.synchronize()
    c = rng.choice([0, 0, 0, 12, 13, 14, 15, 16, 17])
    knobs = {"msm_seg_sort": rng.choice([1, 1, 1, 0, 12, 13, 14, 15])}
    b.set_msm_window_bits(c)
    for k, v in knobs.items(): b.tune(k, v)
    if batch == 2: got = [affine_of(j) for j in b.msm_batch_device(srs, sc.data_ptr(), n, 2)]
    else: got = [affine_of(b.msm_device(srs, sc.data_ptr(), n))]
    want = opy.g1_mul(q, weighted_index_sum(host))
    if any(g != want for g in got):
        bad += 1
        print(f"MISMATCH case {t}: n={n} kind={kind} batch={batch} c={c} {knobs}", flush=True)
    elif t % 10 == 0: print(f"case {t}: n={n} kind={kind} batch={batch} c={c} ok", flush=True)
b.set_msm_window_bits(0)
b.tune("msm_seg_sort", 1)
srs.release()
print(f"{cases} cases, {bad} mismatches")
