#!/usr/bin/env python3
"""Best MSM window size per problem size (general mode), one process, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 22
pts = torch.empty((nmax, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), nmax, 1); b.synth_scalars(sc.data_ptr(), nmax, 2)
srs = b.Srs.from_device(pts.data_ptr(), nmax)
for lg in (10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22):
    n = 1 << lg
    res = {}
    for c in range(6, 19):
        b.set_msm_window_bits(c)
        b.msm_device(srs, sc.data_ptr(), n)
    for rd in range(3):
        for c in range(6, 19):
            b.set_msm_window_bits(c); b.sync(); t = time.perf_counter()
            b.msm_device(srs, sc.data_ptr(), n); b.sync()
            res.setdefault(c, []).append((time.perf_counter() - t) * 1e3)
    best = min(res, key=lambda c: np.median(res[c]))
    print(f"2^{lg}: best c={best}  " + " ".join(f"{c}:{np.median(v):.2f}" for c, v in res.items()))
