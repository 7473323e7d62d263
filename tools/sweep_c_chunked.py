#!/usr/bin/env python3
"""Wider windows at 2^22 .. 2^24 through point chunks: a 19-bit window (14 windows instead of 15) has 18 key bits, which fit the
4-byte packed sort entry only beside <= 22 index bits -- i.e. when the MSM runs as chunks of 2^22 points into one bucket set
(uzk_tune("msm_chunk_log", 22), the path of > 2^24-point MSMs).  Times (window bits, chunk log) pairs against the default; the
results must be the same point.  usage: python tools/sweep_c_chunked.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 24
pts = torch.empty((nmax, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), nmax, 1); b.synth_scalars(sc.data_ptr(), nmax, 2)
srs = b.Srs.from_device(pts.data_ptr(), nmax)
cfgs = [(0, 26), (17, 26), (17, 22), (18, 22), (19, 22), (19, 21), (20, 21), (20, 20)]
for lg in (22, 23, 24):
    n = 1 << lg
    res, ref = {}, None
    for c, ch in cfgs:
        b.set_msm_window_bits(c); b.tune("msm_chunk_log", ch)
        out = b.msm_device(srs, sc.data_ptr(), n)
        aff = b.g1_to_affine(out) if hasattr(b, "g1_to_affine") else out
        if ref is None: ref = np.array(aff).tobytes()
        elif hasattr(b, "g1_to_affine"): assert np.array(aff).tobytes() == ref, (c, ch)
    for rd in range(3):
        for c, ch in cfgs:
            b.set_msm_window_bits(c); b.tune("msm_chunk_log", ch); b.sync(); t = time.perf_counter()
            b.msm_device(srs, sc.data_ptr(), n); b.sync()
            res.setdefault((c, ch), []).append((time.perf_counter() - t) * 1e3)
    print(f"2^{lg}: " + "  ".join(f"c={c},chunk=2^{ch}: {np.median(v):.3f}" for (c, ch), v in res.items()), flush=True)
b.set_msm_window_bits(0); b.tune("msm_chunk_log", 26)
