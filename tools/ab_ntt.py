#!/usr/bin/env python3
"""Interleaved A/B of the NTT pass kernels: 8x32-bit relaxed Montgomery vs 9x29-bit lazy-carry limbs.
Checks that both give identical outputs, then times them alternately (per-call, device resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 24
src = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
o0 = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
o1 = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
b.synth_scalars(src.data_ptr(), nmax, 7)
for lg in (12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24):
    n = 1 << lg
    res = {}
    for inv in (False, True):
        b.tune("ntt_l29", 0); b.ntt_device(src.data_ptr(), o0.data_ptr(), n, inverse=inv, sync=True)
        b.tune("ntt_l29", 1); b.ntt_device(src.data_ptr(), o1.data_ptr(), n, inverse=inv, sync=True)
        same = bool(torch.equal(o0[:n], o1[:n]))
        res[inv] = same
    t = [0.0, 0.0]
    reps = 20 if lg <= 22 else 8
    for rnd in range(3):
        for v in (0, 1):
            b.tune("ntt_l29", v)
            b.ntt_device(src.data_ptr(), o0.data_ptr(), n, sync=True)
            t0 = time.perf_counter()
            for _ in range(reps): b.ntt_device(src.data_ptr(), o0.data_ptr(), n)
            b.sync(); dt = (time.perf_counter() - t0) / reps
            t[v] = dt if rnd == 0 else min(t[v], dt)
    print(f"2^{lg:2d}: equal fwd={res[False]} inv={res[True]}  fp256 {t[0]*1e6:9.1f} us  l29 {t[1]*1e6:9.1f} us  ({(t[1]/t[0]-1)*100:+.1f} %)", flush=True)
