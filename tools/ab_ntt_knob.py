#!/usr/bin/env python3
"""Interleaved A/B of one uzk_tune knob on the NTT: python tools/ab_ntt_knob.py ntt_prio 0 1 2 3 [--logs 18 20 22]
Checks identical outputs against the first value, then times the values alternately (device resident, per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
args = sys.argv[1:]
logs = [18, 20, 22]
if "--logs" in args:
    i = args.index("--logs"); logs = [int(x) for x in args[i + 1:]]; args = args[:i]
key, vals = args[0], [int(v) for v in args[1:]]
b.init(0)
nmax = 1 << max(logs)
src = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
o0 = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
o1 = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
b.synth_scalars(src.data_ptr(), nmax, 7)
for lg in logs:
    n = 1 << lg
    b.tune(key, vals[0]); b.ntt_device(src.data_ptr(), o0.data_ptr(), n, sync=True)
    for v in vals[1:]:
        b.tune(key, v); b.ntt_device(src.data_ptr(), o1.data_ptr(), n, sync=True)
        assert bool(torch.equal(o0[:n], o1[:n])), f"{key}={v} changes the transform"
    t = {v: 1e9 for v in vals}
    reps = 50 if lg <= 22 else 10
    for rnd in range(5):
        for v in vals:
            b.tune(key, v)
            b.ntt_device(src.data_ptr(), o1.data_ptr(), n, sync=True)
            t0 = time.perf_counter()
            for _ in range(reps): b.ntt_device(src.data_ptr(), o1.data_ptr(), n)
            b.sync(); t[v] = min(t[v], (time.perf_counter() - t0) / reps)
    print(f"2^{lg:2d}: " + "  ".join(f"{key}={v}: {t[v]*1e6:8.1f} us" for v in vals), flush=True)
b.tune(key, vals[0])
