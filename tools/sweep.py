#!/usr/bin/env python3
"""Size sweep (device-resident inputs): MSM and NTT latency per call across sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 22
pts = torch.empty((nmax, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
out = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), nmax, 1); b.synth_scalars(sc.data_ptr(), nmax, 2)
srs = b.Srs.from_device(pts.data_ptr(), nmax)
print("MSM (uniform scalars, random points)")
for lg in (10, 12, 14, 16, 18, 20, 22):
    n = 1 << lg
    for _ in range(2): b.msm_device(srs, sc.data_ptr(), n)
    reps = 10; b.profile_reset(); b.profile_enable(True); b.sync(); t = time.perf_counter()
    for _ in range(reps): b.msm_device(srs, sc.data_ptr(), n)
    b.sync(); dt = (time.perf_counter() - t) / reps; b.profile_enable(False)
    prof = {k: v for k, v in b.profile_table().items() if not k.startswith("host_")}; ksum = sum(ms for _, ms in prof.values()) / reps
    top = sorted(prof.items(), key=lambda kv: -kv[1][1])[:4]
    print(f"  2^{lg:2d}: {dt*1e3:8.3f} ms/call  {n/dt:12.3e} pts/s  kernels {ksum:7.3f} ms | " + " ".join(f"{k.replace('msm_','')}={v[1]/reps:.3f}" for k, v in top))
print("NTT forward (device resident)")
for lg in (10, 12, 14, 16, 18, 20, 22):
    n = 1 << lg
    for _ in range(2): b.ntt_device(sc.data_ptr(), out.data_ptr(), n, sync=True)
    reps = 20; b.sync(); t = time.perf_counter()
    for _ in range(reps): b.ntt_device(sc.data_ptr(), out.data_ptr(), n)
    b.sync(); dt = (time.perf_counter() - t) / reps
    print(f"  2^{lg:2d}: {dt*1e6:9.1f} us/call  {n/dt:12.3e} elem/s")
n = 98304
for _ in range(2): b.ntt_device(sc.data_ptr(), out.data_ptr(), n, sync=True)
b.sync(); t = time.perf_counter()
for _ in range(20): b.ntt_device(sc.data_ptr(), out.data_ptr(), n)
b.sync(); print(f"  3*2^15: {(time.perf_counter()-t)/20*1e6:9.1f} us/call")
