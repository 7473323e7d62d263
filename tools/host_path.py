#!/usr/bin/env python3
"""Host-pointer entry points (what the Rust shim calls: scalars / vectors in the caller's pageable memory) next to
the device-resident ones, per call, at the prover's sizes and at the bench sizes.
usage: python tools/host_path.py [--out file]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=""); a = ap.parse_args()
b.init(0)
rows = []
def timeit(fn, reps):
    fn(); fn(); b.sync()
    t = time.perf_counter()
    for _ in range(reps): fn()
    b.sync()
    return (time.perf_counter() - t) / reps * 1e3
for n in (1 << 12, 1 << 14, 1 << 16, 1 << 20, 1 << 22, 1 << 24):
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), n, 2)
    srs = b.Srs.from_device(pts.data_ptr(), n)
    hs = np.ascontiguousarray(sc.cpu().numpy().view(np.uint64).reshape(-1, 4))
    reps = 50 if n <= (1 << 16) else 5
    dev = timeit(lambda: b.msm_device(srs, sc.data_ptr(), n), reps)
    host = timeit(lambda: b.msm(srs, hs), reps)
    rows.append({"op": "msm", "n": n, "device_ms": round(dev, 4), "host_ms": round(host, 4), "bytes_over_pcie": n * 32 + 96})
    print(rows[-1], flush=True)
    srs.release(); del pts, sc
for n in (1 << 12, 1 << 14, 3 << 15, 1 << 20, 1 << 22):
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    b.synth_scalars(x.data_ptr(), n, 3)
    hx = np.ascontiguousarray(x.cpu().numpy().view(np.uint64).reshape(-1, 4))
    reps = 50 if n <= (1 << 17) else 5
    dev = timeit(lambda: b.ntt_device(x.data_ptr(), x.data_ptr(), n, sync=True), reps)
    host = timeit(lambda: b.ntt_inplace(hx), reps)
    rows.append({"op": "ntt", "n": n, "device_ms": round(dev, 4), "host_ms": round(host, 4), "bytes_over_pcie": n * 64})
    print(rows[-1], flush=True)
if a.out:
    json.dump(rows, open(a.out, "w"), indent=1)
