#!/usr/bin/env python3
"""MSM time on skewed scalar sets (the value classes of a real witness) vs uniform."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from uzkge_amd import backend as b
from uzkge_amd.poly_commit import fr_from_int, FR_MODULUS
b.init(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
uni = sc.cpu().numpy().view(np.uint64).reshape(-1, 4)
one, m1 = fr_from_int(1), fr_from_int(FR_MODULUS - 1)
rng = np.random.default_rng(1)
def mix():
    a = uni.copy(); cls = rng.integers(0, 10, n)
    a[cls < 5] = 0; a[(cls >= 5) & (cls < 7)] = one; a[cls == 7] = m1
    small = cls == 8; a[small, 1:] = 0; a[small, 0] &= np.uint64(0xFFFF)     # not Montgomery-small, but narrow values
    return a
sets = {"uniform": uni, "all_ones": np.tile(one, (n, 1)), "all_minus_one": np.tile(m1, (n, 1)), "prover_mix": mix()}
for name, arr in sets.items():
    t = torch.from_numpy(arr.view(np.int64)).cuda(); torch.cuda.synchronize()
    b.msm_device(srs, t.data_ptr(), n); b.sync(); t0 = time.perf_counter()
    for _ in range(3): b.msm_device(srs, t.data_ptr(), n)
    b.sync(); print(f"2^{lg} {name:14s} {(time.perf_counter()-t0)/3*1e3:9.3f} ms")
