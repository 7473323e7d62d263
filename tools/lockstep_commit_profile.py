#!/usr/bin/env python3
"""One lockstep commit as the rounds issue it -- `batch` vectors of n = 2^14 scalars over the 15-bit window table -- kernel by kernel.
usage: python tools/lockstep_commit_profile.py [--batch 32] [--bits 15]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uzkge_amd import backend as b
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--bits", type=int, default=15); ap.add_argument("--log-n", type=int, default=14)
a = ap.parse_args()
b.init(0)
n = (1 << a.log_n) + 6
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
sc = torch.empty((a.batch * n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), a.batch * n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
srs.precompute(a.bits)
for _ in range(5): b.msm_batch_device(srs, sc.data_ptr(), n, a.batch)
b.sync(); t = time.perf_counter()
for _ in range(20): b.msm_batch_device(srs, sc.data_ptr(), n, a.batch)
b.sync(); wall = (time.perf_counter() - t) / 20 * 1e3
b.profile_reset(); b.profile_enable(True)
for _ in range(5): b.msm_batch_device(srs, sc.data_ptr(), n, a.batch)
b.sync(); b.profile_enable(False)
tab = b.profile_table()
print(json.dumps({"batch": a.batch, "bits": a.bits, "wall_ms": round(wall, 4), "us_per_vector": round(wall * 1e3 / a.batch, 2),
                  "kernels_ms": {k: [cnt // 5, round(ms / 5, 4)] for k, (cnt, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1])}}))
