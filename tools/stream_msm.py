#!/usr/bin/env python3
"""The host-scalar MSM (uzk_msm_g1: what a two-call-site integration calls) at large sizes: streamed point chunks under the
accumulation vs upload-then-compute, against the device-resident call.  Results are compared (affine) with the device call.
usage: python tools/stream_msm.py [--log-n 24] [--logs 20,21,22]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uzkge_amd import backend as b

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=24)
ap.add_argument("--logs", default="20,21,22")
a = ap.parse_args()
b.init(0)
n = 1 << a.log_n
d_pts, d_sc = b.dev_alloc(n * 64), b.dev_alloc(n * 32)
b.synth_points_random(d_pts, n, 1); b.synth_scalars(d_sc, n, 2)
srs = b.Srs.from_device(d_pts, n)
hs = b.dev_download(d_sc, (n, 4))
def timeit(fn, reps=4):
    fn(); b.sync()
    t = time.perf_counter()
    for _ in range(reps): r = fn()
    b.sync()
    return (time.perf_counter() - t) / reps * 1e3, r
res = {"log_n": a.log_n}
dev_ms, ref = timeit(lambda: b.msm_device(srs, d_sc, n))
res["device_resident_ms"] = round(dev_ms, 3)
want = b.g1_to_affine(ref)
b.tune("msm_stream_log", -1)
ms, r = timeit(lambda: b.msm(srs, hs))
res["upload_then_compute_ms"] = round(ms, 3)
assert np.array_equal(b.g1_to_affine(r), want)
for lg in [int(x) for x in a.logs.split(",")]:
    b.tune("msm_stream_log", lg)
    ms, r = timeit(lambda: b.msm(srs, hs))
    assert np.array_equal(b.g1_to_affine(r), want), lg
    res[f"streamed_chunk_2^{lg}_ms"] = round(ms, 3)
b.tune("msm_stream_log", 0)
ms, r = timeit(lambda: b.msm(srs, hs))
assert np.array_equal(b.g1_to_affine(r), want)
res["streamed_default_schedule_ms"] = round(ms, 3)
b.profile_reset(); b.profile_enable(True); b.msm(srs, hs); b.profile_enable(False)
res["profile_default"] = {k: [c, round(ms, 3)] for k, (c, ms) in sorted(b.profile_table().items())}
print(json.dumps(res))
