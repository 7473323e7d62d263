#!/usr/bin/env python3
"""The accumulator's product formulations A/B'd where the kernel lives: inside the 2^24-point MSM, for seconds at a time, under
the package's power cap -- ms per accumulation launch, shader clock and socket power DURING the runs, joules per mixed addition.
(the alternatives were only ever priced in sub-millisecond microbenchmarks at the boost clock.)

  variant 0   9 x 29-bit limbs, lazy carries: 162 v_mad_u64_u32 + 43 bookkeeping instructions per product (ec29.hpp)
  variant 2   8 x 32-bit limbs, relaxed Montgomery, FIPS product scanning in assembly: 128 MADs + 128 add-with-carry (ec.hpp)
  variant 1   the same arithmetic with canonical (fully reduced) results after every operation

usage: python tools/ab_acc_power.py [--log-n 24] [--seconds 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from uzkge_amd import backend as b

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=24)
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--variants", default="0,2,1,0")
a = ap.parse_args()
b.init(0)
n = 1 << a.log_n
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1)
b.synth_scalars(sc.data_ptr(), n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
cbits, nwin = b.msm_plan_info(n)
adds = n * nwin
ref = None
for v in [int(x) for x in a.variants.split(",")]:
    b.tune("msm_acc_variant", v)
    r = b.msm_device(srs, sc.data_ptr(), n)
    aff = b.g1_to_affine(r).tobytes()
    ref = ref or aff
    t_end = time.perf_counter() + 1.0                         # ramp: a second of the same work before the measured window
    while time.perf_counter() < t_end:
        b.msm_device(srs, sc.data_ptr(), n)
    b.sync()
    b.profile_reset(); b.profile_enable(True)
    sampler = bench._ClockSampler(0).start()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < a.seconds:
        b.msm_device(srs, sc.data_ptr(), n)
        reps += 1
    b.sync()
    t1 = time.perf_counter()
    clk = sampler.stop(t0, t1) or {}
    b.profile_enable(False)
    ms, cnt = b.profile_get("msm_accumulate")
    acc_ms = ms / max(cnt, 1)
    out = {"variant": v, "same_point_as_variant_0": aff == ref, "msm_ms": round((t1 - t0) / reps * 1e3, 3), "accumulate_ms": round(acc_ms, 3),
           "sclk_mhz": clk.get("sclk_mhz_timed"), "power_w": clk.get("power_w_timed"), "samples": clk.get("samples"),
           "mixed_additions": adds, "ns_per_1e3_additions": round(acc_ms * 1e6 / adds * 1e3, 3)}
    if clk.get("power_w_timed"):
        out["microjoules_per_addition_socket"] = round(clk["power_w_timed"] * acc_ms * 1e-3 / adds * 1e6, 4)
        out["gcycles_per_launch"] = round(clk["sclk_mhz_timed"] * 1e6 * acc_ms * 1e-3 / 1e9, 3)
    print(json.dumps(out), flush=True)
b.tune("msm_acc_variant", 0)
