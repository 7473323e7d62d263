#!/usr/bin/env python3
"""Timing of the small-problem MSM pipeline (msm_small_*) against the general one, single and batched,
uniform and prover-mix scalars, with the per-kernel table of the library's own HIP events.
usage: python tools/small_msm.py [--log-n 14] [--reps 20] [variant ...]   variant = key=value[,key=value]"""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=14)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batches", default="1,2,5,8")
ap.add_argument("variants", nargs="*", default=["msm_small=1", "msm_small=0"])
a = ap.parse_args()
b.init(0)
n = 1 << a.log_n
B = 16
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
sc = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
mx = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), B * n, 2); b.synth_scalars_mix(mx.data_ptr(), B * n, 3)
srs = b.Srs.from_device(pts.data_ptr(), n)
DEF = {"msm_small": 1, "window_bits": 0, "precompute": -1}
def apply(v):
    cfg = dict(DEF)
    for kv in v.split(","):
        if kv and kv != "base":
            k, x = kv.split("="); cfg[k] = int(x)
    b.set_msm_window_bits(cfg.pop("window_bits"))
    pc = cfg.pop("precompute")
    if pc >= 0:
        srs.precompute(pc); b.tune("msm_no_precompute", 0)
    else:
        b.tune("msm_no_precompute", 1)
    for k, x in cfg.items(): b.tune(k, x)
ref = {}
for v in a.variants:
    apply(v)
    for name, buf in (("uniform", sc), ("mix", mx)):
        for bt in [int(x) for x in a.batches.split(",")]:
            fn = (lambda: b.msm_device(srs, buf.data_ptr(), n)) if bt == 1 else (lambda: b.msm_batch_device(srs, buf.data_ptr(), n, bt))
            r = np.atleast_2d(fn())
            aff = np.stack([b.g1_to_affine(x) for x in r])
            key = (name, bt)
            if key not in ref: ref[key] = aff
            assert np.array_equal(aff, ref[key]), f"variant {v} changes the result ({key})"
            b.sync(); t = time.perf_counter()
            for _ in range(a.reps): fn()
            b.sync(); dt = (time.perf_counter() - t) / a.reps * 1e3
            b.profile_reset(); b.profile_enable(True)
            for _ in range(5): fn()
            b.sync(); b.profile_enable(False)
            tab = b.profile_table()
            ks = " ".join(f"{k.replace('msm_', '').replace('host_', 'h:')}={ms / max(cnt, 1) * 1e3:.0f}" for k, (cnt, ms) in sorted(tab.items()))
            print(f"{v:28s} {name:8s} batch {bt:2d}: {dt:7.3f} ms/call {dt / bt * 1e3:7.1f} us/msm | us: {ks}", flush=True)
