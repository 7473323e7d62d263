#!/usr/bin/env python3
"""Interleaved A/B timing of MSM tuning knobs in ONE process (guide rule 24).
usage: python tools/ab_msm.py --log-n 24 --rounds 5 key=value[,key=value] ...   (each arg = one variant)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=24)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
b.init(0)
n = 1 << a.log_n
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
DEFAULTS = {"window_bits": 0, "precompute": -1, "msm_seg_sort": 1, "msm_chunk_log": 26, "msm_small": 1}
def apply(v):
    cfg = dict(DEFAULTS)
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
res = {v: [] for v in a.variants}; kern = {v: {} for v in a.variants}
ref = None
for v in a.variants:
    apply(v); r = b.msm_device(srs, sc.data_ptr(), n)
    aff = b.g1_to_affine(r)
    if ref is None: ref = aff
    assert np.array_equal(aff, ref), f"variant {v} changes the result"
for rd in range(a.rounds):
    for v in a.variants:
        apply(v); b.profile_reset(); b.profile_enable(True); b.sync()
        t = time.perf_counter(); b.msm_device(srs, sc.data_ptr(), n); b.sync(); dt = time.perf_counter() - t
        b.profile_enable(False); res[v].append(dt * 1e3)
        for k, (cnt, ms) in b.profile_table().items(): kern[v].setdefault(k, []).append(ms / max(cnt, 1))
for v in a.variants:
    ks = " ".join(f"{k.replace('msm_','')}={np.median(x):.3f}" for k, x in sorted(kern[v].items()))
    print(f"{v:40s} median {np.median(res[v]):8.3f} ms  min {min(res[v]):8.3f} ms | {ks}")
