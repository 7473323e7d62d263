import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 24
pts = torch.empty((nmax, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), nmax, 1); b.synth_scalars(sc.data_ptr(), nmax, 2)
srs = b.Srs.from_device(pts.data_ptr(), nmax)
b.tune("msm_small", 0)
for lg, cs in ((15, range(8, 15)), (16, range(8, 16)), (17, range(8, 17)), (18, range(12, 18)), (19, range(13, 18)), (20, range(14, 19)), (21, range(15, 19)), (22, range(15, 20)), (23, range(16, 20)), (24, range(16, 20))):
    n = 1 << lg
    res = {}
    for c in cs:
        b.set_msm_window_bits(c); b.msm_device(srs, sc.data_ptr(), n)
    for rd in range(3):
        for c in cs:
            b.set_msm_window_bits(c); b.sync(); t = time.perf_counter()
            b.msm_device(srs, sc.data_ptr(), n); b.sync()
            res.setdefault(c, []).append((time.perf_counter() - t) * 1e3)
    best = min(res, key=lambda c: np.median(res[c]))
    print(f"2^{lg}: best c={best}  " + " ".join(f"{c}:{np.median(v):.3f}" for c, v in res.items()), flush=True)
