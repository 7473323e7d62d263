import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 24
pts = torch.empty((nmax, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), nmax, 1); b.synth_scalars(sc.data_ptr(), nmax, 2)
srs = b.Srs.from_device(pts.data_ptr(), nmax)
for lg in range(12, 25):
    n = 1 << lg
    b.msm_device(srs, sc.data_ptr(), n)
    ts = []
    for r in range(7):
        b.sync(); t = time.perf_counter(); b.msm_device(srs, sc.data_ptr(), n); b.sync(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"2^{lg}: {np.median(ts):.3f} ms  c={b.msm_plan_info(n)[0]}", flush=True)
