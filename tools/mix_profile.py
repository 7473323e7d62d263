import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
for lg in (24, 20):
    n = 1 << lg
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars_mix(sc.data_ptr(), n, 2)
    srs = b.Srs.from_device(pts.data_ptr(), n)
    b.msm_device(srs, sc.data_ptr(), n)
    ts = []
    for r in range(5):
        b.sync(); t = time.perf_counter(); b.msm_device(srs, sc.data_ptr(), n); b.sync(); ts.append((time.perf_counter() - t) * 1e3)
    b.profile_reset(); b.profile_enable(True)
    b.msm_device(srs, sc.data_ptr(), n); b.sync(); b.profile_enable(False)
    print(f"2^{lg} prover-mix: {np.median(ts):.3f} ms |", " ".join(f"{k.replace('msm_','')}={ms:.3f}x{cnt}" for k, (cnt, ms) in sorted(b.profile_table().items())))
    srs.release()
