import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
n = 1 << 24
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
for mode in ("general", 20, 22):
    if mode == "general": b.tune("msm_no_precompute", 1)
    else:
        srs.precompute(mode); b.tune("msm_no_precompute", 0)
    b.msm_device(srs, sc.data_ptr(), n); b.sync()
    b.profile_reset(); b.profile_enable(True)
    t = time.perf_counter(); b.msm_device(srs, sc.data_ptr(), n); b.sync(); dt = time.perf_counter() - t
    b.profile_enable(False)
    print(mode, f"{dt*1e3:.2f} ms", " ".join(f"{k.replace('msm_','')}={cnt}x{ms/max(cnt,1):.3f}" for k, (cnt, ms) in sorted(b.profile_table().items())), flush=True)
