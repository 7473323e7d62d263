import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
b.init(0)
n = 1 << 14; B = int(os.environ.get("BATCH", "8"))
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), B * n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
for _ in range(3): b.msm_batch_device(srs, sc.data_ptr(), n, B)
b.profile_reset(); b.profile_enable(True); b.sync(); t = time.perf_counter()
for _ in range(10): b.msm_batch_device(srs, sc.data_ptr(), n, B)
b.sync(); dt = (time.perf_counter() - t) / 10; b.profile_enable(False)
prof = b.profile_table()
print(f"batch {B} x 2^14: {dt*1e3:.3f} ms/call; kernels: " + " ".join(f"{k.replace('msm_','')}={v[1]/10:.3f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])))
