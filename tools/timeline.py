#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/timeline.py`:
a few 2^24 MSMs with host-side wall-clock per call; tools/timeline_gaps.py then lists kernel gaps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
b.init(0)
lg = int(os.environ.get("LOG_N", "24")); n = 1 << lg
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
for _ in range(2): b.msm_device(srs, sc.data_ptr(), n)
b.sync()
for _ in range(3):
    t = time.perf_counter(); b.msm_device(srs, sc.data_ptr(), n); b.sync()
    print(f"wall {1e3*(time.perf_counter()-t):.3f} ms", flush=True)
