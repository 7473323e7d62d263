#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: NTT 2^22 (both arithmetic variants) and one MSM 2^22."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
b.init(0)
n = 1 << 22
x = torch.empty((n, 4), dtype=torch.int64, device="cuda"); y = torch.empty((n, 4), dtype=torch.int64, device="cuda")
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_scalars(x.data_ptr(), n, 3); b.synth_points_random(pts.data_ptr(), n, 1)
for v in (1, 0):
    b.tune("ntt_l29", v)
    for _ in range(3): b.ntt_device(x.data_ptr(), y.data_ptr(), n, sync=True)
srs = b.Srs.from_device(pts.data_ptr(), n)
b.msm_device(srs, x.data_ptr(), n); b.sync()
