import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
n = 1 << 23
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((2 * n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), 2 * n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
r = b.msm_batch_device(srs, sc.data_ptr(), n, 2)
s0 = b.msm_device(srs, sc.data_ptr(), n); s1 = b.msm_device(srs, sc.data_ptr() + n * 32, n)
ok = np.array_equal(b.g1_to_affine(r[0]), b.g1_to_affine(s0)) and np.array_equal(b.g1_to_affine(r[1]), b.g1_to_affine(s1))
b.tune("msm_fused_hist", 0); b.tune("msm_sort_packed", 0)
r2 = b.msm_batch_device(srs, sc.data_ptr(), n, 2)
ok2 = np.array_equal(b.g1_to_affine(r2[0]), b.g1_to_affine(s0)) and np.array_equal(b.g1_to_affine(r2[1]), b.g1_to_affine(s1))
print("batch 2 x 2^23 fused+packed == singles:", ok, " unfused/unpacked == singles:", ok2)
