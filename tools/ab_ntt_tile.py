#!/usr/bin/env python3
"""A/B of the NTT pass tile (elements per workgroup): 2048 (512 threads), 1024 (256 threads), 512 (128 threads: experiment).
Outputs are compared with the default's; times are per call, device resident, best of three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 22
d_x, d_y, d_r = b.dev_alloc(nmax * 32 * 10), b.dev_alloc(nmax * 32 * 10), b.dev_alloc(nmax * 32 * 10)
b.synth_scalars(d_x, nmax * 10, 7)
for lg, batch in ((12, 1), (14, 1), (14, 10), (16, 1), (18, 1), (20, 1), (22, 1), (22, 4)):
    n = 1 << lg
    b.tune("ntt_tile", 0)
    b.ntt_batch_device(d_x, d_r, n, batch, sync=True)
    ref = b.dev_download(d_r, (batch * n, 4))
    row = []
    for tile in (2048, 1024, 512):
        b.tune("ntt_tile", tile)
        b.ntt_batch_device(d_x, d_y, n, batch, sync=True)
        same = bool(np.array_equal(b.dev_download(d_y, (batch * n, 4)), ref))
        best = 1e9
        for rnd in range(3):
            t0 = time.perf_counter()
            for _ in range(20): b.ntt_batch_device(d_x, d_y, n, batch)
            b.sync()
            best = min(best, (time.perf_counter() - t0) / 20)
        row.append(f"tile {tile}: {best * 1e6:8.1f} us {'ok' if same else 'MISMATCH'}")
    print(f"2^{lg} x{batch}: " + "  ".join(row), flush=True)
b.tune("ntt_tile", 0)
