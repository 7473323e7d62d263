#!/usr/bin/env python3
"""Fixed NTT-only workload for rocprofv3 --pmc passes: three forward 2^22 transforms (default kernels), device-resident.
usage (on the GPU box, program directly after `--`): rocprofv3 --pmc <counters> --output-format csv -d out -- python3 tools/pmc_ntt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uzkge_amd import backend as b
b.init(0)
n = 1 << 22
for key, val in [a.split("=") for a in sys.argv[1:]]:
    b.tune(key, int(val))
d_x, d_y = b.dev_alloc(n * 32), b.dev_alloc(n * 32)
b.synth_scalars(d_x, n, 3)
for _ in range(3):
    b.ntt_device(d_x, d_y, n, sync=True)
