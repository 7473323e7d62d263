#!/usr/bin/env python3
"""Samples the GPU's shader clock and power (rocm-smi) while an MSM 2^24 loop runs: is the long
accumulation kernel running at the 2.4 GHz the microbenchmarks assume?"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
b.init(0)
n = 1 << 24
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda"); sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 1); b.synth_scalars(sc.data_ptr(), n, 2)
srs = b.Srs.from_device(pts.data_ptr(), n)
def smi(tag):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "junction", "fclk"))]
    print(tag, " | ".join(keep), flush=True)
smi("idle:")
stop = False
def sampler():
    k = 0
    while not stop:
        time.sleep(1.0); k += 1; smi(f"load t={k}s:")
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 6.0:
    b.msm_device(srs, sc.data_ptr(), n); reps += 1
b.sync(); dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{reps} MSMs in {dt:.2f} s = {dt/reps*1e3:.2f} ms each")
# the same for back-to-back 2^22 NTTs
nn = 1 << 22
x = sc[:nn]; y = torch.empty((nn, 4), dtype=torch.int64, device="cuda")
b.ntt_device(x.data_ptr(), y.data_ptr(), nn, sync=True)
time.sleep(2.0); smi("idle:")
stop = False
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(50): b.ntt_device(x.data_ptr(), y.data_ptr(), nn)
    b.sync(); reps += 50
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{reps} NTTs 2^22 in {dt:.2f} s = {dt/reps*1e3:.4f} ms each")
