#!/usr/bin/env python3
"""Timing of the prover's transform shapes: batch-10 coset FFT over the 6n = 98304 domain, its
inverse, batch-10 / batch-7 transforms of n = 2^14, and the 2^22 headline, fused stages on and off.
usage: python tools/ntt_shape.py [--reps 50]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=50); a = ap.parse_args()
b.init(0)
n, m = 1 << 14, 98304
buf = torch.empty((10 * m, 4), dtype=torch.int64, device="cuda"); out = torch.empty((10 * m, 4), dtype=torch.int64, device="cuda")
big = torch.empty((1 << 22, 4), dtype=torch.int64, device="cuda"); bout = torch.empty((1 << 22, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_scalars(buf.data_ptr(), 10 * m, 5); b.synth_scalars(big.data_ptr(), 1 << 22, 6)
k = np.array([7, 0, 0, 0], dtype=np.uint64)
kk = torch.empty((1, 4), dtype=torch.int64, device="cuda"); b.synth_scalars(kk.data_ptr(), 1, 77); k = kk.cpu().numpy().view(np.uint64).reshape(4)
cases = [("coset fft 98304 x10", lambda: b.ntt_batch_device(buf.data_ptr(), out.data_ptr(), m, 10, coset_shift=k)),
         ("coset ifft 98304 x1", lambda: b.ntt_device(buf.data_ptr(), out.data_ptr(), m, inverse=True, coset_shift=k)),
         ("fft 98304 x1", lambda: b.ntt_device(buf.data_ptr(), out.data_ptr(), m)),
         ("ifft 2^14 x10", lambda: b.ntt_batch_device(buf.data_ptr(), out.data_ptr(), n, 10, inverse=True)),
         ("fft 2^14 x7", lambda: b.ntt_batch_device(buf.data_ptr(), out.data_ptr(), n, 7)),
         ("fft 2^14 x1", lambda: b.ntt_device(buf.data_ptr(), out.data_ptr(), n)),
         ("fft 2^22", lambda: b.ntt_device(big.data_ptr(), bout.data_ptr(), 1 << 22)),
         ("coset fft 2^22", lambda: b.ntt_device(big.data_ptr(), bout.data_ptr(), 1 << 22, coset_shift=k)),
         ("coset ifft 2^22", lambda: b.ntt_device(big.data_ptr(), bout.data_ptr(), 1 << 22, inverse=True, coset_shift=k))]
for fused in (1,):
    for name, fn in cases:
        fn(); b.sync(); t = time.perf_counter()
        for _ in range(a.reps): fn()
        b.sync(); dt = (time.perf_counter() - t) / a.reps * 1e3
        b.profile_reset(); b.profile_enable(True); fn(); b.sync(); b.profile_enable(False)
        tab = b.profile_table()
        ks = " ".join(f"{kn.replace('ntt_', '')}={ms * 1e3:.0f}x{cnt}" for kn, (cnt, ms) in sorted(tab.items()))
        print(f"fused={fused} {name:22s} {dt * 1e3:8.1f} us | {ks}", flush=True)
