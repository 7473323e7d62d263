#!/usr/bin/env python3
"""Interleaved A/B of the NTT pass tile size (uzk_tune "ntt_tile": 2048 elements / 512 threads vs 1024 / 256): identical
outputs, alternating timing, single transforms and the prover's batched shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uzkge_amd import backend as b
b.init(0)
nmax = 1 << 24
src = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
o0 = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
o1 = torch.empty((nmax, 4), dtype=torch.int64, device="cuda")
b.synth_scalars(src.data_ptr(), nmax, 7)
shift = src[:1].cpu().numpy().view("uint64")[0]
cases = [(1 << lg, 1, None) for lg in (12, 14, 16, 18, 20, 22, 24)] + [(1 << 14, 10, None), (98304, 10, shift), (98304, 1, shift), (3 << 20, 1, None)]
for n, batch, cs in cases:
    same = True
    for inv in (False, True):
        b.tune("ntt_tile", 2048); b.ntt_batch_device(src.data_ptr(), o0.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
        b.tune("ntt_tile", 1024); b.ntt_batch_device(src.data_ptr(), o1.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
        same = same and bool(torch.equal(o0[:n * batch], o1[:n * batch]))
    t = {2048: 1e9, 1024: 1e9}
    reps = 30 if n * batch <= (1 << 22) else 8
    for rnd in range(4):
        for v in (2048, 1024):
            b.tune("ntt_tile", v)
            b.ntt_batch_device(src.data_ptr(), o0.data_ptr(), n, batch, coset_shift=cs, sync=True)
            t0 = time.perf_counter()
            for _ in range(reps): b.ntt_batch_device(src.data_ptr(), o0.data_ptr(), n, batch, coset_shift=cs)
            b.sync(); t[v] = min(t[v], (time.perf_counter() - t0) / reps)
    print(f"n={n:9d} batch={batch:3d} coset={cs is not None!s:5s} equal={same}  tile2048 {t[2048]*1e6:9.1f} us  tile1024 {t[1024]*1e6:9.1f} us  ({(t[1024]/t[2048]-1)*100:+.1f} %)", flush=True)
b.tune("ntt_tile", 0)
