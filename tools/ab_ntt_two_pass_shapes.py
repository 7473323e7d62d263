#!/usr/bin/env python3
"""The two-pass split of 2^17 .. 2^21 on batched and fused shapes -- what a prover of a larger circuit issues (n = 2^16..2^19:
batches of n-point transforms, coset transforms over 6n = 3 * 2^(k+1)).  Interleaved A/B of uzk_tune("ntt_two_pass", v) (1 = two
passes of 9 .. 11 bits, the default; 0 = three passes of 5 .. 8 bits); outputs must be identical.
usage: python tools/ab_ntt_two_pass_shapes.py 1 0
(profiles/r05_ab_ntt_two_pass_shapes.txt was taken on the experiment build, whose values were 0 = three passes, 2 = two passes with the
XCD-neighbour workgroup mapping that shipped.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b
vals = [int(v) for v in sys.argv[1:]] or [1, 0]
b.init(0)
cap = 10 * 3 * (1 << 19)
src = torch.empty((cap, 4), dtype=torch.int64, device="cuda"); o0 = torch.empty((cap, 4), dtype=torch.int64, device="cuda"); o1 = torch.empty((cap, 4), dtype=torch.int64, device="cuda")
b.synth_scalars(src.data_ptr(), cap, 5)
kk = torch.empty((1, 4), dtype=torch.int64, device="cuda"); b.synth_scalars(kk.data_ptr(), 1, 77); k = kk.cpu().numpy().view(np.uint64).reshape(4)
cases = []
for lg in (17, 18, 19, 20, 21):
    cases.append((f"fft 2^{lg} x8", 8 << lg, lambda o, lg=lg: b.ntt_batch_device(src.data_ptr(), o, 1 << lg, 8)))
for lg in (17, 18, 19):
    m = 3 << lg
    cases.append((f"coset fft 3*2^{lg} x10", 10 * m, lambda o, m=m: b.ntt_batch_device(src.data_ptr(), o, m, 10, coset_shift=k)))
    cases.append((f"coset ifft 3*2^{lg} x1", m, lambda o, m=m: b.ntt_device(src.data_ptr(), o, m, inverse=True, coset_shift=k)))
for name, count, fn in cases:
    if count > cap: continue
    b.tune("ntt_two_pass", vals[0]); fn(o0.data_ptr()); b.sync()
    for v in vals[1:]:
        b.tune("ntt_two_pass", v); fn(o1.data_ptr()); b.sync()
        assert bool(torch.equal(o0[:count], o1[:count])), f"{name}: ntt_two_pass={v} changes the transform"
    t = {v: 1e9 for v in vals}
    for rnd in range(5):
        for v in vals:
            b.tune("ntt_two_pass", v); fn(o1.data_ptr()); b.sync()
            t0 = time.perf_counter()
            for _ in range(20): fn(o1.data_ptr())
            b.sync(); t[v] = min(t[v], (time.perf_counter() - t0) / 20)
    print(f"{name:26s} " + "  ".join(f"two_pass={v}: {t[v] * 1e6:8.1f} us" for v in vals), flush=True)
b.tune("ntt_two_pass", 1)
