#!/usr/bin/env python3
"""The hot-path call mix of ONE zshuffle 52-card proof (n = 2^14, quotient
domain 6n = 98304; 16 MSMs <= n, 17 radix-2 NTTs of n, 11 mixed-radix NTTs of 6n), on device-resident
data: one call per primitive (the order the Rust prover issues them) vs the batched entry points
(commits that do not depend on each other through Fiat-Shamir go out together:
8 wires/selectors, 1 z, 5 t-chunks, 2 openings)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b


def run(reps=5, precompute=False):
    b.init(0)
    n, m = 1 << 14, 98304
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((16 * n, 4), dtype=torch.int64, device="cuda")
    big = torch.empty((11 * m, 4), dtype=torch.int64, device="cuda")
    out = torch.empty((17 * n, 4), dtype=torch.int64, device="cuda")
    outb = torch.empty((11 * m, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    b.synth_points_random(pts.data_ptr(), n, 7); b.synth_scalars(sc.data_ptr(), 16 * n, 8); b.synth_scalars(big.data_ptr(), 11 * m, 9)
    srs = b.Srs.from_device(pts.data_ptr(), n)
    b.tune("msm_no_precompute", 0 if precompute else 1)
    if precompute: srs.precompute(0)
    res = {}
    def timed(name, fn):
        fn(); b.sync(); t = time.perf_counter()
        for _ in range(reps): fn()
        b.sync(); res[name] = (time.perf_counter() - t) / reps * 1e3
    def single():
        for k in range(17): b.ntt_device(sc.data_ptr() + (k % 16) * n * 32, out.data_ptr() + k * n * 32, n, inverse=(k < 10))
        for k in range(11): b.ntt_device(big.data_ptr() + k * m * 32, outb.data_ptr() + k * m * 32, m, inverse=(k == 10))
        for k in range(16): b.msm_device(srs, sc.data_ptr() + k * n * 32, n)
    def batched():
        b.ntt_batch_device(sc.data_ptr(), out.data_ptr(), n, 10, inverse=True)      # pi, 5 w, 3 sel, z
        b.ntt_batch_device(sc.data_ptr(), out.data_ptr(), n, 7)                      # 5 t-chunks + 2 openings
        b.ntt_batch_device(big.data_ptr(), outb.data_ptr(), m, 10)                   # t_poly coset FFTs
        b.ntt_device(big.data_ptr(), outb.data_ptr(), m, inverse=True)
        for cnt, off in ((8, 0), (1, 8), (5, 9), (2, 14)):
            b.msm_batch_device(srs, sc.data_ptr() + off * n * 32, n, cnt)
    # the fused quotient kernel over the 6n domain (56 input vectors; 46 are per-circuit tables)
    qv = torch.empty((56 * m, 4), dtype=torch.int64, device="cuda")
    b.synth_scalars(qv.data_ptr(), 56 * m, 10)
    sca = torch.empty((32, 4), dtype=torch.int64, device="cuda"); b.synth_scalars(sca.data_ptr(), 32, 11)
    sh = sca.cpu().numpy().view(np.uint64)
    ptrs = [qv.data_ptr() + i * m * 32 for i in range(56)]
    def quotient():
        b.t_quotient_device(n, 6, ptrs, sh[0], sh[1], sh[2], sh[3:8], sh[8], sh[9], sh[10], sh[11:17], outb.data_ptr(), sync=False)
    timed("t_quotient_ms", quotient)
    timed("single_calls_ms", single)
    timed("batched_calls_ms", batched)
    timed("msm_16_single_ms", lambda: [b.msm_device(srs, sc.data_ptr() + k * n * 32, n) for k in range(16)])
    timed("msm_16_in_4_batches_ms", lambda: [b.msm_batch_device(srs, sc.data_ptr() + off * n * 32, n, cnt) for cnt, off in ((8, 0), (1, 8), (5, 9), (2, 14))])
    srs.release()
    b.tune("msm_no_precompute", 0)
    return {k: round(v, 3) for k, v in res.items()}


if __name__ == "__main__":
    print(json.dumps({"general": run(), "precomputed_srs": run(precompute=True)}))
