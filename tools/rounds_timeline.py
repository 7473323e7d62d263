#!/usr/bin/env python3
"""Per-round wall time and per-kernel device time of one proof through uzk_prove_round1..5 (batch 1): where the 2 ms go.
usage: python tools/rounds_timeline.py [--log-n 14]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
from uzkge_amd import backend as b
import prover_chain as pch

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=14); ap.add_argument("--precompute", type=int, default=1)
a = ap.parse_args()
b.init(0)
inp = pch.ChainInputs(1 << a.log_n, 11)
n = inp.n
cir = b.Circuit(n, inp.lagrange_wire, inp.bases[n:], inp.perm, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a,
                [inp.table_polys[i] for i in range(pch.N_TABLES)], precompute=a.precompute, synthetic=True)
pr = b.Prover(n, 1, shared=False)          # profiled on the calling context
hiding = list(pch.HIDE_W) + [pch.HIDE_WSEL] * 3
w, s = inp.w_evals.reshape(1, 5 * n, 4), inp.wsel_evals.reshape(1, 3 * n, 4)
bl = np.concatenate([inp.blinds_w, inp.blinds_wsel])
steps = [("round1", lambda: pr.round1(cir, w, s, np.arange(8, dtype=np.uint32), inp.pi_evals[:8].reshape(1, 8, 4), hiding, bl)),
         ("round2", lambda: pr.round2(inp.beta, inp.gamma, inp.blinds_z)),
         ("round3", lambda: pr.round3(inp.alpha, inp.t_rands)),
         ("round4", lambda: pr.round4(inp.zeta)),
         ("round5", lambda: pr.round5(inp.r_scalars, inp.alpha_open, inp.alpha_open2))]
for _ in range(5):
    for _, f in steps: f()
wall = {k: 0.0 for k, _ in steps}
reps = 20
for _ in range(reps):
    for k, f in steps:
        t = time.perf_counter(); f(); wall[k] += time.perf_counter() - t
kern = {}
for k, f in steps:
    # run the earlier rounds un-profiled, then this one with the brackets
    pass
for target, _ in steps:
    for k, f in steps:
        if k == target:
            b.profile_reset(); b.profile_enable(True); f(); b.sync(); b.profile_enable(False)
            tab = b.profile_table()
            kern[k] = {name: round(ms, 4) for name, (cnt, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]) if not name.startswith("host_")}
        else:
            f()
for k, _ in steps:
    print(json.dumps({"round": k, "wall_ms": round(wall[k] / reps * 1e3, 4), "kernel_ms": round(sum(kern[k].values()), 4), "kernels": kern[k]}))
pr.destroy(); cir.release()
