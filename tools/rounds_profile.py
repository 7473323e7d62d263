#!/usr/bin/env python3
"""Where one proof's time goes through uzk_prove_round1..5: per-kernel device time (hipEvent brackets of the library) for a
lockstep batch of B proofs, per proof, next to the wall time of the same chain without the brackets.
usage: python tools/rounds_profile.py [--batch 1,4,8] [--log-n 14] [--reps 10]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np

from uzkge_amd import backend as b
import prover_chain as pch


def run(B, inp, cir, reps):
    n = inp.n
    pr = b.Prover(n, B, shared=False)        # profiled on the calling context
    rep = lambda a: np.concatenate([np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)] * B)
    hiding = list(pch.HIDE_W) + [pch.HIDE_WSEL] * 3
    w, s = rep(inp.w_evals).reshape(B, 5 * n, 4), rep(inp.wsel_evals).reshape(B, 3 * n, 4)
    bl = rep(np.concatenate([inp.blinds_w, inp.blinds_wsel]))
    pi_idx, pi_val = np.arange(8, dtype=np.uint32), rep(inp.pi_evals[:8]).reshape(B, 8, 4)

    def chain():
        pr.round1(cir, w, s, pi_idx, pi_val, hiding, bl)
        pr.round2(rep(inp.beta), rep(inp.gamma), rep(inp.blinds_z))
        pr.round3(rep(inp.alpha), rep(inp.t_rands))
        pr.round4(rep(inp.zeta))
        pr.round5(rep(inp.r_scalars), rep(inp.alpha_open), rep(inp.alpha_open2))
    for _ in range(3):
        chain()
    b.sync()
    t = time.perf_counter()
    for _ in range(reps):
        chain()
    b.sync()
    wall = (time.perf_counter() - t) / reps * 1e3
    b.profile_reset(); b.profile_enable(True)
    for _ in range(3):
        chain()
    b.sync(); b.profile_enable(False)
    tab = b.profile_table()
    pr.destroy()
    kern = {k: round(ms / 3 / B, 4) for k, (cnt, ms) in tab.items() if not k.startswith("host_")}
    host = {k: round(ms / 3 / B, 4) for k, (cnt, ms) in tab.items() if k.startswith("host_")}
    return {"batch": B, "wall_ms_per_proof": round(wall / B, 4), "device_kernel_ms_per_proof": round(sum(kern.values()), 4),
            "kernels_ms_per_proof": dict(sorted(kern.items(), key=lambda kv: -kv[1])), "host_sections_ms_per_proof": host}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default="1,4,8")
    ap.add_argument("--log-n", type=int, default=14)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    b.init(0)
    inp = pch.ChainInputs(1 << a.log_n, 11)
    cir = b.Circuit(inp.n, inp.lagrange_wire, inp.bases[inp.n:], inp.perm, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a,
                    [inp.table_polys[i] for i in range(pch.N_TABLES)], precompute=True, synthetic=True)
    for B in [int(x) for x in a.batch.split(",")]:
        print(json.dumps(run(B, inp, cir, a.reps)))
    cir.release()
