#!/usr/bin/env python3
"""Shared provers under churn (GPU box).

T threads prove in a loop through provers of ONE proof (uzk_prover_create(n, 1): the library runs callers that stand at the same
round together, uzkge_amd/csrc/coalesce.cpp) while everything that can happen around a shared round happens:
  * two circuits, four witnesses each -- a caller's next proof is any of them (callers over different circuits never share),
  * the main thread keeps swapping circuit A's twelve public-key tables between two sets: a proof must be, as a whole, the proof
    under set a or under set b,
  * callers that dawdle between two rounds for longer than the straggler wait (their lane is moved out and finishes alone),
  * proofs abandoned after a round (the next round 1 drops them),
  * provers destroyed and made anew, also in the middle of a proof,
  * one thread that proves pairs on a lockstep prover of its own context beside them.
Every finished proof is compared, commitment by commitment and evaluation by evaluation, with what a private prover made of the
same inputs single-threaded before the threads started.

usage: python tools/soak_shared_provers.py [seconds=40] [threads=12] [log_n=12]
"""
import os
import sys
import threading
import time

import numpy as np

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "oracle"), os.path.join(R, "tools")):
    sys.path.insert(0, p)
from uzkge_amd import backend as b   # noqa: E402

b.init(0)
import prover_chain as pch   # noqa: E402
import test_gpu_circuit_rounds as T   # noqa: E402
from test_gpu_msm import affine_of, rand_fr_wire   # noqa: E402


def digest(o, with_tables=True):
    d = [[affine_of(j) for j in o["cm1"]], [affine_of(j) for j in o["cm_z"]]]
    if with_tables:
        d += [[affine_of(j) for j in o["cm_t"]], [affine_of(j) for j in o["cm_q"]], o["evals"].tobytes()]
    return d


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    inp_a, inp_b = pch.ChainInputs(n, 91), pch.ChainInputs(n, 191)
    cir = {"A": T._circuit_of(b, inp_a, precompute=1), "B": T._circuit_of(b, inp_b, precompute=1)}
    lanes = {"A": T._round_inputs(inp_a, 4), "B": T._round_inputs(inp_b, 4)}
    set_a = [np.ascontiguousarray(inp_a.table_polys[pch.T_QPK + t]) for t in range(12)]
    set_b = [rand_fr_wire(n, 950 + t) for t in range(12)]
    want = {}
    p0 = b.Prover(n, 1, shared=False)
    for name, tables in (("b", set_b), ("a", set_a)):
        cir["A"].update_tables(b.CS_QPK, tables)
        for k, x in enumerate(lanes["A"]):
            want[("A", k, name)] = digest(T._run_rounds(b, cir["A"], p0, [x]))
    for k, x in enumerate(lanes["B"]):
        want[("B", k, "-")] = digest(T._run_rounds(b, cir["B"], p0, [x]))
    p0.destroy()
    b.coalesce_config(8, 0, 0, 0)                 # the defaults, statistics from zero

    stop = threading.Event()
    lock = threading.Lock()
    count = dict(good=0, a=0, b=0, abandoned=0, dawdled=0, remade=0, pairs=0)
    errors = []

    def rounds(pr, c, x, rng, st):
        """the five rounds with this thread's mischief between them; None: abandoned"""
        B, hiding = 1, list(pch.HIDE_W) + [pch.HIDE_WSEL] * 3
        flat = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        dawdle = rng.random() < 0.06
        quit_after = int(rng.integers(1, 5)) if rng.random() < 0.05 else 0
        o = {}

        def between(k):
            if dawdle and rng.random() < 0.5:
                st["dawdled"] += 1
                time.sleep(0.035)
            return quit_after == k
        o["cm1"] = pr.round1(cir[c], flat(x.w_evals).reshape(B, 5 * n, 4), flat(x.wsel_evals).reshape(B, 3 * n, 4), np.arange(8, dtype=np.uint32),
                             flat(x.pi_evals[:8]).reshape(B, 8, 4), hiding, flat(np.concatenate([x.blinds_w, x.blinds_wsel])))
        if between(1): return None
        o["cm_z"] = pr.round2(flat(x.beta), flat(x.gamma), flat(x.blinds_z))
        if between(2): return None
        o["cm_t"] = pr.round3(flat(x.alpha), flat(x.t_rands))
        if between(3): return None
        o["evals"] = pr.round4(flat(x.zeta), True)
        if between(4): return None
        o["cm_q"] = pr.round5(flat(x.r_scalars[:len(pch.r_plan(True))]), flat(x.alpha_open), flat(x.alpha_open2))
        return o

    def worker(t):
        rng = np.random.default_rng(7000 + t)
        st = dict(good=0, a=0, b=0, abandoned=0, dawdled=0, remade=0, pairs=0)
        pr = b.Prover(n, 1)
        try:
            while not stop.is_set():
                c = "A" if rng.random() < 0.75 else "B"
                k = int(rng.integers(0, 4))
                o = rounds(pr, c, lanes[c][k], rng, st)
                if o is None:
                    st["abandoned"] += 1
                    if rng.random() < 0.5:                     # ... and the prover goes away with the proof half done
                        pr.destroy(); pr = b.Prover(n, 1); st["remade"] += 1
                    continue
                got = digest(o)
                if c == "B":
                    ok = got == want[("B", k, "-")]
                else:
                    kind = "a" if got == want[("A", k, "a")] else "b" if got == want[("A", k, "b")] else None
                    ok = kind is not None
                    if ok: st[kind] += 1
                if not ok:
                    raise AssertionError(f"thread {t}: a proof of circuit {c}, witness {k} is neither of the expected proofs")
                st["good"] += 1
                if rng.random() < 0.02:
                    pr.destroy(); pr = b.Prover(n, 1); st["remade"] += 1
        except Exception as e:            # noqa: BLE001 -- surfaced by the main thread
            errors.append(e)
            stop.set()
        finally:
            pr.destroy()
            with lock:
                for key in st: count[key] += st[key]

    def pair_worker():
        """a lockstep prover of two proofs on its own context, beside the shared ones"""
        st = 0
        try:
            ctx = b.ctx_create(); b.ctx_set_current(ctx)
            pr = b.Prover(n, 2)
            try:
                while not stop.is_set():
                    o = T._run_rounds(b, cir["B"], pr, lanes["B"][:2])
                    for lane in range(2):
                        got = [[affine_of(j) for j in o["cm1"][8 * lane:8 * lane + 8]], [affine_of(o["cm_z"][lane])], [affine_of(j) for j in o["cm_t"][5 * lane:5 * lane + 5]],
                               [affine_of(j) for j in o["cm_q"][2 * lane:2 * lane + 2]], o["evals"][19 * lane:19 * lane + 19].tobytes()]
                        if got != want[("B", lane, "-")]: raise AssertionError(f"lockstep pair: lane {lane} differs")
                    st += 1
            finally:
                pr.destroy(); b.ctx_set_current(0); b.ctx_destroy(ctx)
        except Exception as e:            # noqa: BLE001
            errors.append(e)
            stop.set()
        with lock: count["pairs"] += st

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)] + [threading.Thread(target=pair_worker)]
    t0 = time.time()
    for th in threads: th.start()
    swaps = 0
    while time.time() - t0 < seconds and not stop.is_set():
        cir["A"].update_tables(b.CS_QPK, set_b if swaps % 2 == 0 else set_a)
        swaps += 1
        time.sleep(0.004)
        if swaps % 500 == 0: print(f"{time.time() - t0:6.1f} s  {swaps} table swaps", flush=True)
    stop.set()
    for th in threads: th.join()
    st = b.coalesce_stats()
    print({"seconds": round(time.time() - t0, 1), "threads": n_threads, "table_swaps": swaps, **count,
           "shared_rounds": st["rounds"], "calls_per_shared_round": round(st["calls"] / max(1, st["rounds"]), 2), "widest": st["widest"], "moved_out": st["moved_out"]})
    for c in cir.values(): c.release()
    if errors:
        print("FAILED:", errors[0])
        sys.exit(1)
    assert count["good"] > 50 and count["a"] > 0 and count["b"] > 0 and st["widest"] >= 2, "the run did not exercise what it is for"
    print("OK: every finished proof equals the single-threaded proof of its inputs")


if __name__ == "__main__":
    main()
