#!/usr/bin/env python3
"""Throughput of independent commit streams on ONE GPU: T threads, each with its own context (uzk_ctx_create), each
issuing the 16 commits of a proof (batches of 8 + 1 + 5 + 2 over n = 2^14) back to back.  One context = today's
single-stream behaviour; more contexts fill the chip that a single latency-bound proof leaves idle.
usage: python tools/concurrent_proofs.py [--threads 1,2,4] [--proofs 20]"""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uzkge_amd import backend as b

ap = argparse.ArgumentParser()
ap.add_argument("--threads", default="1,2,4")
ap.add_argument("--proofs", type=int, default=20)
a = ap.parse_args()
b.init(0)
n = 1 << 14
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
sc = torch.empty((16 * n, 4), dtype=torch.int64, device="cuda")
coef = torch.empty((10 * n, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_points_random(pts.data_ptr(), n, 7); b.synth_scalars(sc.data_ptr(), 16 * n, 8); b.synth_scalars(coef.data_ptr(), 10 * n, 9)
srs = b.Srs.from_device(pts.data_ptr(), n)
srs.precompute(0)
res = {}
for T in [int(x) for x in a.threads.split(",")]:
    outs = [torch.empty((10 * n, 4), dtype=torch.int64, device="cuda") for _ in range(T)]
    torch.cuda.synchronize()
    def proof(out):
        b.ntt_batch_device(coef.data_ptr(), out.data_ptr(), n, 10, inverse=True)
        for cnt, off in ((8, 0), (1, 8), (5, 9), (2, 14)):
            b.msm_batch_device(srs, sc.data_ptr() + off * n * 32, n, cnt)
    def worker(i, ready, go):
        h = b.ctx_create(); b.ctx_set_current(h)
        proof(outs[i]); b.sync()
        ready.release(); go.wait()
        for _ in range(a.proofs): proof(outs[i])
        b.sync()
        b.ctx_set_current(0); b.ctx_destroy(h)
    ready, go = threading.Semaphore(0), threading.Event()
    ths = [threading.Thread(target=worker, args=(i, ready, go)) for i in range(T)]
    for t in ths: t.start()
    for _ in range(T): ready.acquire()
    t0 = time.perf_counter(); go.set()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    res[f"{T}_contexts"] = {"proofs_per_s": round(T * a.proofs / dt, 1), "ms_per_proof_stream": round(dt / a.proofs * 1e3, 3)}
    print(T, res[f"{T}_contexts"], flush=True)
print(json.dumps({"what": "16 commits (8+1+5+2, n = 2^14, window table) + 10 iFFT per 'proof', T threads with their own contexts", **res}))
