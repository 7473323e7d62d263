"""A/B of the quotient kernel at the prover's size: one lane per point (0), four term groups on four waves per 64 points at two
(1) or three (3) waves per SIMD.  Outputs must agree."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
from uzkge_amd import backend as b
from prover_chain import ChainInputs, ProverChain
c = ProverChain(inputs=ChainInputs(1 << 14, 11))
c.run(); b.sync()
def tq():
    b.t_quotient_device(c.n, 6, c.tq_ptrs, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv, c.d_tq.ptr, sync=False)
ref = None
for split in (0, 1, 3, 0, 1, 3):
    b.tune("tq_split", split)
    tq(); b.sync()
    out = c.d_tq.host()
    if ref is None: ref = out
    assert np.array_equal(out, ref)
    t = time.perf_counter()
    for _ in range(50): tq()
    b.sync()
    print(f"tq_split={split}: {(time.perf_counter() - t) / 50 * 1e6:.1f} us", flush=True)
