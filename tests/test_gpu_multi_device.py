"""Several devices behind the C ABI from ONE process (SURVEY.md 8b "uzk_init(n_devices)", 8e): contexts bound to
a device, an SRS cut into point chunks over a device list with the partial sums folded on the host, circuits and provers that live
on the device of the context that made them.  The test box has one GPU, so the chunks are VIRTUAL shards on device 0 -- the same
ordinal several times: every code path of the multi-device form runs (a context, a stream and a registry entry per chunk, one host
thread per chunk, the fold), only the second piece of silicon is missing."""
import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chunks,window_bits", [(2, -1), (3, 0), (8, -1)])
def test_sharded_msm_equals_the_single_msm_and_the_oracle(gpu, chunks, window_bits):
    b = gpu
    wire, _ = load_srs("lagrange-srs-4096.bin")
    n = 4096
    sh = b.ShardedSrs(wire, [0] * chunks, window_bits)
    one = b.Srs.from_host(wire)
    try:
        total, parts = sh.info()
        assert total == n and [p[0] for p in parts] == [0] * chunks
        assert [(lo, hi) for _, lo, hi in parts] == [(i * n // chunks, (i + 1) * n // chunks) for i in range(chunks)]     # uzkge_amd/sharded.py chunk_bounds
        for seed, count in ((1, n), (2, n - 1), (3, n // chunks + 1), (4, 1)):
            s = rand_fr_wire(count, seed)
            if seed == 2:
                s[::3] = 0                                           # zero scalars contribute the identity
            got, partials = sh.msm(s, want_partials=True)
            assert affine_of(got) == affine_of(b.msm(one, s)) == oc.jac_to_affine_ints(oc.msm_pippenger(wire, s, 0, 4)), (seed, count)
            # the partial sums are the chunks' own MSMs: folding them by hand gives the same point
            acc = None
            for i, (_, lo, hi) in enumerate(parts):
                lo, hi = min(lo, count), min(hi, count)
                want = oc.jac_to_affine_ints(oc.msm_pippenger(wire[lo:hi], s[lo:hi], 0, 4)) if hi > lo else None
                assert affine_of(partials[i]) == want, (seed, i)
                acc = opy.g1_add(acc, want)
            assert acc == affine_of(got)
        # the Lagrange identity of the reference's own file through the sharded path: sum_i L_i = G
        assert affine_of(sh.msm(oc.fr_from_ints([1] * n))) == opy.G1_GEN
        with pytest.raises(Exception) as e:
            sh.msm(rand_fr_wire(n + 1, 9))                          # commit longer than the SRS: DegreeError, as kzg_poly_commitment.rs:283-285
        assert "exceeds SRS length" in str(e.value)
    finally:
        sh.release(); one.release()


@pytest.mark.parametrize("log_n,chunks", [(20, 2), (21, 3), (20, 8)])
def test_sharded_chunks_that_stream_their_scalars_equal_the_single_msm(gpu, log_n, chunks):
    """Chunks of >= 2^19 points stream their host scalars in sub-chunks under their own accumulation (sharded.cpp; a single call
    starts streaming at 2^22): 2^20 / 2 and 2^21 / 3 chunks stream, 2^20 / 8 do not -- the same commitment as one MSM on one context,
    also for a vector shorter than the SRS (the last chunks run short or empty) and for the prover-like scalar mix."""
    import torch
    b = gpu
    n = 1 << log_n
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((2 * n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    b.synth_points_random(pts.data_ptr(), n, 41)
    b.synth_scalars(sc.data_ptr(), n, 42)
    b.synth_scalars_mix(sc.data_ptr() + n * 32, n, 43)
    hp = pts.cpu().numpy().view(np.uint64).reshape(-1, 8)
    hs = sc.cpu().numpy().view(np.uint64).reshape(-1, 4)
    one = b.Srs.from_device(pts.data_ptr(), n)
    sh = b.ShardedSrs(hp, [0] * chunks, -1)
    try:
        for lo, count in ((0, n), (n, n), (0, n - 12345), (0, n // chunks + 7)):
            s = hs[lo:lo + count]
            assert affine_of(sh.msm(s)) == affine_of(b.msm(one, s)), (lo, count)
    finally:
        sh.release(); one.release()


def test_a_context_on_a_named_device_carries_its_handles(gpu):
    """uzk_ctx_create_on(0): what is made under it lives on its device; a proof through a circuit and a prover made there equals
    the default context's."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import prover_chain as pch
    from test_gpu_circuit_rounds import _circuit_of, _run_rounds
    from test_gpu_coalesce import _digest
    b = gpu
    n = 1 << 12
    inp = pch.ChainInputs(n, 77)
    cir0 = _circuit_of(b, inp)
    p0 = b.Prover(n, 1, shared=False)
    want = _digest(_run_rounds(b, cir0, p0, [inp]))
    p0.destroy(); cir0.release()
    ctx = b.ctx_create_on(0)
    assert b.ctx_device(ctx) == 0 and b.ctx_device(0) == 0
    b.ctx_set_current(ctx)
    try:
        cir = _circuit_of(b, inp)
        assert cir.info()[3] == 0
        for shared in (False, True):
            p = b.Prover(n, 1, shared=shared)
            try:
                assert _digest(_run_rounds(b, cir, p, [inp])) == want
            finally:
                p.destroy()
        cir.release()
        with pytest.raises(Exception):
            b.ctx_create_on(b.device_count())                       # no such device
    finally:
        b.ctx_set_current(0)
        b.ctx_destroy(ctx)
