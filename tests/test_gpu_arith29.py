"""The prover's polynomial kernels run on the lazy 29-bit-limb arithmetic (uzkge_amd/csrc/lz29.hpp; round 6) -- loaded values enter the
2^261-form by re-limbing alone, bounds are carried in the types -- with the 8 x 32-bit Montgomery kernels kept behind
uzk_tune("arith29", mask).  Exact field arithmetic either way: both forms must produce the SAME BYTES, and those must be the CPU
oracle's (helpers.rs:284-656 restated term by term in oracle/bn254_oracle.c), on random vectors and on vectors made of the values
where lazy limbs and lazy reductions are at their bounds: 0, 1, r - 1, r - 2, (r - 1) / 2, 2^29 k - 1, all-ones limbs."""
import numpy as np
import pytest
import torch

import bn254_py as opy
import oracle_c as oc
from util import affine_of, rand_fr_wire

pytestmark = pytest.mark.gpu


def _edge_pool():
    r = opy.R
    vals = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, (1 << 253) - 1, (1 << 252), (1 << 29) - 1, (1 << 58) - 1, (1 << 232) - 1,
            int("1" * 253, 2) % r, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, r - (1 << 29), r - (1 << 232), 5, pow(5, -1, r)]
    # as WIRE values (Montgomery words): what the kernels load; to_mont of small numbers and raw words near r both matter
    wire = np.asarray(oc.fr_from_ints(vals), dtype=np.uint64).reshape(-1, 4)
    raw = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        raw[i] = opy.int_to_limbs(v % r)
    return np.concatenate([wire, raw])


def _vectors(m, seed, edge):
    v = rand_fr_wire(56 * m, seed).reshape(56, m, 4)
    if edge:
        pool = _edge_pool()
        rng = np.random.default_rng(seed)
        pick = rng.integers(0, pool.shape[0], size=(56, m))
        mask = rng.random((56, m)) < 0.7
        v[mask] = pool[pick[mask]]
    return v


@pytest.fixture
def both(gpu):
    yield gpu
    gpu.tune("arith29", 7)


@pytest.mark.parametrize("n,edge,shuffle", [(64, False, True), (64, True, True), (1024, True, True), (1024, True, False), (1 << 14, False, True),
                                            (1 << 17, True, True), (1 << 17, False, False)])
def test_quotient_kernels_agree_and_match_the_oracle(both, n, edge, shuffle):
    """m = 6 n points: up to 2^19 the term groups run on the waves of a workgroup (t_quotient_split*), above one lane per point."""
    gpu = both
    factor = 6
    m = n * factor
    vecs = _vectors(m, 77 + n, edge)
    sc = rand_fr_wire(6 + 5 + factor, 5 + n)
    if edge:
        pool = _edge_pool()
        sc[1] = pool[3]; sc[2] = pool[0]; sc[6] = pool[1]                    # beta = r - 1, gamma = 0, k_0 = 1
    alpha, beta, gamma, g, g_inv, ea = sc[:6]
    k, zhi = sc[6:11], sc[11:]
    if not shuffle:
        for slot in list(range(5, 8)) + list(range(31, 56)):
            vecs[slot] = 0
    want = oc.t_quotient(n, factor, vecs, alpha, beta, gamma, k, g, g_inv, ea, zhi)
    d = torch.from_numpy(vecs.view(np.int64)).cuda()
    out = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    ptrs = [d[i].data_ptr() for i in range(56)]
    if not shuffle:
        for slot in list(range(5, 8)) + list(range(31, 56)):
            ptrs[slot] = 0
    got = {}
    for mask in (7, 0):
        gpu.tune("arith29", mask)
        out.zero_()
        torch.cuda.synchronize()
        gpu.t_quotient_device(n, factor, ptrs, alpha, beta, gamma, k, g, g_inv, ea, zhi, out.data_ptr())
        got[mask] = out.cpu().numpy().view(np.uint64).reshape(m, 4)
    assert np.array_equal(got[7], got[0]), "29-bit lazy kernel and 8 x 32-bit kernel differ"
    assert np.array_equal(got[7], want), "quotient differs from the oracle"


@pytest.mark.parametrize("log_n,batch,mix", [(14, 8, False), (14, 3, True), (12, 5, True), (16, 2, False)])
def test_msm_bucket_side_kernels_agree(both, log_n, batch, mix):
    """The bucket-side additions (class sums of the general pipeline, the quad folds of the small one) on the lazy 29-bit limbs against
    the 8 x 32-bit words: the same commitments, with and without the window table, uniform and skewed scalars (skew makes deep
    fold levels and all-equal buckets: the doubling and infinity branches)."""
    gpu = both
    n = 1 << log_n
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((batch * n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 31)
    (gpu.synth_scalars_mix if mix else gpu.synth_scalars)(sc.data_ptr(), batch * n, 32)
    if mix:
        sc[n // 2: n // 2 + 64] = sc[0:1].expand(64, 4)          # one value 64 times in a row: equal buckets meet in the folds
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        res = {}
        for table in (0, 1):
            if table:
                srs.precompute(0)
            for small in (1, 0):
                gpu.tune("msm_small", small)
                for mask in (7, 0):
                    gpu.tune("arith29", mask)
                    res[(table, small, mask)] = [affine_of(x) for x in gpu.msm_batch_device(srs, sc.data_ptr(), n, batch)]
        ref = res[(0, 1, 0)]
        for key, val in res.items():
            assert val == ref, key
        if log_n <= 12:      # anchor on the oracle
            hp = pts.cpu().numpy().view(np.uint64); hs = sc[:n].cpu().numpy().view(np.uint64)
            assert ref[0] == oc.jac_to_affine_ints(oc.msm_pippenger(hp, hs, 0, 8))
    finally:
        gpu.tune("msm_small", 1)
        srs.release()


@pytest.mark.parametrize("lanes", [1, 3])
def test_whole_proofs_agree_between_the_arithmetics(both, lanes):
    """Five rounds of `lanes` proofs in lockstep (commitments, the 19 evaluations, both openings): every mask of uzk_tune("arith29")
    gives the same proof -- the quotient (bit 0), the lane evaluations and linear combinations (bit 1: 19 evaluations, r(X) over 43
    polynomials, the opening's combinations), the commits' bucket-side additions (bit 2)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import prover_chain as pch
    from test_gpu_circuit_rounds import _circuit_of, _round_inputs, _run_rounds
    gpu = both
    n = 1 << 12
    inp = pch.ChainInputs(n, 77)
    xs = _round_inputs(inp, lanes)
    cir = _circuit_of(gpu, inp, precompute=1)
    p = gpu.Prover(n, lanes, shared=False)
    try:
        ref = None
        for mask in (0, 7, 1, 2, 4):
            gpu.tune("arith29", mask)
            o = _run_rounds(gpu, cir, p, xs)
            d = tuple(tuple(affine_of(j) for j in o[k]) for k in ("cm1", "cm_z", "cm_t", "cm_q")) + (o["evals"].tobytes(),)
            if ref is None:
                ref = d
            assert d == ref, mask
    finally:
        p.destroy(); cir.release()
