"""Full-size checks through size-independent properties (no oracle pass over 2^24 points):
with P_i = (i+1) Q the MSM has the closed form (sum_i s_i (i+1)) Q."""
import numpy as np
import pytest
import torch

import bn254_py as opy
import oracle_c as oc
from util import affine_of, weighted_index_sum

pytestmark = pytest.mark.gpu


def _wire(t):
    return t.cpu().numpy().view(np.uint64).reshape(-1, 4)


@pytest.mark.parametrize("log_n", [10, 16, 20, 22, 24])   # 22: packed sort entries; 24: BASELINE's headline size
def test_arith_points_closed_form(gpu, log_n):
    n = 1 << log_n
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    seed = oc.fr_from_ints([0x755A6B67655F6D73])[0]
    gpu.synth_points_arith(pts.data_ptr(), n, seed)
    gpu.synth_scalars(sc.data_ptr(), n, 42)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        got = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
    finally:
        srs.release()
    k = weighted_index_sum(_wire(sc))
    q = opy.g1_mul(opy.G1_GEN, 0x755A6B67655F6D73)
    assert got == opy.g1_mul(q, k)
    # the generated points themselves: first few equal (i+1) Q and lie on the curve
    head = pts[:4].cpu().numpy().view(np.uint64)
    for i in range(4):
        assert opy.wire_to_affine(head[i].tobytes()) == opy.g1_mul(q, i + 1)


def test_random_points_on_curve_and_split_property(gpu):
    """hash-to-curve points are valid; MSM(all) == MSM(first half) + MSM(second half); and the
    result equals the CPU oracle at 2^16."""
    n = 1 << 16
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 1234)
    gpu.synth_scalars(sc.data_ptr(), n, 99)
    hp = pts.cpu().numpy().view(np.uint64).reshape(-1, 8)
    hs = sc.cpu().numpy().view(np.uint64).reshape(-1, 4)
    for i in (0, 1, 1000, n - 1):
        assert oc.lib.oracle_g1_is_on_curve(oc._p(np.ascontiguousarray(hp[i]))) == 1
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        full = gpu.msm_device(srs, sc.data_ptr(), n)
        h1 = gpu.msm_device(srs, sc.data_ptr(), n // 2)
        h2 = gpu.msm_device(srs, sc.data_ptr() + (n // 2) * 32, n // 2, offset=n // 2)
    finally:
        srs.release()
    assert affine_of(gpu.g1_fold(np.stack([h1, h2]))) == affine_of(full)
    assert affine_of(full) == affine_of(oc.msm_pippenger(hp, hs, 0, 16))


@pytest.mark.parametrize("kind", ["all_ones", "all_minus_one", "two_values", "prover_mix"])
def test_skewed_scalars_closed_form(gpu, kind):
    """Heavily skewed scalar sets at 2^16 (one bucket holds every point, so the multi-level
    partial-sum folding runs) checked with the arithmetic-progression closed form."""
    n = 1 << 16
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    seed_int = 0x1234567
    gpu.synth_points_arith(pts.data_ptr(), n, oc.fr_from_ints([seed_int])[0])
    rng = np.random.default_rng(5)
    if kind == "all_ones":
        ints = [1] * n
    elif kind == "all_minus_one":
        ints = [opy.R - 1] * n
    elif kind == "two_values":
        ints = [3 if v else (opy.R - 7) for v in rng.integers(0, 2, n)]
    else:
        cls = rng.integers(0, 10, n)
        u = rng.integers(1, 1 << 62, n)
        ints = [0 if c < 5 else 1 if c < 7 else opy.R - 1 if c < 8 else int(u[i]) & 0xFFFF if c < 9 else int(u[i]) ** 4 % opy.R
                for i, c in enumerate(cls)]
    wire = oc.fr_from_ints(ints)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        got = affine_of(gpu.msm(srs, wire))
    finally:
        srs.release()
    k = sum(s * (i + 1) for i, s in enumerate(ints)) % opy.R
    assert got == opy.g1_mul(opy.g1_mul(opy.G1_GEN, seed_int), k)


@pytest.mark.parametrize("log_n,c", [(16, 0), (20, 0), (20, 22)])
def test_precomputed_closed_form(gpu, log_n, c):
    """Window-table mode at size: MSM(P_i = (i+1)Q) == (sum s_i (i+1)) Q."""
    n = 1 << log_n
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    seed_int = 0xABCDEF12345
    gpu.synth_points_arith(pts.data_ptr(), n, oc.fr_from_ints([seed_int])[0])
    gpu.synth_scalars(sc.data_ptr(), n, 4242)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        srs.precompute(c)
        got = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
    finally:
        srs.release()
    k = weighted_index_sum(_wire(sc))
    assert got == opy.g1_mul(opy.g1_mul(opy.G1_GEN, seed_int), k)


def test_maximum_size_chunked_closed_form(gpu):
    """Beyond 2^26 points one sort pass would overflow its 31-bit index: the ABI cuts the input into
    point chunks and folds the partial sums.  n = 2^26 + 4097 (BASELINE.json's largest configuration is
    2^26), checked with the arithmetic-progression closed form."""
    n = (1 << 26) + 4097
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    seed_int = 0x5EED5EED
    gpu.synth_points_arith(pts.data_ptr(), n, oc.fr_from_ints([seed_int])[0])
    gpu.synth_scalars(sc.data_ptr(), n, 2626)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        got = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
    finally:
        srs.release()
    k = weighted_index_sum(_wire(sc))
    del pts, sc
    torch.cuda.empty_cache()
    assert got == opy.g1_mul(opy.g1_mul(opy.G1_GEN, seed_int), k)


def test_random_2p20_points_vs_oracle(gpu):
    """BASELINE config #2 literally: 2^20 random points and uniform scalars, the GPU result against the CPU
    oracle's Pippenger over the same inputs."""
    n = 1 << 20
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 0x755A6B67655F6D73)
    gpu.synth_scalars(sc.data_ptr(), n, 0x5CA1AB1E)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        got = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
    finally:
        srs.release()
    hp = pts.cpu().numpy().view(np.uint64).reshape(-1, 8)
    hs = sc.cpu().numpy().view(np.uint64).reshape(-1, 4)
    assert got is not None and got == affine_of(oc.msm_pippenger(hp, hs, 0, 16))


def test_headline_paths_2p23_variants_agree(gpu):
    """The code paths only the large configurations reach -- digits + first histogram in one kernel (needs >= 256
    sort chunks, i.e. n >= 2^23), the 512-lane scatter of sorts with >= 2^26 entries -- against the generic last sort pass
    and, through the split property, against smaller sizes the oracle checks."""
    n = 1 << 23
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 77)
    gpu.synth_scalars(sc.data_ptr(), n, 78)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        ref = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
        gpu.tune("msm_seg_sort", 0)
        try:
            assert affine_of(gpu.msm_device(srs, sc.data_ptr(), n)) == ref
        finally:
            gpu.tune("msm_seg_sort", 1)
        # split property ties the value to smaller, oracle-checked sizes
        h1 = gpu.msm_device(srs, sc.data_ptr(), n // 2)
        h2 = gpu.msm_device(srs, sc.data_ptr() + (n // 2) * 32, n // 2, offset=n // 2)
        assert affine_of(gpu.g1_fold(np.stack([h1, h2]))) == ref
    finally:
        srs.release()


def test_point_chunk_loop_small(gpu):
    """The > 2^26-point chunk loop of the dispatcher (api.cpp msm_dispatch) with the chunk size lowered to 2^10:
    3000 points run as three chunks whose partial sums are folded; same commitment as the single pass and the oracle."""
    from util import load_srs, rand_fr_wire
    wire, _ = load_srs("lagrange-srs-4096.bin")
    s = rand_fr_wire(3000, 31)
    srs = gpu.Srs.from_host(wire)
    try:
        one = affine_of(gpu.msm(srs, s))
        gpu.tune("msm_chunk_log", 10)
        try:
            chunked = affine_of(gpu.msm(srs, s))
        finally:
            gpu.tune("msm_chunk_log", 26)
    finally:
        srs.release()
    assert chunked == one == affine_of(oc.msm_pippenger(wire[:3000], s, 0, 2))


@pytest.mark.parametrize("cfg", [1, 10, 13, 15, 0])
def test_segment_sort_skewed_segments(gpu, cfg):
    """The one-workgroup-per-segment last sort pass (msm_radix_segment_kernel) next to the generic kernels it leaves the
    long segments to: 2^20 points whose scalars are uniform except for runs that put (a) 40000 entries into one bucket
    (longer than any instantiation holds: the generic path), (b) 30000 into one bucket (fits the registers of the largest
    instantiation but not its LDS buffer: written directly), (c) 2 x 12500 into two buckets of one segment (two LDS
    rounds), (d) 3000 into one bucket (beyond the small instantiations).  Every instantiation forced in turn (10 + k),
    the automatic choice (1) and the generic path alone (0) give the closed-form result.  The same runs reach the first
    pass's one-workgroup-per-chunk kernel (msm_radix_chunk_kernel): run (a) fills one bin of chunk 0 beyond the LDS buffer
    (direct writes), the uniform rest goes out in rounds of whole bins.  The planted buckets hold hundreds of partial sums, so
    the extra fold levels run, with one wave per long fold (msm_fold_big_kernel)."""
    n = 1 << 20
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    seed_int = 0x5E65047
    gpu.synth_points_arith(pts.data_ptr(), n, oc.fr_from_ints([seed_int])[0])
    gpu.synth_scalars(sc.data_ptr(), n, 777)
    host = sc.cpu().numpy().view(np.uint64).reshape(-1, 4).copy()
    vals = oc.fr_from_ints([0x1234, 0x7ABC, 0x7ABD + 0x100, 0x7ABD + 0x101, opy.R - 0x2222])
    at = 1000
    for v, cnt in ((0, 40000), (1, 30000), (2, 12500), (3, 12500), (4, 3000)):
        host[at:at + cnt] = vals[v]
        at += cnt + 17
    sc.copy_(torch.from_numpy(host.view(np.int64)).reshape(n, 4))
    torch.cuda.synchronize()
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    gpu.tune("msm_seg_sort", cfg)
    try:
        got = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
    finally:
        gpu.tune("msm_seg_sort", 1)
        srs.release()
    k = weighted_index_sum(host)
    assert got == opy.g1_mul(opy.g1_mul(opy.G1_GEN, seed_int), k)


@pytest.mark.parametrize("log_n,batch", [(17, 3), (19, 2)])
def test_batched_general_pipeline_closed_form(gpu, log_n, batch):
    """uzk_msm_g1_batch_device above the small pipeline's sizes: `batch` vectors through one launch sequence of the general
    pipeline (batch * W sort segments per pass, 2 * batch * W logical windows in the class-sum reduction); every vector against
    the closed form and against its own single call."""
    n = 1 << log_n
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((batch * n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    seed_int = 0xBA7C4ED
    gpu.synth_points_arith(pts.data_ptr(), n, oc.fr_from_ints([seed_int])[0])
    for k in range(batch):
        if k == 1:
            gpu.synth_scalars_mix(sc.data_ptr() + k * n * 32, n, 50 + k)
        else:
            gpu.synth_scalars(sc.data_ptr() + k * n * 32, n, 50 + k)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        outs = [affine_of(j) for j in gpu.msm_batch_device(srs, sc.data_ptr(), n, batch)]
        singles = [affine_of(gpu.msm_device(srs, sc.data_ptr() + k * n * 32, n)) for k in range(batch)]
    finally:
        srs.release()
    q = opy.g1_mul(opy.G1_GEN, seed_int)
    host = _wire(sc).reshape(batch, n, 4)
    for k in range(batch):
        assert outs[k] == singles[k] == opy.g1_mul(q, weighted_index_sum(host[k])), k
