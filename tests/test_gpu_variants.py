"""The pipelines the library chooses between by size give the same commitment / the same bytes: window widths on either side of
the automatic choice (the scan, quad and class-sum bucket reductions, one / two / three sort passes), the segment sort and the
generic last sort pass, the small-problem pipeline against the general one, both NTT tile sizes -- reached at sizes the CPU
oracle can check through the switches uzk_tune keeps for exactly that (include/uzkge_gpu.h)."""
import numpy as np
import pytest
import torch

import oracle_c as oc
from util import affine_of

pytestmark = pytest.mark.gpu

KNOBS = [
    {}, {"window_bits": 9}, {"window_bits": 12}, {"window_bits": 13}, {"window_bits": 14}, {"window_bits": 16}, {"window_bits": 17},
    {"window_bits": 18}, {"window_bits": 19}, {"msm_seg_sort": 0}, {"msm_seg_sort": 0, "window_bits": 13}, {"msm_seg_sort": 15, "window_bits": 15},
    {"msm_small": 0}, {"msm_small": 0, "window_bits": 9},
]
DEFAULTS = {"msm_small": 1, "window_bits": 0, "msm_seg_sort": 1}


def _apply(gpu, cfg):
    full = dict(DEFAULTS); full.update(cfg)
    gpu.set_msm_window_bits(full.pop("window_bits"))
    for k, v in full.items():
        gpu.tune(k, v)


@pytest.mark.parametrize("log_n", [10, 16, 19])
def test_msm_variants_agree(gpu, log_n):
    n = 1 << log_n
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 5)
    gpu.synth_scalars(sc.data_ptr(), n, 6)
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        ref = None
        for cfg in KNOBS:
            _apply(gpu, cfg)
            got = affine_of(gpu.msm_device(srs, sc.data_ptr(), n))
            if ref is None:
                ref = got
                if log_n <= 16:     # anchor the reference itself on the oracle
                    hp = pts.cpu().numpy().view(np.uint64); hs = sc.cpu().numpy().view(np.uint64)
                    assert ref == oc.jac_to_affine_ints(oc.msm_pippenger(hp, hs, 0, 8))
            assert got == ref, cfg
    finally:
        _apply(gpu, {})
        srs.release()


@pytest.mark.parametrize("n", [4096, 1 << 15, 3 << 13, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21])
def test_ntt_tiles_and_splits_agree_and_match_the_oracle(gpu, n):
    """Both workgroup tile sizes of the pass kernels (1024 / 2048 elements; 0 = chosen by size), out of place and in place (the
    limb-plane buffers between the passes), forward and inverse: the same bytes, and they are the oracle's.  2^17 .. 2^21 run as
    two passes of 9 .. 11 bits (9 + 8, 9 + 9, 10 + 9, 10 + 10, 11 + 10: one- and two-column tiles with the XCD-neighbour workgroup
    mapping) and, with uzk_tune("ntt_two_pass", 0), as the three passes of 5 .. 8 bits they had: both forms at every such size."""
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), n, 9)
    hx = x.cpu().numpy().view(np.uint64).reshape(-1, 4)
    try:
        for inv in (False, True):
            want = torch.from_numpy(oc.ntt(hx, inverse=inv, threads=4).view(np.int64)).reshape(n, 4).cuda()
            for two in ((1, 0) if (1 << 17) <= n <= (1 << 21) else (1,)):
                gpu.tune("ntt_two_pass", two)
                for tile in (0, 1024, 2048):
                    gpu.tune("ntt_tile", tile)
                    gpu.ntt_device(x.data_ptr(), a.data_ptr(), n, inverse=inv, sync=True)
                    assert torch.equal(a, want), (inv, tile, two)
                    y = x.clone()
                    torch.cuda.synchronize()
                    gpu.ntt_device(y.data_ptr(), y.data_ptr(), n, inverse=inv, sync=True)
                    assert torch.equal(y, want), (inv, tile, two, "in place")
    finally:
        gpu.tune("ntt_tile", 0)
        gpu.tune("ntt_two_pass", 1)


@pytest.mark.parametrize("n", [1, 2, 33, 1000, 4096, 16384, 32768])
def test_small_pipeline_agrees_with_general(gpu, n):
    """n <= 2^15 takes the one-workgroup-per-slot pipeline (msm_small_*): same commitments as the general
    pipeline for uniform, prover-mix, all-equal and all-zero scalar vectors, single and batched, at every
    window size the small path accepts, with and without a window table."""
    B = 3
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    sc = torch.empty((4 * B * n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_points_random(pts.data_ptr(), n, 11)
    gpu.synth_scalars(sc.data_ptr(), B * n, 12)
    gpu.synth_scalars_mix(sc.data_ptr() + B * n * 32, B * n, 13)
    sc[2 * B * n:3 * B * n] = sc[0:1].expand(B * n, 4)         # one value everywhere: a single bucket per window
    sc[3 * B * n:] = 0
    srs = gpu.Srs.from_device(pts.data_ptr(), n)
    try:
        def run():
            out = []
            for k in range(4):
                base = sc.data_ptr() + k * B * n * 32
                out.append(affine_of(gpu.msm_device(srs, base, n)))
                out += [affine_of(x) for x in gpu.msm_batch_device(srs, base, n, B)]
            return out
        gpu.tune("msm_small", 0)
        want = run()
        if n <= 4096:      # anchor the general path on the oracle
            hp = pts.cpu().numpy().view(np.uint64); hs = sc[:n].cpu().numpy().view(np.uint64)
            assert want[0] == oc.jac_to_affine_ints(oc.msm_pippenger(hp, hs, 0, 8))
        assert want[-1] is None and want[-2] is None               # all-zero scalars commit to infinity
        gpu.tune("msm_small", 1)
        for cfg in ({}, {"window_bits": 5}, {"window_bits": 9}, {"window_bits": 10}):
            _apply(gpu, cfg)
            gpu.tune("msm_small", 1)
            assert run() == want, cfg
        _apply(gpu, {})
        # window table (uzk_srs_precompute) on the small pipeline: per-window rows, window sums added without doublings
        gpu.tune("msm_no_precompute", 0)
        for c in (0, 6, 9):
            srs.precompute(c)
            assert run() == want, ("precompute", c)
    finally:
        _apply(gpu, {})
        gpu.tune("msm_small", 1)
        srs.release()


@pytest.mark.parametrize("n", [4096, 1 << 14, 3 << 12, 3 << 13, 98304, 1 << 17, 3 << 16, 3 << 17])
def test_ntt_fused_stages_match_the_oracle(gpu, n):
    """Coset scaling and the radix-3 stage of 3 * 2^k domains run inside the first / last Stockham pass (sub-transforms of at
    least 4096 elements; smaller ones keep separate scaling / decimation kernels).  Forward and inverse, with and without a coset
    shift, single and batched, in place and out of place, both tile sizes: the oracle's bytes."""
    B = 3
    x = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
    a = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
    b_ = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), B * n, 90 + n % 7)
    hx = x.cpu().numpy().view(np.uint64).reshape(B, n, 4)
    shift = oc.fr_from_ints([7, 12345678901234567890123])      # two different coset shifts
    try:
        for inv in (False, True):
            for cs in (None, shift[0], shift[1]):
                # the reference's wrappers (field_polynomial.rs:589-607): forward = scale by shift^j, then FFT; inverse = iFFT, then
                # scale by shift^j (the caller passes k^-1)
                if cs is None:
                    want = np.stack([oc.ntt(np.ascontiguousarray(hx[k]), inverse=inv, threads=4) for k in range(B)])
                elif not inv:
                    want = np.stack([oc.ntt(oc.mul_var(np.ascontiguousarray(hx[k]), cs), threads=4) for k in range(B)])
                else:
                    want = np.stack([oc.mul_var(oc.ntt(np.ascontiguousarray(hx[k]), inverse=True, threads=4), cs) for k in range(B)])
                want_t = torch.from_numpy(want.view(np.int64)).reshape(B * n, 4).cuda()
                for batch in (1, B):
                    for tile, two in ((1024, 1), (2048, 1), (0, 0)):      # (0, 0): sub-transforms of 2^17 in three passes instead of two
                        gpu.tune("ntt_tile", tile)
                        gpu.tune("ntt_two_pass", two)
                        gpu.ntt_batch_device(x.data_ptr(), a.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
                        assert torch.equal(a[:batch * n], want_t[:batch * n]), (inv, cs is not None, batch, tile, two)
                    gpu.tune("ntt_tile", 0)
                    gpu.tune("ntt_two_pass", 1)
                    b_[:batch * n] = x[:batch * n]                           # in place
                    torch.cuda.synchronize()                                 # torch's copy runs on torch's stream, the library on its own
                    gpu.ntt_batch_device(b_.data_ptr(), b_.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
                    assert torch.equal(b_[:batch * n], want_t[:batch * n]), ("in place", inv, cs is not None, batch)
    finally:
        gpu.tune("ntt_tile", 0)
