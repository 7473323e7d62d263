"""Every selectable code path of the MSM gives the same commitment: accumulator variants (29-bit limbs,
8 x 32-bit relaxed, canonical), packed / unpacked sort entries, fused / separate first histogram,
both bucket reductions, forced task lengths and fold-group widths, window sizes on either side of the
automatic choice -- and the two NTT arithmetic variants give identical bytes."""
import numpy as np
import pytest
import torch

import oracle_c as oc
from util import affine_of

pytestmark = pytest.mark.gpu

KNOBS = [
    {}, {"msm_acc_variant": 1}, {"msm_acc_variant": 2}, {"msm_sort_packed": 0}, {"msm_fused_hist": 0},
    {"msm_quad_reduce": 0}, {"msm_scan_reduce": 0}, {"msm_scan_reduce": 2}, {"msm_scan_reduce": 3}, {"msm_scan_reduce": 3, "msm_reduce_seg": 16},
    {"msm_scan_reduce": 3, "window_bits": 17}, {"msm_scan_reduce": 3, "window_bits": 9}, {"msm_reduce_seg": 4}, {"msm_task_len": 5}, {"msm_task_len": 300},
    {"msm_fold_group": 1}, {"msm_fold_group": 16}, {"window_bits": 9}, {"window_bits": 12}, {"window_bits": 16}, {"window_bits": 17},
    # round 3: segment / chunk sort kernels and the class-sum reduction, off one at a time and at window widths on either side
    {"msm_seg_sort": 0}, {"msm_chunk_sort": 0}, {"msm_class_reduce": 0}, {"msm_class_reduce": 0, "window_bits": 13}, {"window_bits": 13},
    {"window_bits": 14}, {"window_bits": 18}, {"window_bits": 19}, {"msm_class_reduce": 0, "window_bits": 19}, {"msm_seg_sort": 15, "window_bits": 15},
    {"msm_bucket_fill": 0}, {"msm_bucket_fill": 0, "msm_task_len": 5}, {"msm_task_len": 3}, {"msm_fold_big": 0, "msm_task_len": 2}, {"msm_direct": 0}, {"msm_direct": 0, "msm_task_len": 2}, {"msm_acc_variant": 1, "msm_task_len": 40},
]
DEFAULTS = {"msm_small": 1, "msm_fold_mode": 0, "msm_acc_variant": 0, "msm_sort_packed": 1, "msm_fused_hist": 1, "msm_scan_reduce": 1, "msm_quad_reduce": 1, "msm_x29": 1, "msm_reduce_seg": 0,
            "msm_task_len": 0, "msm_fold_group": 0, "window_bits": 0, "msm_seg_sort": 1, "msm_chunk_sort": 1, "msm_class_reduce": 1, "msm_bucket_fill": 1, "msm_fold_big": 1, "msm_direct": 1}


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


@pytest.mark.parametrize("n", [4096, 1 << 15, 3 << 13, 1 << 19])
def test_ntt_variants_agree(gpu, n):
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    b_ = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), n, 9)
    try:
        for inv in (False, True):
            gpu.tune("ntt_l29", 0); gpu.ntt_device(x.data_ptr(), a.data_ptr(), n, inverse=inv, sync=True)
            gpu.tune("ntt_l29", 1); gpu.ntt_device(x.data_ptr(), b_.data_ptr(), n, inverse=inv, sync=True)
            assert torch.equal(a, b_)
            for tile in (1024, 2048):      # both workgroup tile sizes of the pass kernels (0 = chosen by size)
                for mulc, planes in ((1, 2), (0, 2), (1, 0), (0, 0), (2, 0), (2, 2)):   # tile twiddles by the constant-operand product (default; 2: with the full reduce) /
                    gpu.tune("ntt_tile", tile); gpu.tune("ntt_mulc", mulc)     # Montgomery products throughout; limb planes (default) / 8 x 32-bit
                    gpu.tune("ntt_planes", planes)                             # words between the passes
                    gpu.ntt_device(x.data_ptr(), a.data_ptr(), n, inverse=inv, sync=True)
                    assert torch.equal(a, b_), (tile, mulc, planes)
                    if planes:             # in place and batched through the plane buffers
                        y = x.clone()
                        gpu.ntt_device(y.data_ptr(), y.data_ptr(), n, inverse=inv, sync=True)
                        assert torch.equal(y, b_), (tile, mulc, "in place")
            gpu.tune("ntt_tile", 0); gpu.tune("ntt_mulc", 1); gpu.tune("ntt_planes", 1)
    finally:
        gpu.tune("ntt_l29", 1); gpu.tune("ntt_tile", 0); gpu.tune("ntt_mulc", 1); gpu.tune("ntt_planes", 1)


@pytest.mark.parametrize("n", [1, 2, 33, 1000, 4096, 16384, 32768])
def test_small_pipeline_agrees_with_general(gpu, n):
    """n <= 2^15 takes the one-workgroup-per-slot pipeline (msm_small_*): same commitments as the general
    pipeline for uniform, prover-mix, all-equal and all-zero scalar vectors, single and batched, at every
    window size the small path accepts, and for forced task lengths on both sides of the automatic one."""
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
        for cfg in ({}, {"window_bits": 5}, {"window_bits": 9}, {"window_bits": 10}, {"msm_task_len": 2}, {"msm_task_len": 3}, {"msm_task_len": 200},
                    {"msm_acc_variant": 1}, {"msm_acc_variant": 2}, {"msm_x29": 0}, {"msm_x29": 0, "window_bits": 10}, {"msm_fold_mode": 1 + 16 + 8}, {"msm_fold_mode": 1 + 16 + 2},
                    {"msm_fold_mode": 1 + 8}, {"msm_fold_mode": 1 + 4}, {"msm_fold_mode": 1 + 2}, {"msm_fold_mode": 1 + 1}):
            _apply(gpu, cfg)
            gpu.tune("msm_small", 1)
            assert run() == want, cfg
        _apply(gpu, {})
        # window table (uzk_srs_precompute) on the small pipeline: per-window rows, window sums added without doublings
        gpu.tune("msm_no_precompute", 0)
        for c in (0, 6, 9):
            srs.precompute(c)
            assert run() == want, ("precompute", c)
        gpu.tune("msm_small", 2)                                   # one lane per addition in the folds and scans
        assert run() == want
    finally:
        _apply(gpu, {})
        gpu.tune("msm_small", 1)
        srs.release()


@pytest.mark.parametrize("n", [4096, 1 << 14, 3 << 12, 3 << 13, 98304, 1 << 17, 3 << 16])
def test_ntt_fused_stages_agree_with_separate_kernels(gpu, n):
    """Coset scaling and the radix-3 stage of 3 * 2^k domains run inside the first / last Stockham pass
    (uzk_tune("ntt_fused", 1), the default); the separate scaling / decimation / combination kernels remain behind
    ntt_fused = 0.  Same bytes for forward and inverse, with and without a coset shift, single and batched, in place
    and out of place -- and the small sizes are anchored on the oracle."""
    B = 3
    x = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
    a = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
    b_ = torch.empty((B * n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), B * n, 90 + n % 7)
    shift = oc.fr_from_ints([7, 12345678901234567890123])      # two different coset shifts
    try:
        for inv in (False, True):
            for cs in (None, shift[0], shift[1]):
                for batch in (1, B):
                    gpu.tune("ntt_tile", 2048 if (n // 4096) % 2 else 1024)
                    gpu.tune("ntt_fused", 0)
                    gpu.ntt_batch_device(x.data_ptr(), a.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
                    gpu.tune("ntt_fused", 1); gpu.tune("ntt_tile", 1024 if (n // 4096) % 2 else 2048)      # the other tile size
                    gpu.ntt_batch_device(x.data_ptr(), b_.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
                    assert torch.equal(a[:batch * n], b_[:batch * n]), (inv, cs is not None, batch)
                    gpu.tune("ntt_tile", 0)
                    b_[:batch * n] = x[:batch * n]                           # in place
                    torch.cuda.synchronize()                                 # torch's copy runs on torch's stream, the library on its own
                    gpu.ntt_batch_device(b_.data_ptr(), b_.data_ptr(), n, batch, inverse=inv, coset_shift=cs, sync=True)
                    assert torch.equal(a[:batch * n], b_[:batch * n]), ("in place", inv, cs is not None, batch)
        if n <= 98304:
            hx = x[:n].cpu().numpy().view(np.uint64).reshape(-1, 4)
            gpu.ntt_device(x.data_ptr(), a.data_ptr(), n, coset_shift=shift[0], sync=True)
            want = oc.ntt(oc.mul_var(hx, shift[0]))
            assert np.array_equal(a[:n].cpu().numpy().view(np.uint64).reshape(-1, 4), want)
    finally:
        gpu.tune("ntt_fused", 1); gpu.tune("ntt_tile", 0)
