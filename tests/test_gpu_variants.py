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
    {"msm_scan_reduce": 0}, {"msm_scan_reduce": 2}, {"msm_reduce_seg": 4}, {"msm_task_len": 5}, {"msm_task_len": 300},
    {"msm_fold_group": 1}, {"msm_fold_group": 16}, {"window_bits": 9}, {"window_bits": 12}, {"window_bits": 16}, {"window_bits": 17},
]
DEFAULTS = {"msm_acc_variant": 0, "msm_sort_packed": 1, "msm_fused_hist": 1, "msm_scan_reduce": 1, "msm_reduce_seg": 0,
            "msm_task_len": 0, "msm_fold_group": 0, "window_bits": 0}


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
    finally:
        gpu.tune("ntt_l29", 1)
