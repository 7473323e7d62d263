"""NTT parity at the top of the range the boundary accepts: every size class `uzk_domain_supported` says yes to has a whole-vector
comparison with the CPU oracle here -- three-pass plans (2^23 = 8 + 8 + 7, 2^24 = 8 + 8 + 8), the FIRST FOUR-PASS plan
(2^25 = 7 + 6 + 6 + 6: the `npass > 2` and odd / even in-place buffer schedules of `ntt_pow2`), and the mixed-radix domains
3 * 2^20 / 3 * 2^22 (radix-3 stage in the first pass over sub-transforms of 2^20 / 2^22).  Forward and inverse, out of place and
in place on the device, the host-pointer call, one coset case each, a round trip.  The largest sizes here ARE the supported
bound (include/uzkge_gpu.h: UZK_NTT_MAX_LOG2, UZK_NTT_MAX_LOG2_MIXED): an accepted size is a promised result.
Reference: FpPolynomial::{fft_with_domain, ifft_with_domain, coset_fft_with_domain, coset_ifft_with_domain},
uzkge/src/poly_commit/field_polynomial.rs:554-567,583-607."""
import os
import re

import numpy as np
import pytest
import torch

import bn254_py as opy
import oracle_c as oc
from uzkge_amd import UzkgeError

pytestmark = pytest.mark.gpu

MAX_LOG2 = 25          # == UZK_NTT_MAX_LOG2
MAX_LOG2_MIXED = 22    # == UZK_NTT_MAX_LOG2_MIXED
LARGE = [1 << 23, 1 << 24, 1 << MAX_LOG2, 3 << 20, 3 << MAX_LOG2_MIXED]
THREADS = 16


def _dev(h):
    return torch.from_numpy(h.view(np.int64)).reshape(-1, 4).cuda()


@pytest.mark.parametrize("n", LARGE)
def test_large_transform_matches_oracle(gpu, n):
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), n, 600 + n % 1000)
    hx = x.cpu().numpy().view(np.uint64).reshape(n, 4)
    fwd = None
    for inv in (False, True):
        want = _dev(oc.ntt(hx, inverse=inv, threads=THREADS))
        gpu.ntt_device(x.data_ptr(), a.data_ptr(), n, inverse=inv, sync=True)
        assert torch.equal(a, want), (n, inv, "out of place")
        a.copy_(x)
        torch.cuda.synchronize()
        gpu.ntt_device(a.data_ptr(), a.data_ptr(), n, inverse=inv, sync=True)
        assert torch.equal(a, want), (n, inv, "in place")
        if not inv:
            fwd = want
    # ifft(fft(x)) == x, in place on the forward result
    gpu.ntt_device(fwd.data_ptr(), fwd.data_ptr(), n, inverse=True, sync=True)
    assert torch.equal(fwd, x), (n, "round trip")
    del fwd, want
    # the host-pointer entry point (what the two-call-site integration calls): uzk_ntt_fr transforms the caller's Vec in place
    h = hx.copy()
    gpu.ntt_inplace(h, inverse=True)
    assert np.array_equal(h.view(np.int64), a.cpu().numpy()), (n, "host pointer")


@pytest.mark.parametrize("n", LARGE)
def test_large_coset_transform_matches_oracle(gpu, n):
    """coset_fft_with_domain = mul_var(k) then fft; coset_ifft_with_domain = ifft then mul_var(k^-1) (field_polynomial.rs:589-607)."""
    k = 7
    kw = oc.fr_from_ints([k])[0]
    kinv = oc.fr_from_ints([pow(k, -1, opy.R)])[0]
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), n, 700 + n % 1000)
    hx = x.cpu().numpy().view(np.uint64).reshape(n, 4)
    want = _dev(oc.ntt(oc.mul_var(hx, kw), threads=THREADS))
    a = torch.empty_like(x)
    torch.cuda.synchronize()
    gpu.ntt_device(x.data_ptr(), a.data_ptr(), n, coset_shift=kw, sync=True)
    assert torch.equal(a, want), (n, "coset forward")
    gpu.ntt_device(a.data_ptr(), a.data_ptr(), n, inverse=True, coset_shift=kinv, sync=True)
    assert torch.equal(a, x), (n, "coset inverse undoes coset forward")


@pytest.mark.parametrize("n,batch", [(1 << 23, 4), (3 << 20, 3)])
def test_large_batch_equals_single_transforms(gpu, n, batch):
    """More than 2^24 elements in one launch (the 2048-element tiles) as a batch: each vector equals its single transform."""
    x = torch.empty((batch * n, 4), dtype=torch.int64, device="cuda")
    out = torch.empty_like(x)
    one = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu.synth_scalars(x.data_ptr(), batch * n, 55)
    for inv in (False, True):
        gpu.ntt_batch_device(x.data_ptr(), out.data_ptr(), n, batch, inverse=inv, sync=True)
        for j in range(batch):
            gpu.ntt_device(x.data_ptr() + j * n * 32, one.data_ptr(), n, inverse=inv, sync=True)
            assert torch.equal(out[j * n:(j + 1) * n], one), (n, batch, inv, j)
    # anchor one vector of the batch on the oracle
    hx = x[:n].cpu().numpy().view(np.uint64).reshape(n, 4)
    gpu.ntt_batch_device(x.data_ptr(), out.data_ptr(), n, batch, sync=True)
    assert torch.equal(out[:n], _dev(oc.ntt(hx, threads=THREADS)))


def test_supported_bound_is_the_tested_bound(gpu):
    """Sizes above the largest one compared with the oracle are refused with FFTError -- never computed unchecked.  The domain
    itself exists up to 2^28 (group_gen stays answerable: host arithmetic pinned by tests/test_oracle_pinning.py)."""
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "uzkge_gpu.h")).read()
    assert int(re.search(r"#define UZK_NTT_MAX_LOG2 (\d+)", hdr).group(1)) == MAX_LOG2
    assert int(re.search(r"#define UZK_NTT_MAX_LOG2_MIXED (\d+)", hdr).group(1)) == MAX_LOG2_MIXED
    for n in (1 << MAX_LOG2, 3 << MAX_LOG2_MIXED):
        assert gpu.domain_supported(n)
    for n in (1 << (MAX_LOG2 + 1), 3 << (MAX_LOG2_MIXED + 1), 1 << 28):
        assert not gpu.domain_supported(n)
        with pytest.raises(UzkgeError) as e:
            gpu.ntt_device(1, 1, n, sync=True)          # refused before any pointer is touched
        assert e.value.kind == "FFTError"
    assert np.array_equal(gpu.domain_group_gen(1 << 28), oc.root_of_unity(1 << 28))
