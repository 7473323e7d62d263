"""uzk_dev_* / uzk_host_*: the device-memory entry points that make the device-resident flow reachable through the C ABI
alone (no HIP binding in the host language).  Round trips, pitched copies, fills, the pinned-upload rule, error mapping,
and the whole point: a transform and a commit on buffers that only these calls ever touched."""
import ctypes

import numpy as np
import pytest

import oracle_c as oc
from util import affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu


def test_round_trip_and_pitched_copies(gpu):
    rows, w, pitch = 5, 96, 160
    src = np.arange(rows * pitch, dtype=np.uint8).reshape(rows, pitch)
    d_a, d_b = gpu.dev_alloc(rows * pitch), gpu.dev_alloc(rows * w)
    try:
        gpu.dev_upload(d_a, src)
        assert np.array_equal(gpu.dev_download(d_a, (rows, pitch), np.uint8), src)
        gpu.dev_copy2d(d_b, w, d_a, pitch, w, rows)                       # gather the row heads
        assert np.array_equal(gpu.dev_download(d_b, (rows, w), np.uint8), src[:, :w])
        gpu.dev_memset2d(d_a, pitch, 0xAB, 7, rows)
        got = gpu.dev_download(d_a, (rows, pitch), np.uint8)
        assert (got[:, :7] == 0xAB).all() and np.array_equal(got[:, 7:], src[:, 7:])
        gpu.dev_memset(d_b, 0, rows * w)
        assert not gpu.dev_download(d_b, (rows * w,), np.uint8).any()
        heads = np.zeros((rows, 8), dtype=np.uint8)                       # pitched device -> host
        gpu.dev_copy2d(heads.ctypes.data, 8, d_a + 7, pitch, 8, rows, gpu.COPY_D2H)
        assert np.array_equal(heads, src[:, 7:15])
    finally:
        gpu.dev_free(d_a); gpu.dev_free(d_b)


def test_pinned_uploads_are_ordered_with_the_stream(gpu):
    """Uploads from uzk_host_alloc memory are asynchronous but stream-ordered: a D2H copy issued afterwards sees them."""
    n = 1 << 16
    h = gpu.host_alloc(n * 8)
    d = gpu.dev_alloc(n * 8)
    try:
        view = np.ctypeslib.as_array(ctypes.cast(h, ctypes.POINTER(ctypes.c_uint64)), shape=(n,))
        for rep in range(3):
            view[:] = np.arange(n, dtype=np.uint64) * (rep + 3)
            gpu.dev_copy2d(d, n * 8, h, n * 8, n * 8, 1, gpu.COPY_H2D)
            assert np.array_equal(gpu.dev_download(d, (n,)), view)
    finally:
        gpu.dev_free(d); gpu.host_free(h)


def test_argument_errors(gpu):
    from uzkge_amd import UzkgeError
    d = gpu.dev_alloc(64)
    try:
        with pytest.raises(UzkgeError):
            gpu.dev_copy2d(d, 8, d, 16, 16, 2)                            # pitch < width
        with pytest.raises(UzkgeError):
            gpu.dev_copy2d(d, 16, d, 16, 16, 1, 7)                        # unknown kind
        with pytest.raises(UzkgeError):
            gpu.host_free(d)                                              # not a uzk_host_alloc block
        assert gpu.dev_alloc(0) == 0
        gpu.dev_free(0); gpu.host_free(0)                                 # null frees are no-ops
        with pytest.raises(UzkgeError) as e:
            gpu.dev_alloc(1 << 60)
        assert e.value.kind == "DeviceError"
    finally:
        gpu.dev_free(d)


def test_transform_and_commit_on_abi_only_buffers(gpu):
    """No torch, no HIP binding: buffers from uzk_dev_alloc carry a batched inverse transform, a pitched gather and a
    batched commit; results against the oracle."""
    n, batch = 4096, 3
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)
    x = np.stack([rand_fr_wire(n, 40 + b) for b in range(batch)])
    d_x, d_y = gpu.dev_alloc(batch * n * 32), gpu.dev_alloc(batch * n * 32)
    try:
        gpu.dev_upload(d_x, x)
        gpu.ntt_batch_device(d_x, d_y, n, batch, inverse=True)
        y = gpu.dev_download(d_y, (batch, n, 4))
        for b in range(batch):
            assert np.array_equal(y[b], oc.ntt(x[b], inverse=True))
        got = gpu.msm_batch_device(srs, d_x, n, batch)
        for b in range(batch):
            assert affine_of(got[b]) == affine_of(oc.msm_pippenger(wire, x[b], 0, 4))
    finally:
        gpu.dev_free(d_x); gpu.dev_free(d_y); srs.release()
