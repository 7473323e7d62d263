"""Threading and lifecycle of the C ABI: callers are prover threads that may overlap
(the ABI must be thread-safe, no thread affinity), and a process may shut the
backend down and bring it up again."""
import threading

import numpy as np
import pytest

import oracle_c as oc
from util import affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu


def test_concurrent_callers(gpu):
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)
    errors = []

    def ntt_worker(seed):
        try:
            for k in range(6):
                n = (1024, 4096, 3 << 10)[k % 3]
                x = rand_fr_wire(n, seed * 100 + k)
                assert np.array_equal(gpu.ntt(x), oc.ntt(x))
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    def msm_worker(seed):
        try:
            for k in range(6):
                n = (33, 1000, 4096)[k % 3]
                s = rand_fr_wire(n, seed * 1000 + k)
                assert affine_of(gpu.msm(srs, s)) == affine_of(oc.msm_pippenger(wire[:n], s, 0, 1))
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=ntt_worker, args=(1,)), threading.Thread(target=msm_worker, args=(2,)),
               threading.Thread(target=ntt_worker, args=(3,)), threading.Thread(target=msm_worker, args=(4,))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    srs.release()
    assert not errors, errors


def test_shutdown_and_reinit(gpu):
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)
    s = rand_fr_wire(500, 9)
    before = affine_of(gpu.msm(srs, s))
    gpu.shutdown()                       # frees every workspace, plan, SRS and table
    from uzkge_amd import UzkgeError
    with pytest.raises(UzkgeError):      # the old handle is gone
        gpu.msm(srs, s)
    gpu.init(0)
    srs2 = gpu.Srs.from_host(wire)
    try:
        assert affine_of(gpu.msm(srs2, s)) == before
        x = rand_fr_wire(8192, 3)
        assert np.array_equal(gpu.ntt(x), oc.ntt(x))
    finally:
        srs2.release()


def test_calls_from_a_fresh_thread_use_the_bound_device(gpu):
    """HIP's current device is per thread: every entry point rebinds the calling thread to the device uzk_init
    chose.  A thread that has never touched HIP registers an SRS, runs an MSM and an NTT."""
    wire, _ = load_srs("lagrange-srs-4096.bin")
    out = {}

    def worker():
        try:
            srs = gpu.Srs.from_host(wire[:777])
            s = rand_fr_wire(777, 21)
            out["msm"] = affine_of(gpu.msm(srs, s)) == affine_of(oc.msm_pippenger(wire[:777], s, 0, 1))
            x = rand_fr_wire(4096, 22)
            out["ntt"] = bool(np.array_equal(gpu.ntt(x), oc.ntt(x)))
            srs.release()
        except Exception as e:   # noqa: BLE001
            out["err"] = e

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert out == {"msm": True, "ntt": True}, out


def test_reinit_returns_to_the_last_bound_device(gpu):
    """After uzk_shutdown a lazy re-initialisation (a compute call without uzk_init) goes back to the device the
    process was bound to, not silently to device 0."""
    gpu.shutdown()
    x = rand_fr_wire(2048, 5)
    assert np.array_equal(gpu.ntt(x), oc.ntt(x))       # lazy init
    gpu.init(0)                                         # same ordinal: idempotent
