"""Contexts (uzk_ctx_create / uzk_ctx_set_current): every prover thread can own a stream, workspaces and a lock, so
independent proofs overlap on one GPU; SRS handles are process-wide.  Results must be what the default context gives."""
import threading

import numpy as np
import pytest

import oracle_c as oc
from util import affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu


def test_threads_with_their_own_contexts(gpu):
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)                       # registered in the default context, used by all
    srs.precompute(0)
    scal = [rand_fr_wire(4096, 900 + i) for i in range(4)]
    vecs = [rand_fr_wire(3 << 12, 950 + i) for i in range(4)]
    want_msm = [affine_of(oc.msm_pippenger(wire, s, 0, 4)) for s in scal]
    want_ntt = [oc.ntt(v) for v in vecs]
    errors, ctxs = [], []

    def worker(i):
        try:
            h = gpu.ctx_create()
            ctxs.append(h)
            gpu.ctx_set_current(h)
            for rep in range(5):
                assert affine_of(gpu.msm(srs, scal[i])) == want_msm[i]
                assert np.array_equal(gpu.ntt(vecs[i]), want_ntt[i])
                b3 = gpu.msm_batch(srs, np.stack([scal[i], scal[(i + 1) % 4], scal[i]]))
                assert [affine_of(x) for x in b3] == [want_msm[i], want_msm[(i + 1) % 4], want_msm[i]]
            gpu.ctx_set_current(0)
        except Exception as e:   # noqa: BLE001
            errors.append((i, e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the default context still works, and the contexts can be destroyed
    assert affine_of(gpu.msm(srs, scal[0])) == want_msm[0]
    for h in ctxs:
        gpu.ctx_destroy(h)
    from uzkge_amd import UzkgeError
    with pytest.raises(UzkgeError):
        gpu.ctx_set_current(ctxs[0])
    srs.release()


def test_profile_tables_are_per_context(gpu):
    wire, _ = load_srs("lagrange-srs-4096.bin")
    srs = gpu.Srs.from_host(wire)
    s = rand_fr_wire(1000, 5)
    h = gpu.ctx_create()
    try:
        gpu.profile_reset()
        gpu.ctx_set_current(h)
        gpu.profile_reset(); gpu.profile_enable(True)
        gpu.msm(srs, s)
        gpu.profile_enable(False)
        assert gpu.profile_table().get("msm_accumulate", (0, 0))[0] >= 1
        gpu.ctx_set_current(0)
        assert "msm_accumulate" not in gpu.profile_table()
    finally:
        gpu.ctx_set_current(0)
        gpu.ctx_destroy(h)
        srs.release()


def test_worker_thread_survives_shutdown_of_its_context(gpu):
    """a pooled worker that made a context current, then lives through uzk_shutdown + re-initialisation (or a
    uzk_ctx_destroy from another thread), must not touch the freed context: its handle no longer resolves and its next
    call runs on the default context."""
    x = rand_fr_wire(4096, 77)
    want = oc.ntt(x)
    go, done, out = threading.Event(), threading.Event(), {}

    def worker():
        try:
            h = gpu.ctx_create()
            gpu.ctx_set_current(h)
            out["first"] = bool(np.array_equal(gpu.ntt(x), want))
            done.set()
            go.wait(60)                                   # the main thread shuts the library down meanwhile
            out["after_shutdown"] = bool(np.array_equal(gpu.ntt(x), want))       # lazy re-init, default context
            h2 = gpu.ctx_create()
            gpu.ctx_set_current(h2)
            out["h2"] = h2
            out["own_again"] = bool(np.array_equal(gpu.ntt(x), want))
        except Exception as e:   # noqa: BLE001
            out["err"] = e
        finally:
            done.set()

    t = threading.Thread(target=worker)
    t.start()
    assert done.wait(120) and out.get("first") is True, out
    done.clear()
    gpu.shutdown()
    gpu.init(0)
    go.set()
    t.join(120)
    assert out.get("after_shutdown") is True and out.get("own_again") is True and "err" not in out, out
    gpu.ctx_destroy(out["h2"])                            # destroyed from ANOTHER thread than the one that uses it


def test_new_contexts_inherit_the_creators_tuning(gpu):
    """uzk_tune / uzk_msm_set_window_bits are per context; uzk_ctx_create copies the creator's current settings."""
    gpu.set_msm_window_bits(7)
    h = gpu.ctx_create()
    try:
        gpu.ctx_set_current(h)
        assert gpu.msm_plan_info(4096)[0] == 7
        gpu.set_msm_window_bits(0)
        gpu.ctx_set_current(0)
        assert gpu.msm_plan_info(4096)[0] == 7            # the default context keeps its own value
    finally:
        gpu.ctx_set_current(0)
        gpu.set_msm_window_bits(0)
        gpu.ctx_destroy(h)


def test_ctx_wait_orders_two_contexts_of_one_thread(gpu):
    """uzk_ctx_wait: a thread that owns two contexts (two streams, two workspace sets) makes one wait, on the device, for what
    the other has queued.  Context B transforms a vector context A is still producing: without the edge B would read it too
    early; with it the result is the oracle's.  uzk_ctx_current names the thread's context."""
    from uzkge_amd import UzkgeError
    n, batch = 1 << 14, 24
    x = np.stack([rand_fr_wire(n, 700 + i) for i in range(batch)])
    d_x, d_y, d_z = gpu.dev_alloc(batch * n * 32), gpu.dev_alloc(batch * n * 32), gpu.dev_alloc(batch * n * 32)
    a = gpu.ctx_current()
    assert a == 0
    h = gpu.ctx_create()
    try:
        gpu.dev_upload(d_x, x)
        for rep in range(3):
            gpu.ntt_batch_device(d_x, d_y, n, batch)               # context A: asynchronous
            gpu.ctx_set_current(h)
            assert gpu.ctx_current() == h
            gpu.ctx_wait(a)                                        # B: after everything A has queued
            gpu.ntt_batch_device(d_y, d_z, n, batch, inverse=True, sync=True)
            assert np.array_equal(gpu.dev_download(d_z, (batch, n, 4)), x), rep
            gpu.ctx_set_current(a)
            gpu.ctx_wait(h)                                        # and back: A may overwrite d_y only after B has read it
        gpu.ctx_wait(a)                                            # waiting for oneself is a no-op
        with pytest.raises(UzkgeError):
            gpu.ctx_wait(987654)
    finally:
        gpu.ctx_set_current(0)
        gpu.ctx_destroy(h)
        for p in (d_x, d_y, d_z):
            gpu.dev_free(p)
