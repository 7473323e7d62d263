"""The batched / strided / pointer-list entry points that let one prover step be one launch (VERDICT r2 task 5): each one
against the single-polynomial entry point it batches, against the oracle, or against the reference's lines restated on ints.
Buffers come from uzk_dev_alloc: nothing here needs torch."""
import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import affine_of, load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu
R = opy.R


class Dev:
    """A device buffer of `count` field elements from uzk_dev_alloc."""

    def __init__(self, gpu, count, init=None):
        self.gpu, self.count = gpu, count
        self.ptr = gpu.dev_alloc(count * 32)
        if init is not None:
            gpu.dev_upload(self.ptr, np.ascontiguousarray(init, dtype=np.uint64).reshape(-1, 4))
        else:
            gpu.dev_memset(self.ptr, 0xA5, count * 32)           # poison: whatever the kernel must write, it must write

    def at(self, elem):
        return self.ptr + 32 * elem

    def get(self):
        return self.gpu.dev_download(self.ptr, (self.count, 4))

    def free(self):
        self.gpu.dev_free(self.ptr)


@pytest.mark.parametrize("n,coset", [(4096, False), (1 << 14, False), (1 << 14, True), (3 << 12, True), (98304, True), (1024, False),
                                     (1024, True), (3 << 8, False), (1 << 16, False)])
def test_strided_ntt_equals_contiguous(gpu, n, coset):
    batch, in_stride, out_stride = 3, n + 5, 2 * n + 3
    k = rand_fr_wire(1, 9)[0] if coset else None
    for inverse in (False, True):
        x = np.stack([rand_fr_wire(n, 100 + b + 10 * inverse) for b in range(batch)])
        wide = np.zeros((batch, in_stride, 4), dtype=np.uint64)
        wide[:, :n] = x
        wide[:, n:] = rand_fr_wire(batch * 5, 7).reshape(batch, 5, 4)            # junk between the vectors must not matter
        d_in, d_out, d_ref = Dev(gpu, batch * in_stride, wide), Dev(gpu, batch * out_stride), Dev(gpu, batch * n)
        d_x = Dev(gpu, batch * n, x)
        try:
            before = d_out.get().reshape(batch, out_stride, 4)
            gpu.ntt_batch_strided_device(d_in.ptr, in_stride, d_out.ptr, out_stride, n, batch, inverse=inverse, coset_shift=k)
            gpu.ntt_batch_device(d_x.ptr, d_ref.ptr, n, batch, inverse=inverse, coset_shift=k)
            got = d_out.get().reshape(batch, out_stride, 4)
            want = d_ref.get().reshape(batch, n, 4)
            assert np.array_equal(got[:, :n], want)
            assert np.array_equal(got[:, n:], before[:, n:])                         # the gaps of the output are not touched
            if n <= 1 << 14 and not coset:
                assert np.array_equal(want[1], oc.ntt(x[1], inverse=inverse))
            # in place with equal strides
            gpu.ntt_batch_strided_device(d_in.ptr, in_stride, d_in.ptr, in_stride, n, batch, inverse=inverse, coset_shift=k)
            assert np.array_equal(d_in.get().reshape(batch, in_stride, 4)[:, :n], want)
        finally:
            for d in (d_in, d_out, d_ref, d_x):
                d.free()


def test_strided_ntt_argument_errors(gpu):
    from uzkge_amd import UzkgeError
    d = Dev(gpu, 3 * 4096)
    try:
        with pytest.raises(UzkgeError):
            gpu.ntt_batch_strided_device(d.ptr, 4000, d.ptr, 4000, 4096, 2)              # stride < n
        with pytest.raises(UzkgeError):
            gpu.ntt_batch_strided_device(d.ptr, 4096, d.ptr, 6000, 4096, 2)              # in place, different strides
    finally:
        d.free()


@pytest.mark.parametrize("precompute", [False, True])
def test_tail_msm_is_commit_plus_blind_factors(gpu, precompute):
    """Bases lagrange || pcs[0..3) || pcs[n..n+3), tail = blinds || -blinds: commit(evals) + apply_blind_factors
    (prover.rs:132-142, kzg_poly_commitment.rs:299-313), against the oracle's MSM and scalar multiplications over the
    reference's own SRS files."""
    from uzkge_amd import poly_commit as pc
    n, batch, stride = 4096, 5, 4096 + 11
    lag_wire, _ = load_srs("lagrange-srs-4096.bin")
    import os
    from util import GOLDEN
    mono = pc.srs_params_wire(open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read(), n)
    bases = np.concatenate([lag_wire, mono[:3], mono[n:n + 3]])
    srs = gpu.Srs.from_host(bases)
    if precompute:
        srs.precompute(0)
    evals = np.stack([rand_fr_wire(n, 300 + b) for b in range(batch)])
    evals[1] = 0                                                                  # an all-zero vector: only the blinds remain
    blinds = [rand_fr_wire(k, 320 + b) for b, k in enumerate((2, 3, 3, 0, 1))]
    tail = np.zeros((batch, 6, 4), dtype=np.uint64)
    for b, bl in enumerate(blinds):
        tail[b, : bl.shape[0]] = bl
        tail[b, 3:3 + bl.shape[0]] = pc.fr_neg(bl)
    wide = np.zeros((batch, stride, 4), dtype=np.uint64)
    wide[:, :n] = evals
    wide[:, n:] = rand_fr_wire(batch * 11, 5).reshape(batch, 11, 4)
    d_s, d_tail = Dev(gpu, batch * stride, wide), Dev(gpu, batch * 6, tail)
    try:
        got_host_tail = gpu.msm_batch_tail_device(srs, d_s.ptr, stride, n, batch, tail, 6)
        got_dev_tail = gpu.msm_batch_tail_device(srs, d_s.ptr, stride, n, batch, d_tail.ptr, 6)
        mono_pts = {i: opy.wire_to_affine(mono[i].tobytes()) for i in (0, 1, 2, n, n + 1, n + 2)}
        for b in range(batch):
            want = affine_of(oc.msm_pippenger(lag_wire, evals[b], 0, 4))
            for i, bl in enumerate(oc.fr_to_ints(blinds[b])):
                want = opy.g1_add(want, opy.g1_mul(mono_pts[i], bl))
                want = opy.g1_add(want, opy.g1_mul(mono_pts[n + i], (-bl) % R))
            assert affine_of(got_host_tail[b]) == want, b
            assert affine_of(got_dev_tail[b]) == want, b
        # no tail at all, batch 1, stride == n: the plain entry point's result
        one = gpu.msm_batch_tail_device(srs, d_s.ptr, stride, n, 1, None, 0)
        assert affine_of(one[0]) == affine_of(oc.msm_pippenger(lag_wire, evals[0], 0, 4))
    finally:
        d_s.free(); d_tail.free(); srs.release()


def test_tail_msm_general_pipeline(gpu):
    """n above the small pipeline's limit (2^15): the digit kernels of the general pipeline read the same view."""
    n, tail_n, batch, stride = (1 << 16), 4, 2, (1 << 16) + 3
    pts = Dev(gpu, 2 * (n + tail_n))                 # affine points are two field elements each
    seed = rand_fr_wire(1, 77)[0]
    gpu.synth_points_arith(pts.ptr, n + tail_n, seed)             # P_i = (i + 1) Q
    srs = gpu.Srs.from_device(pts.ptr, n + tail_n)
    sc = np.stack([rand_fr_wire(stride, 400 + b) for b in range(batch)])
    tail = rand_fr_wire(batch * tail_n, 410).reshape(batch, tail_n, 4)
    d_s = Dev(gpu, batch * stride, sc)
    try:
        got = gpu.msm_batch_tail_device(srs, d_s.ptr, stride, n, batch, tail, tail_n)
        q = opy.g1_mul(opy.G1_GEN, oc.fr_to_ints(seed.reshape(1, 4))[0])
        for b in range(batch):
            s_main, s_tail = oc.fr_to_ints(sc[b, :n]), oc.fr_to_ints(tail[b])
            tot = (sum((i + 1) * v for i, v in enumerate(s_main)) + sum((n + j + 1) * v for j, v in enumerate(s_tail))) % R
            assert affine_of(got[b]) == opy.g1_mul(q, tot), b
    finally:
        d_s.free(); srs.release(); pts.free()


def test_hide_polynomial_batch_follows_the_reference_resize(gpu):
    n, stride, count, hd = 1000, 1300, 5, 3
    blinds = rand_fr_wire(count * hd, 50).reshape(count, hd, 4)
    for len_in in (n, n + 2, n - 7):
        buf = np.stack([rand_fr_wire(stride, 60 + i) for i in range(count)])      # junk beyond len_in: must be overwritten up to n + hd
        d = Dev(gpu, count * stride, buf)
        try:
            gpu.hide_polynomial_batch_device(d.ptr, stride, len_in, blinds, n)
            got = d.get().reshape(count, stride, 4)
            for i in range(count):
                c = oc.fr_to_ints(buf[i, :len_in]) + [0] * max(0, n + hd - len_in)    # helpers.rs:146-148: resize with zeros
                for j, bl in enumerate(oc.fr_to_ints(blinds[i])):
                    c[j] = (c[j] + bl) % R
                    c[n + j] = (c[n + j] - bl) % R
                top = max(len_in, n + hd)
                assert oc.fr_to_ints(got[i, :top]) == c[:top], (len_in, i)
                assert np.array_equal(got[i, top:], buf[i, top:])                     # nothing beyond is touched
        finally:
            d.free()


def test_fold_blinds_batch_and_device_tail(gpu):
    N, in_stride, out_stride, K = 4096, 4096 + 16, 4096 + 7, 3
    lens = [N + 3, N, N - 5, N + 1, N + 2]
    polys = np.stack([rand_fr_wire(in_stride, 70 + i) for i in range(len(lens))])
    d_p, d_o, d_t = Dev(gpu, len(lens) * in_stride, polys), Dev(gpu, len(lens) * out_stride), Dev(gpu, len(lens) * 2 * K)
    try:
        blinds = gpu.fold_blinds_batch_device(d_p.ptr, in_stride, lens, N, d_o.ptr, out_stride, d_t.ptr, 2 * K, want_blinds=True)
        out = d_o.get().reshape(len(lens), out_stride, 4)
        tail = d_t.get().reshape(len(lens), 2 * K, 4)
        for b, ln in enumerate(lens):
            ci = oc.fr_to_ints(polys[b, :ln])
            want = [(ci[i] if i < ln else 0) for i in range(N)]
            bl = [0] * K
            for i in range(max(0, ln - N)):
                want[i] = (want[i] + ci[N + i]) % R
                bl[i] = (-ci[N + i]) % R
            assert oc.fr_to_ints(out[b, :N]) == want, b
            assert oc.fr_to_ints(tail[b, :K]) == bl and oc.fr_to_ints(tail[b, K:]) == [(-v) % R for v in bl], b
            assert oc.fr_to_ints(blinds[b]) == bl
            # the single-polynomial entry point agrees
            d_one = Dev(gpu, N)
            single = gpu.fold_blinds_device(d_p.at(b * in_stride), ln, N, d_one.ptr)
            gpu.sync()
            assert np.array_equal(d_one.get(), out[b, :N]) and oc.fr_to_ints(single) == bl[: max(0, ln - N)]
            d_one.free()
    finally:
        d_p.free(); d_o.free(); d_t.free()


@pytest.mark.parametrize("t_len_extra", [8, 0, -4100, -3 * 4098 - 5])
def test_split_t_follows_the_reference(gpu, t_len_extra):
    """helpers.rs:1335-1363 restated on ints, with the reference's own parameter (chunk = n_constraints + 2) and t lengths
    that end inside the last chunk, at its start (the `[-prev]` branch) and two chunks early."""
    n, n_chunks = 4096, 5
    chunk, stride = n + 2, n + 8
    t_len = 4 * chunk + t_len_extra if t_len_extra <= 0 else 5 * n + t_len_extra
    t = rand_fr_wire(6 * n, 90)
    rands = rand_fr_wire(n_chunks, 91)
    d_t, d_c = Dev(gpu, 6 * n, t), Dev(gpu, n_chunks * stride)
    try:
        lens = gpu.split_t_device(d_t.ptr, t_len, chunk, rands, d_c.ptr, stride)
        gpu.sync()
        got = d_c.get().reshape(n_chunks, stride, 4)
        ti, ri = oc.fr_to_ints(t[:t_len]), oc.fr_to_ints(rands)
        prev = 0
        for i in range(n_chunks):
            start = i * chunk
            end = t_len if i == n_chunks - 1 else (i + 1) * chunk
            coefs = ti[start:min(t_len, end)] if start < t_len else []
            if i != n_chunks - 1:
                coefs = coefs + [0] * (chunk + 1 - len(coefs))
                coefs[chunk] = (coefs[chunk] + ri[i]) % R
                coefs[0] = (coefs[0] - prev) % R
            elif not coefs:
                coefs = [(-prev) % R]
            else:
                coefs[0] = (coefs[0] - prev) % R
            prev = ri[i]
            assert int(lens[i]) == len(coefs), i
            assert oc.fr_to_ints(got[i]) == coefs + [0] * (stride - len(coefs)), i
    finally:
        d_t.free(); d_c.free()


def test_eval_pointer_list_two_points(gpu):
    lens = [4099, 4099, 98304, 1, 16387, 5000]
    polys = [rand_fr_wire(ln, 110 + i) for i, ln in enumerate(lens)]
    pts = rand_fr_wire(2, 120)
    idx = [0, 1, 0, 1, 1, 0]
    devs = [Dev(gpu, ln, p) for ln, p in zip(lens, polys)]
    try:
        got = gpu.poly_eval_ptrs_device([d.ptr for d in devs], lens, idx, pts)
        for k in range(len(lens)):
            assert np.array_equal(got[k], oc.poly_eval(polys[k], pts[idx[k]])), k
        # beyond one launch's limits (a 2^19-coefficient polynomial): the per-polynomial path, same values
        big = rand_fr_wire(1 << 19, 130)
        d_big = Dev(gpu, 1 << 19, big)
        got2 = gpu.poly_eval_ptrs_device([devs[0].ptr, d_big.ptr], [lens[0], 1 << 19], [1, 0], pts)
        assert np.array_equal(got2[0], oc.poly_eval(polys[0], pts[1])) and np.array_equal(got2[1], oc.poly_eval(big, pts[0]))
        d_big.free()
    finally:
        for d in devs:
            d.free()


@pytest.mark.parametrize("want_evals", [False, True])
def test_open_quotient_pointer_list(gpu, want_evals):
    """batch_prove's polynomial work (pcs.rs:119-135) over polynomials of different lengths living in different buffers:
    against the oracle's restatement on the zero-padded stack."""
    n = 4096
    lens = [n + 3, n + 3, n, n + 2, 17, n + 3]
    polys = [rand_fr_wire(ln, 140 + i) for i, ln in enumerate(lens)]
    z, alpha = rand_fr_wire(1, 150)[0], rand_fr_wire(1, 151)[0]
    devs = [Dev(gpu, ln, p) for ln, p in zip(lens, polys)]
    cap = n + 8
    d_q = Dev(gpu, cap)
    try:
        ev = gpu.open_quotient_ptrs_device([d.ptr for d in devs], lens, z, alpha, d_q.ptr, cap, want_evals=want_evals)
        gpu.sync()
        stack = np.zeros((len(lens), n + 3, 4), dtype=np.uint64)
        for k, p in enumerate(polys):
            stack[k, : lens[k]] = p
        q, want_ev, rem_zero = oc.open_quotient(stack, z, alpha)
        assert rem_zero
        got = d_q.get()
        assert np.array_equal(got[: n + 3], q) and not got[n + 2:].any()
        if want_evals:
            assert np.array_equal(ev, want_ev)
    finally:
        for d in devs:
            d.free()
        d_q.free()


def test_trimmed_length_is_from_coefs_length(gpu):
    stride = 5000
    polys = np.zeros((4, stride, 4), dtype=np.uint64)
    polys[0, :4099] = rand_fr_wire(4099, 1)                   # full
    polys[1, :4000] = rand_fr_wire(4000, 2); polys[1, 3990:4000] = 0         # trailing zeros inside the given length
    polys[3, 0] = rand_fr_wire(1, 3)[0]                       # a constant; polys[2] is the zero polynomial
    polys[0, 4500] = rand_fr_wire(1, 4)[0]                    # junk beyond the given length is not looked at
    d = Dev(gpu, 4 * stride, polys)
    try:
        got = gpu.poly_trimmed_len_device(d.ptr, stride, [4099, 4000, 4099, 4099])
        assert [int(v) for v in got] == [4099, 3990, 0, 1]
        assert int(gpu.poly_trimmed_len_device(d.ptr, stride, [4600])[0]) == 4501
        # asynchronous form: pinned result words, valid after the next synchronising call; ordinary memory is refused
        import ctypes
        from uzkge_amd import UzkgeError
        h = gpu.host_alloc(4 * 8)
        try:
            view = np.ctypeslib.as_array(ctypes.cast(h, ctypes.POINTER(ctypes.c_uint64)), shape=(4,))
            for rep in range(3):                 # alternating result sets inside the library: repeated calls do not disturb each other
                view[:] = 12345
                gpu.poly_trimmed_len_async_device(d.ptr, stride, [4099, 4000, 4099, 4099], h)
                gpu.sync()
                assert [int(v) for v in view] == [4099, 3990, 0, 1]
            with pytest.raises(UzkgeError):
                gpu.poly_trimmed_len_async_device(d.ptr, stride, [4099], np.zeros(1, dtype=np.uint64).ctypes.data)
        finally:
            gpu.host_free(h)
    finally:
        d.free()
