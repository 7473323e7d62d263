"""Thin numpy-facing wrappers over the C ABI.  Field elements travel as uint64 arrays in the wire
format of include/uzkge_gpu.h (Montgomery, 4 LE limbs): scalars [n,4], affine points [n,8],
Jacobian results [12]."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from .errors import check

lib = N.lib


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need C-contiguous uint64"
    return a.ctypes.data_as(ctypes.c_void_p)


def init(device: int = 0) -> None:
    check(lib.uzk_init(device))


def shutdown() -> None:
    check(lib.uzk_shutdown())


def device_count() -> int:
    return lib.uzk_device_count()


def sync() -> None:
    check(lib.uzk_sync())


def ctx_create() -> int:
    """A new context (stream + workspaces + lock) on the bound device; see uzk_ctx_create."""
    h = ctypes.c_uint64(0)
    check(lib.uzk_ctx_create(ctypes.byref(h)))
    return h.value


def ctx_create_on(device: int) -> int:
    """A new context on HIP device `device` (uzk_ctx_create_on): SRS handles, circuits and provers made while it is current live
    there, so one process can drive several GPUs."""
    h = ctypes.c_uint64(0)
    check(lib.uzk_ctx_create_on(device, ctypes.byref(h)))
    return h.value


def ctx_device(ctx: int = 0) -> int:
    d = ctypes.c_int(0)
    check(lib.uzk_ctx_device(ctx, ctypes.byref(d)))
    return d.value


def ctx_set_current(ctx: int) -> None:
    """Make `ctx` (0 = default) the calling thread's current context."""
    check(lib.uzk_ctx_set_current(ctx))


def ctx_destroy(ctx: int) -> None:
    check(lib.uzk_ctx_destroy(ctx))


def ctx_current() -> int:
    h = ctypes.c_uint64(0)
    check(lib.uzk_ctx_current(ctypes.byref(h)))
    return h.value


def ctx_wait(other: int) -> None:
    """The calling thread's current context waits on the device for everything queued so far on context `other`."""
    check(lib.uzk_ctx_wait(other))


class ShardedSrs:
    """An SRS cut into contiguous point chunks over several devices of ONE process (uzk_srs_register_sharded): `devices` names the
    device of every chunk (an ordinal may repeat).  msm() runs the chunks side by side and folds the partial sums on the host."""

    def __init__(self, points: np.ndarray, devices, window_bits: int = -1):
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
        dev = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_uint64(0)
        check(lib.uzk_srs_register_sharded(_ptr(pts), pts.shape[0], dev, len(devices), window_bits, ctypes.byref(h)))
        self.handle, self.n, self.n_chunks = h.value, pts.shape[0], len(devices)

    def msm(self, scalars: np.ndarray, want_partials: bool = False):
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(12, dtype=np.uint64)
        parts = np.zeros((self.n_chunks, 12), dtype=np.uint64)
        check(lib.uzk_msm_g1_sharded(self.handle, _ptr(s), s.shape[0], _ptr(parts) if want_partials else None, _ptr(out)))
        return (out, parts) if want_partials else out

    def info(self):
        """(n, [(device, lo, hi) per chunk])"""
        n, k = ctypes.c_size_t(0), ctypes.c_uint32(0)
        dev = (ctypes.c_int * self.n_chunks)()
        bounds = (ctypes.c_size_t * (2 * self.n_chunks))()
        check(lib.uzk_srs_sharded_info(self.handle, ctypes.byref(n), ctypes.byref(k), dev, bounds))
        return n.value, [(dev[i], bounds[2 * i], bounds[2 * i + 1]) for i in range(k.value)]

    def release(self) -> None:
        if self.handle:
            check(lib.uzk_srs_release_sharded(self.handle))
            self.handle = 0


class Srs:
    """Device-resident SRS (static bases of KZG commit)."""

    def __init__(self, handle: int, n: int):
        self.handle, self.n = handle, n

    @classmethod
    def from_host(cls, points: np.ndarray) -> "Srs":
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
        h = ctypes.c_uint64(0)
        check(lib.uzk_srs_register(_ptr(pts), pts.shape[0], ctypes.byref(h)))
        return cls(h.value, pts.shape[0])

    @classmethod
    def from_device(cls, d_ptr: int, n: int) -> "Srs":
        h = ctypes.c_uint64(0)
        check(lib.uzk_srs_register_device(ctypes.c_void_p(d_ptr), n, ctypes.byref(h)))
        return cls(h.value, n)

    def precompute(self, window_bits: int = 0) -> None:
        """Build the window table for a static SRS (uzk_srs_precompute)."""
        check(lib.uzk_srs_precompute(self.handle, window_bits))

    def release(self) -> None:
        if self.handle:
            check(lib.uzk_srs_release(self.handle))
            self.handle = 0


def msm(srs: Srs, scalars: np.ndarray, offset: int = 0) -> np.ndarray:
    """sum_i scalars[i] * SRS[offset+i] -> Jacobian [12] (G1Projective::msm, kzg_poly_commitment.rs:290)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    check(lib.uzk_msm_g1(srs.handle, offset, _ptr(s) if s.shape[0] else None, s.shape[0], _ptr(out)))
    return out


def msm_device(srs: Srs, d_scalars: int, n: int, offset: int = 0) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    check(lib.uzk_msm_g1_device(srs.handle, offset, ctypes.c_void_p(d_scalars), n, _ptr(out)))
    return out


def msm_batch(srs: Srs, scalars: np.ndarray, offset: int = 0) -> np.ndarray:
    """scalars [batch, n, 4] against the same bases -> [batch, 12] Jacobian results."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    assert s.ndim == 3 and s.shape[2] == 4
    out = np.zeros((s.shape[0], 12), dtype=np.uint64)
    check(lib.uzk_msm_g1_batch(srs.handle, offset, _ptr(s) if s.size else None, s.shape[1], s.shape[0], _ptr(out)))
    return out


def msm_batch_device(srs: Srs, d_scalars: int, n: int, batch: int, offset: int = 0) -> np.ndarray:
    out = np.zeros((batch, 12), dtype=np.uint64)
    check(lib.uzk_msm_g1_batch_device(srs.handle, offset, ctypes.c_void_p(d_scalars), n, batch, _ptr(out)))
    return out


def msm_raw(points: np.ndarray, scalars: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    if p.shape[0] != s.shape[0]:
        # ark_ec::VariableBaseMSM::msm returns Err(min_len) here; the reference unwraps it
        from .errors import UzkgeError
        raise UzkgeError(N.UZK_ERR_COMMITMENT, f"points ({p.shape[0]}) and scalars ({s.shape[0]}) differ in length")
    out = np.zeros(12, dtype=np.uint64)
    check(lib.uzk_msm_g1_raw(_ptr(p) if p.shape[0] else None, _ptr(s) if s.shape[0] else None, s.shape[0], _ptr(out)))
    return out


def g1_fold(partials: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    check(lib.uzk_g1_fold(_ptr(p), p.shape[0], _ptr(out)))
    return out


def g1_to_affine(jac: np.ndarray) -> np.ndarray:
    j = np.ascontiguousarray(jac, dtype=np.uint64).reshape(12)
    out = np.zeros(8, dtype=np.uint64)
    check(lib.uzk_g1_to_affine(_ptr(j), _ptr(out)))
    return out


def domain_supported(n: int) -> bool:
    return bool(lib.uzk_domain_supported(n))


def domain_group_gen(n: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    check(lib.uzk_domain_group_gen(n, _ptr(out)))
    return out


def ntt(data: np.ndarray, inverse: bool = False, coset_shift: Optional[np.ndarray] = None) -> np.ndarray:
    """EvaluationDomain::fft / ifft over the size-len(data) domain (field_polynomial.rs:583-597);
    returns a new [n,4] array, natural order."""
    a = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4).copy()
    cs = None
    if coset_shift is not None:
        cs = np.ascontiguousarray(coset_shift, dtype=np.uint64).reshape(4)
    check(lib.uzk_ntt_fr(_ptr(a), a.shape[0], int(inverse), _ptr(cs) if cs is not None else None))
    return a


def ntt_inplace(a: np.ndarray, inverse: bool = False, coset_shift: Optional[np.ndarray] = None) -> None:
    """uzk_ntt_fr as the C ABI defines it: `a` ([n,4] u64, C-contiguous) is transformed in place (no copy here)."""
    assert a.dtype == np.uint64 and a.flags.c_contiguous and a.ndim == 2 and a.shape[1] == 4
    cs = None
    if coset_shift is not None:
        cs = np.ascontiguousarray(coset_shift, dtype=np.uint64).reshape(4)
    check(lib.uzk_ntt_fr(_ptr(a), a.shape[0], int(inverse), _ptr(cs) if cs is not None else None))


def ntt_batch(data: np.ndarray, inverse: bool = False, coset_shift: Optional[np.ndarray] = None) -> np.ndarray:
    """`data` [batch, n, 4]: batch independent transforms of size n in one call."""
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    assert a.ndim == 3 and a.shape[2] == 4
    cs = None
    if coset_shift is not None:
        cs = np.ascontiguousarray(coset_shift, dtype=np.uint64).reshape(4)
    check(lib.uzk_ntt_fr_batch(_ptr(a), a.shape[1], a.shape[0], int(inverse), _ptr(cs) if cs is not None else None))
    return a


def ntt_batch_device(d_in: int, d_out: int, n: int, batch: int, inverse: bool = False,
                     coset_shift: Optional[np.ndarray] = None, sync: bool = False) -> None:
    cs = None
    if coset_shift is not None:
        cs = np.ascontiguousarray(coset_shift, dtype=np.uint64).reshape(4)
    check(lib.uzk_ntt_fr_batch_device(ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), n, batch, int(inverse),
                                      _ptr(cs) if cs is not None else None, int(sync)))


def ntt_device(d_in: int, d_out: int, n: int, inverse: bool = False, coset_shift: Optional[np.ndarray] = None,
               sync: bool = False) -> None:
    cs = None
    if coset_shift is not None:
        cs = np.ascontiguousarray(coset_shift, dtype=np.uint64).reshape(4)
    check(lib.uzk_ntt_fr_device(ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), n, int(inverse),
                                _ptr(cs) if cs is not None else None, int(sync)))


def poly_eval_batch(coefs: np.ndarray, x_mont: np.ndarray) -> np.ndarray:
    """coefs [batch, n, 4] -> [batch, 4]: p_b(x) (FpPolynomial::eval for a batch at one point)."""
    c = np.ascontiguousarray(coefs, dtype=np.uint64)
    assert c.ndim == 3 and c.shape[2] == 4
    x = np.ascontiguousarray(x_mont, dtype=np.uint64).reshape(4)
    out = np.zeros((c.shape[0], 4), dtype=np.uint64)
    check(lib.uzk_poly_eval_batch(_ptr(c) if c.size else None, c.shape[1], c.shape[0], _ptr(x), _ptr(out)))
    return out


def poly_eval_batch_device(d_coefs: int, n: int, batch: int, x_mont: np.ndarray) -> np.ndarray:
    """batch device-resident polynomials of n coefficients each (contiguous) evaluated at one point -> [batch, 4]."""
    x = np.ascontiguousarray(x_mont, dtype=np.uint64).reshape(4)
    out = np.zeros((batch, 4), dtype=np.uint64)
    check(lib.uzk_poly_eval_batch_device(ctypes.c_void_p(d_coefs), n, batch, _ptr(x), _ptr(out)))
    return out


def z_poly(w: np.ndarray, perm: np.ndarray, group: np.ndarray, k: np.ndarray, beta: np.ndarray, gamma: np.ndarray) -> np.ndarray:
    """Permutation grand product evaluations (helpers.rs:160-220).  w [n_wires, n, 4], perm [n_wires, n] uint32."""
    wv = np.ascontiguousarray(w, dtype=np.uint64)
    n_wires, n = wv.shape[0], wv.shape[1]
    pm = np.ascontiguousarray(perm, dtype=np.uint32).reshape(n_wires, n)
    g = np.ascontiguousarray(group, dtype=np.uint64).reshape(n, 4)
    kk = np.ascontiguousarray(k, dtype=np.uint64).reshape(n_wires, 4)
    out = np.zeros((n, 4), dtype=np.uint64)
    check(lib.uzk_z_poly(_ptr(wv), pm.ctypes.data_as(ctypes.c_void_p), _ptr(g), _ptr(kk),
                         _ptr(np.ascontiguousarray(beta, dtype=np.uint64).reshape(4)),
                         _ptr(np.ascontiguousarray(gamma, dtype=np.uint64).reshape(4)), n, n_wires, _ptr(out)))
    return out


def z_poly_device(d_w: int, d_perm: int, d_group: int, k: np.ndarray, beta: np.ndarray, gamma: np.ndarray, n: int,
                  n_wires: int, d_z: int) -> None:
    """z_poly on device-resident wires / permutation / domain (helpers.rs:160-220); z -> d_z (n elements)."""
    kk = np.ascontiguousarray(k, dtype=np.uint64).reshape(n_wires, 4)
    check(lib.uzk_z_poly_device(ctypes.c_void_p(d_w), ctypes.c_void_p(d_perm), ctypes.c_void_p(d_group), _ptr(kk),
                                _ptr(np.ascontiguousarray(beta, dtype=np.uint64).reshape(4)),
                                _ptr(np.ascontiguousarray(gamma, dtype=np.uint64).reshape(4)), n, n_wires,
                                ctypes.c_void_p(d_z)))


def open_quotient_device(d_polys: int, n: int, batch: int, z: np.ndarray, alpha: np.ndarray, d_q: int) -> np.ndarray:
    """batch_prove's polynomial work (pcs.rs:119-135): writes q = sum_k alpha^k (p_k - p_k(z)) / (X - z) to d_q
    (n elements, the last one zero) and returns the evaluations p_k(z) [batch, 4]."""
    ev = np.zeros((batch, 4), dtype=np.uint64)
    check(lib.uzk_open_quotient_device(ctypes.c_void_p(d_polys), n, batch,
                                       _ptr(np.ascontiguousarray(z, dtype=np.uint64).reshape(4)),
                                       _ptr(np.ascontiguousarray(alpha, dtype=np.uint64).reshape(4)),
                                       ctypes.c_void_p(d_q), _ptr(ev)))
    return ev


def open_quotient(polys: np.ndarray, z: np.ndarray, alpha: np.ndarray):
    """Host-array form of open_quotient_device: polys [batch, n, 4] -> (q [n, 4] with a trailing zero, evals [batch, 4])."""
    p = np.ascontiguousarray(polys, dtype=np.uint64)
    assert p.ndim == 3 and p.shape[2] == 4
    batch, n = p.shape[0], p.shape[1]
    q = np.zeros((n, 4), dtype=np.uint64)
    ev = np.zeros((batch, 4), dtype=np.uint64)
    check(lib.uzk_open_quotient(_ptr(p), n, batch, _ptr(np.ascontiguousarray(z, dtype=np.uint64).reshape(4)),
                                _ptr(np.ascontiguousarray(alpha, dtype=np.uint64).reshape(4)), _ptr(q), _ptr(ev)))
    return q, ev


def fold_blinds_device(d_coefs: int, length: int, n_fold: int, d_out: int) -> np.ndarray:
    """Fold modulo X^N - 1 on the device (pcs.rs:137-156 / helpers.rs:1366-1383); returns blinds [len - N, 4]."""
    nb = max(0, length - n_fold)
    blinds = np.zeros((max(nb, 1), 4), dtype=np.uint64)
    check(lib.uzk_fold_blinds_device(ctypes.c_void_p(d_coefs), length, n_fold, ctypes.c_void_p(d_out), _ptr(blinds)))
    return blinds[:nb]


def poly_lincomb_device(d_polys, lens, scalars: np.ndarray, d_out: int, out_len: int) -> None:
    """d_out[j] = sum_k scalars[k] * poly_k[j] (r_poly's shape, helpers.rs:681-999); d_polys: device pointers, lens: lengths."""
    cnt = len(d_polys)
    ptrs = (ctypes.c_void_p * cnt)(*[ctypes.c_void_p(p) for p in d_polys])
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(cnt, 4)
    check(lib.uzk_poly_lincomb_device(ptrs, _ptr(ln), _ptr(sc), cnt, ctypes.c_void_p(d_out), out_len))


def hide_polynomial_device(d_coefs: int, length: int, blinds: np.ndarray, zeroing_degree: int) -> None:
    """hide_polynomial (helpers.rs:139-158) on device-resident coefficients."""
    bl = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(-1, 4)
    check(lib.uzk_hide_polynomial_device(ctypes.c_void_p(d_coefs), length, _ptr(bl) if bl.shape[0] else None, bl.shape[0], zeroing_degree))


def t_quotient_device(n: int, factor: int, vec_ptrs, alpha, beta, gamma, k, anemoi_g, anemoi_g_inv, edwards_a,
                      z_h_inv, d_out: int, sync: bool = True) -> None:
    """The quotient evaluations of t_poly (helpers.rs:284-656) on device-resident coset evaluations.
    vec_ptrs: 56 device pointers in UZK_TQ_* slot order; scalars as [4] uint64 Montgomery limbs,
    k [5, 4], z_h_inv [factor, 4]; d_out: n * factor elements."""
    from ._native import QuotientArgs, TQ_NVEC
    assert len(vec_ptrs) == TQ_NVEC
    a = QuotientArgs()
    a.n, a.factor = n, factor
    for i, p in enumerate(vec_ptrs):
        a.vec[i] = p
    def put(dst, src):
        v = np.ascontiguousarray(src, dtype=np.uint64).reshape(4)
        for j in range(4):
            dst[j] = int(v[j])
    put(a.alpha, alpha); put(a.beta, beta); put(a.gamma, gamma)
    put(a.anemoi_g, anemoi_g); put(a.anemoi_g_inv, anemoi_g_inv); put(a.edwards_a, edwards_a)
    kk = np.ascontiguousarray(k, dtype=np.uint64).reshape(5, 4)
    for j in range(5):
        put(a.k[j], kk[j])
    zz = np.ascontiguousarray(z_h_inv, dtype=np.uint64).reshape(-1, 4)
    for j in range(min(16, zz.shape[0])):
        put(a.z_h_inv[j], zz[j])
    check(lib.uzk_t_quotient_device(ctypes.byref(a), ctypes.c_void_p(d_out), int(sync)))


def synth_points_arith(d_points: int, n: int, seed_scalar_mont: np.ndarray) -> None:
    s = np.ascontiguousarray(seed_scalar_mont, dtype=np.uint64).reshape(4)
    check(lib.uzk_synth_points_arith(ctypes.c_void_p(d_points), n, _ptr(s)))


def synth_points_random(d_points: int, n: int, seed: int) -> None:
    check(lib.uzk_synth_points_random(ctypes.c_void_p(d_points), n, seed))


def synth_scalars(d_scalars: int, n: int, seed: int) -> None:
    check(lib.uzk_synth_scalars(ctypes.c_void_p(d_scalars), n, seed))


def synth_scalars_mix(d_scalars: int, n: int, seed: int) -> None:
    """Prover-like scalar mix (50 % zero, 20 % one, 10 % r-1, 10 % < 2^16, 10 % uniform)."""
    check(lib.uzk_synth_scalars_mix(ctypes.c_void_p(d_scalars), n, seed))


def field_op(field: str, op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Device field primitive applied element-wise (KATs: include/uzkge_gpu_test.h, every opcode).  field: 'fq' | 'fr'."""
    x = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    y = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(x)
    check(lib.uzk_test_field_kat(0 if field == "fq" else 1, op, _ptr(x), _ptr(y), _ptr(out), x.shape[0]))
    return out


def field_elementwise(field: str, op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """uzk_field_op_device, the PRODUCT entry point: mul 0, add 1, sub 2, sqr 4, neg 5, from_mont 6, to_mont 7 on host arrays."""
    x = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    y = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(x)
    check(lib.uzk_field_op_device(0 if field == "fq" else 1, op, _ptr(x), _ptr(y), _ptr(out), x.shape[0]))
    return out


def g1_op(op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 8)
    y = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros((x.shape[0], 12), dtype=np.uint64)
    check(lib.uzk_test_g1_kat(op, _ptr(x), _ptr(y), _ptr(out), x.shape[0]))
    return out


def profile_enable(on: bool) -> None:
    check(lib.uzk_profile_enable(int(on)))


def profile_reset() -> None:
    check(lib.uzk_profile_reset())


def profile_get(name: str):
    ms = ctypes.c_double(0)
    cnt = ctypes.c_uint64(0)
    check(lib.uzk_profile_get(name.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
    return ms.value, cnt.value


def profile_table() -> dict:
    buf = ctypes.create_string_buffer(1 << 16)
    check(lib.uzk_profile_dump(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


def set_msm_window_bits(c: int) -> None:
    check(lib.uzk_msm_set_window_bits(c))


def msm_plan_info(n: int) -> Tuple[int, int]:
    """(window bits, windows) a general-mode MSM over n points will use."""
    cb, w = ctypes.c_int(0), ctypes.c_int(0)
    check(lib.uzk_msm_plan_info(n, ctypes.byref(cb), ctypes.byref(w)))
    return cb.value, w.value


def tune(key: str, value: int) -> None:
    check(lib.uzk_tune(key.encode(), value))


# ---- device memory (uzk_dev_*): what a host language uses instead of a HIP binding ----------------------------
COPY_H2D, COPY_D2H, COPY_D2D = 0, 1, 2


def dev_alloc(nbytes: int) -> int:
    p = ctypes.c_void_p(0)
    check(lib.uzk_dev_alloc(nbytes, ctypes.byref(p)))
    return p.value or 0


def dev_free(d_ptr: int) -> None:
    check(lib.uzk_dev_free(ctypes.c_void_p(d_ptr)))


def host_alloc(nbytes: int) -> int:
    """Pinned host memory (uploads from it are asynchronous); returns the address."""
    p = ctypes.c_void_p(0)
    check(lib.uzk_host_alloc(nbytes, ctypes.byref(p)))
    return p.value or 0


def host_free(h_ptr: int) -> None:
    check(lib.uzk_host_free(ctypes.c_void_p(h_ptr)))


def dev_upload(d_dst: int, a: np.ndarray) -> None:
    a = np.ascontiguousarray(a)
    check(lib.uzk_dev_copy(ctypes.c_void_p(d_dst), a.ctypes.data_as(ctypes.c_void_p), a.nbytes, COPY_H2D))


def dev_download(d_src: int, shape, dtype=np.uint64) -> np.ndarray:
    out = np.empty(shape, dtype=dtype)
    check(lib.uzk_dev_copy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_src), out.nbytes, COPY_D2H))
    return out


def dev_copy(d_dst: int, d_src: int, nbytes: int) -> None:
    check(lib.uzk_dev_copy(ctypes.c_void_p(d_dst), ctypes.c_void_p(d_src), nbytes, COPY_D2D))


def dev_copy2d(dst: int, dst_pitch: int, src: int, src_pitch: int, width: int, rows: int, kind: int = COPY_D2D) -> None:
    check(lib.uzk_dev_copy2d(ctypes.c_void_p(dst), dst_pitch, ctypes.c_void_p(src), src_pitch, width, rows, kind))


def dev_memset(d_dst: int, byte: int, nbytes: int) -> None:
    check(lib.uzk_dev_memset(ctypes.c_void_p(d_dst), byte, nbytes))


def dev_memset2d(d_dst: int, pitch: int, byte: int, width: int, rows: int) -> None:
    check(lib.uzk_dev_memset2d(ctypes.c_void_p(d_dst), pitch, byte, width, rows))


# ---- batched / strided / pointer-list forms (one launch per prover step) -----------------------------------------
def _fr4(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(4)


def ntt_batch_strided_device(d_in: int, in_stride: int, d_out: int, out_stride: int, n: int, batch: int, inverse: bool = False,
                             coset_shift: Optional[np.ndarray] = None, sync: bool = False) -> None:
    cs = _fr4(coset_shift) if coset_shift is not None else None
    check(lib.uzk_ntt_fr_batch_strided_device(ctypes.c_void_p(d_in), in_stride, ctypes.c_void_p(d_out), out_stride, n, batch,
                                              int(inverse), _ptr(cs) if cs is not None else None, int(sync)))


def msm_batch_tail_device(srs: Srs, d_scalars: int, stride: int, n: int, batch: int, tail, tail_n: int, offset: int = 0) -> np.ndarray:
    """`tail`: a host array [batch, tail_n, 4] or a device address (int) of batch * tail_n elements."""
    out = np.zeros((batch, 12), dtype=np.uint64)
    if isinstance(tail, (int, np.integer)):
        tp, on_dev = ctypes.c_void_p(int(tail)), 1
    else:
        t = np.ascontiguousarray(tail, dtype=np.uint64).reshape(batch * tail_n, 4) if tail_n else None
        tp, on_dev = (_ptr(t) if t is not None else None), 0
    check(lib.uzk_msm_g1_batch_tail_device(srs.handle, offset, ctypes.c_void_p(d_scalars), stride, n, batch, tp, tail_n, on_dev, _ptr(out)))
    return out


def hide_polynomial_batch_device(d_coefs: int, stride: int, len_in: int, blinds: np.ndarray, zeroing_degree: int) -> None:
    """blinds [count, hiding_degree, 4]."""
    bl = np.ascontiguousarray(blinds, dtype=np.uint64)
    assert bl.ndim == 3 and bl.shape[2] == 4
    check(lib.uzk_hide_polynomial_batch_device(ctypes.c_void_p(d_coefs), stride, len_in, bl.shape[0], _ptr(bl.reshape(-1, 4)), bl.shape[1],
                                               zeroing_degree))


def fold_blinds_batch_device(d_polys: int, in_stride: int, lens, n_fold: int, d_out: int, out_stride: int, d_tail: int, tail_n: int,
                             want_blinds: bool = False):
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    batch = ln.shape[0]
    bl = np.zeros((batch, max(tail_n // 2, 1), 4), dtype=np.uint64) if want_blinds else None
    check(lib.uzk_fold_blinds_batch_device(ctypes.c_void_p(d_polys), in_stride, _ptr(ln), n_fold, batch, ctypes.c_void_p(d_out), out_stride,
                                           ctypes.c_void_p(d_tail), tail_n, _ptr(bl.reshape(-1, 4)) if want_blinds else None))
    return bl[:, : tail_n // 2] if want_blinds else None


def poly_trimmed_len_device(d_polys: int, stride: int, lens) -> np.ndarray:
    """FpPolynomial::from_coefs' trimmed length of device-resident polynomials (1 + the highest non-zero index; 0 if zero)."""
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    out = np.zeros(ln.shape[0], dtype=np.uint64)
    check(lib.uzk_poly_trimmed_len_device(ctypes.c_void_p(d_polys), stride, _ptr(ln), ln.shape[0], _ptr(out), 1))
    return out


def poly_trimmed_len_async_device(d_polys: int, stride: int, lens, h_out: int) -> None:
    """The same without waiting: h_out = address of uzk_host_alloc memory (len(lens) u64), valid after the next synchronising call."""
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    check(lib.uzk_poly_trimmed_len_device(ctypes.c_void_p(d_polys), stride, _ptr(ln), ln.shape[0], ctypes.c_void_p(h_out), 0))


def split_t_device(d_t: int, t_len: int, chunk: int, rands: np.ndarray, d_chunks: int, chunk_stride: int) -> np.ndarray:
    """Returns the chunk lengths (the reference's coefs.len())."""
    r = np.ascontiguousarray(rands, dtype=np.uint64).reshape(-1, 4)
    lens = np.zeros(r.shape[0], dtype=np.uint64)
    check(lib.uzk_split_t_device(ctypes.c_void_p(d_t), t_len, chunk, r.shape[0], _ptr(r), ctypes.c_void_p(d_chunks), chunk_stride, _ptr(lens)))
    return lens


def _ptr_list(d_polys):
    return (ctypes.c_void_p * len(d_polys))(*[ctypes.c_void_p(p) for p in d_polys])


def poly_eval_ptrs_device(d_polys, lens, point_idx, points: np.ndarray) -> np.ndarray:
    cnt = len(d_polys)
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    pi = np.ascontiguousarray(point_idx, dtype=np.uint32)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((cnt, 4), dtype=np.uint64)
    check(lib.uzk_poly_eval_ptrs_device(_ptr_list(d_polys), _ptr(ln), pi.ctypes.data_as(ctypes.c_void_p), cnt, _ptr(pts), pts.shape[0], _ptr(out)))
    return out


def open_quotient_ptrs_device(d_polys, lens, z: np.ndarray, alpha: np.ndarray, d_q: int, q_cap: int, want_evals: bool = False):
    cnt = len(d_polys)
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    ev = np.zeros((cnt, 4), dtype=np.uint64) if want_evals else None
    check(lib.uzk_open_quotient_ptrs_device(_ptr_list(d_polys), _ptr(ln), cnt, _ptr(_fr4(z)), _ptr(_fr4(alpha)), ctypes.c_void_p(d_q), q_cap,
                                            _ptr(ev) if want_evals else None))
    return ev


# ---- circuits and the five prover rounds (uzk_circuit_*, uzk_prover_*, uzk_prove_round1..5) ---------------------------
CS_Q, CS_S, CS_L1, CS_QB, CS_QPRK, CS_COSET_QUOTIENT, CS_QPK, CS_QG, CS_QECC = 0, 9, 14, 15, 16, 20, 21, 33, 45
PB_EVALS, PB_COEFS, PB_COSET, PB_TQ, PB_T, PB_CHUNKS, PB_FOLD, PB_TAIL, PB_Q, PB_R = range(10)


def trimmed_len(coefs: np.ndarray) -> int:
    """coefs.len() after FpPolynomial::from_coefs (field_polynomial.rs:86-90): trailing zero coefficients dropped."""
    nz = np.nonzero(np.any(np.asarray(coefs).reshape(-1, 4) != 0, axis=1))[0]
    return int(nz[-1]) + 1 if nz.size else 0


class Circuit:
    """A circuit resident in HBM (uzk_circuit_create): commit bases, permutation, the 46 (21) polynomials and their coset tables."""

    def __init__(self, n: int, lagrange_bases: np.ndarray, blind_bases: np.ndarray, permutation: np.ndarray, k: np.ndarray, anemoi_g, anemoi_g_inv,
                 edwards_a, polys, shuffle: bool = True, precompute: bool = False, group_gen=None, lens=None, synthetic: bool = False):
        d = N.CircuitDesc()
        d.n, d.shuffle, d.precompute = n, int(shuffle), int(precompute)
        keep = []
        lag = np.ascontiguousarray(lagrange_bases, dtype=np.uint64).reshape(-1, 8)
        bb = np.ascontiguousarray(blind_bases, dtype=np.uint64).reshape(-1, 8)
        perm = np.ascontiguousarray(permutation, dtype=np.uint32).reshape(-1)
        assert lag.shape[0] == n and bb.shape[0] == 6 and perm.shape[0] == 5 * n
        keep += [lag, bb, perm]
        d.lagrange_bases, d.blind_bases, d.permutation = lag.ctypes.data, bb.ctypes.data, perm.ctypes.data

        kk = np.ascontiguousarray(k, dtype=np.uint64).reshape(5, 4)
        for j in range(5):
            for w in range(4):
                d.k[j][w] = int(kk[j, w])
        for name, val in (("anemoi_g", anemoi_g), ("anemoi_g_inv", anemoi_g_inv), ("edwards_a", edwards_a),
                          ("group_gen", domain_group_gen(n) if group_gen is None else group_gen)):
            v = np.ascontiguousarray(val, dtype=np.uint64).reshape(4)
            for w in range(4):
                getattr(d, name)[w] = int(v[w])
        n_slots = N.CIRCUIT_SLOTS if shuffle else CS_QPK
        for s in range(n_slots):
            if polys[s] is None:            # slot 20 (coset_quotient) is normally None: the library builds it
                continue
            a = np.ascontiguousarray(polys[s], dtype=np.uint64).reshape(-1, 4)
            keep.append(a)
            d.polys[s] = a.ctypes.data
            d.poly_lens[s] = trimmed_len(a) if lens is None else int(lens[s])
        h = ctypes.c_uint64(0)
        check(lib.uzk_circuit_create(ctypes.byref(d), ctypes.byref(h)))
        self.handle, self.n, self.shuffle, self.n_slots = h.value, n, shuffle, n_slots
        if synthetic:
            self.truncate_t(True)

    def truncate_t(self, on: bool) -> None:
        """TESTS / TIMING ONLY (uzk_test_circuit_truncate_t): a synthetic circuit no witness satisfies -- round 3 reads t as its
        first 5n - 2 + sum(hiding) coefficients, as tests/chain_oracle.py does."""
        check(lib.uzk_test_circuit_truncate_t(self.handle, int(on)))

    def info(self):
        """(n, evaluations per proof in round 4, r_poly scalars per proof in round 5, device)."""
        a, e, r, dv = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_int(0)
        check(lib.uzk_circuit_info(self.handle, ctypes.byref(a), ctypes.byref(e), ctypes.byref(r), ctypes.byref(dv)))
        return a.value, e.value, r.value, dv.value

    def update_tables(self, first_slot: int, polys, lens=None) -> None:
        arrs = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ln = (ctypes.c_uint64 * len(arrs))(*[trimmed_len(a) if lens is None else int(lens[i]) for i, a in enumerate(arrs)])
        check(lib.uzk_circuit_update_tables(self.handle, first_slot, len(arrs), ptrs, ln))

    def refresh_tables(self, first_slot: int, evals: np.ndarray, want_polys: bool = True, want_coset: bool = False):
        """The refresh / indexer loop on the device: evaluation vectors -> iFFT -> coset FFT -> Lagrange commit, installed in the
        slots.  Returns (commitments [count,12], polys [count,n,4] | None, lens, coset [count,6n,4] | None)."""
        e = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, self.n, 4)
        cnt = e.shape[0]
        polys = np.zeros((cnt, self.n, 4), dtype=np.uint64) if want_polys else None
        coset = np.zeros((cnt, 6 * self.n, 4), dtype=np.uint64) if want_coset else None
        lens = np.zeros(cnt, dtype=np.uint64)
        cms = np.zeros((cnt, 12), dtype=np.uint64)
        check(lib.uzk_circuit_refresh_tables(self.handle, first_slot, cnt, _ptr(e), _ptr(polys) if want_polys else None, _ptr(lens),
                                             _ptr(coset) if want_coset else None, _ptr(cms)))
        return cms, polys, lens, coset

    def table(self, slot: int, coset: bool = False):
        """(device address, element count) of a slot's polynomial or coset table."""
        p, ln = ctypes.c_void_p(0), ctypes.c_uint64(0)
        check(lib.uzk_circuit_table(self.handle, slot, int(coset), ctypes.byref(p), ctypes.byref(ln)))
        return p.value or 0, ln.value

    def release(self) -> None:
        if self.handle:
            check(lib.uzk_circuit_release(self.handle))
            self.handle = 0


def preprocess_tables_device(srs: Srs, evals: np.ndarray, k1=None, want_coset: bool = False, want_commit: bool = True):
    """uzk_preprocess_tables: [count, n, 4] evaluation vectors -> (polys [count, n, 4], lens, coset [count, 6n, 4] | None,
    commitments [count, 12] | None).  Without commitments the SRS handle is not looked at (srs may be None)."""
    e = np.ascontiguousarray(evals, dtype=np.uint64)
    cnt, n = e.shape[0], e.shape[1]
    polys = np.zeros((cnt, n, 4), dtype=np.uint64)
    lens = np.zeros(cnt, dtype=np.uint64)
    coset = np.zeros((cnt, 6 * n, 4), dtype=np.uint64) if want_coset else None
    cms = np.zeros((cnt, 12), dtype=np.uint64) if want_commit else None
    check(lib.uzk_preprocess_tables(srs.handle if srs is not None else 0, n, cnt, _ptr(e), _ptr(_fr4(k1)) if k1 is not None else None, _ptr(polys), _ptr(lens),
                                    _ptr(coset) if want_coset else None, _ptr(cms) if want_commit else None))
    return polys, lens, coset, cms


def coalesce_config(max_lanes: int = 8, gather_wait_us: int = 0, straggler_wait_us: int = 0, groups: int = 0) -> None:
    """uzk_coalesce_config: how provers of one proof made from now on are shared (max_lanes <= 1: not at all; 0 = the defaults)."""
    check(lib.uzk_coalesce_config(max_lanes, gather_wait_us, straggler_wait_us, groups))


def coalesce_stats() -> dict:
    """uzk_coalesce_stats since the last coalesce_config."""
    a = (ctypes.c_uint64 * 16)()
    check(lib.uzk_coalesce_stats(a))
    return dict(rounds=a[0], calls=a[1], widest=a[2], moved_out=a[3], groups=a[4], gap_us_avg=(a[5] / a[6] if a[6] else 0.0),
                gather_us_avg=(a[7] / a[4] if a[4] else 0.0), group_sizes=[a[8 + i] for i in range(8)])


class Prover:
    """The device buffers of `batch` proofs in lockstep (uzk_prover_create) and the five rounds.  shared=True with batch == 1 is
    the library's default kind -- a prover of one proof whose concurrent round calls the library may run together with other
    threads' (uzk_coalesce_config); shared=False (uzk_prover_create_private) owns its lanes and can show its buffers."""

    def __init__(self, n: int, batch: int = 1, shared: bool = True):
        h = ctypes.c_uint64(0)
        check((lib.uzk_prover_create if shared else lib.uzk_prover_create_private)(n, batch, ctypes.byref(h)))
        self.handle, self.n, self.batch = h.value, n, batch

    def round1(self, circuit: Circuit, witness, wsel, pi_index, pi_value, hiding, blinds, on_device: bool = False) -> np.ndarray:
        """witness / wsel: arrays [batch, 5n, 4] / [batch, 3n, 4] (or None), or device addresses with on_device."""
        B, n_first = self.batch, 8 if wsel is not None else 5
        if on_device:
            w_ptr, s_ptr = ctypes.c_void_p(witness), (ctypes.c_void_p(wsel) if wsel is not None else None)
        else:
            w = np.ascontiguousarray(witness, dtype=np.uint64).reshape(B, 5 * self.n, 4)
            s = None if wsel is None else np.ascontiguousarray(wsel, dtype=np.uint64).reshape(B, 3 * self.n, 4)
            w_ptr, s_ptr = _ptr(w), (None if s is None else _ptr(s))
        idx = np.ascontiguousarray(pi_index, dtype=np.uint32).reshape(-1)
        val = np.ascontiguousarray(pi_value, dtype=np.uint64).reshape(B, idx.size, 4) if idx.size else np.zeros((B, 0, 4), dtype=np.uint64)
        hd = np.ascontiguousarray(hiding, dtype=np.uint32).reshape(n_first)
        bl = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(B, n_first, 3, 4)
        out = np.zeros((B * n_first, 12), dtype=np.uint64)
        check(lib.uzk_prove_round1(self.handle, circuit.handle, w_ptr, s_ptr, int(on_device), idx.ctypes.data_as(ctypes.c_void_p) if idx.size else None,
                                   _ptr(val) if idx.size else None, idx.size, hd.ctypes.data_as(ctypes.c_void_p), _ptr(bl), _ptr(out)))
        return out

    def round2(self, beta, gamma, blinds_z) -> np.ndarray:
        B = self.batch
        out = np.zeros((B, 12), dtype=np.uint64)
        check(lib.uzk_prove_round2(self.handle, _ptr(_frs(beta, B)), _ptr(_frs(gamma, B)), _ptr(_frs(blinds_z, 3 * B)), _ptr(out)))
        return out

    def round3(self, alpha, t_rands) -> np.ndarray:
        B = self.batch
        out = np.zeros((5 * B, 12), dtype=np.uint64)
        check(lib.uzk_prove_round3(self.handle, _ptr(_frs(alpha, B)), _ptr(_frs(t_rands, 5 * B)), _ptr(out)))
        return out

    def round4(self, zeta, shuffle: bool = True) -> np.ndarray:
        B, per = self.batch, 19 if shuffle else 15
        out = np.zeros((B * per, 4), dtype=np.uint64)
        check(lib.uzk_prove_round4(self.handle, _ptr(_frs(zeta, B)), _ptr(out), B * per))
        return out

    def round5(self, r_scalars, alpha_zeta, alpha_zeta_omega) -> np.ndarray:
        B = self.batch
        rs = np.ascontiguousarray(r_scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((2 * B, 12), dtype=np.uint64)
        check(lib.uzk_prove_round5(self.handle, _ptr(rs), rs.shape[0], _ptr(_frs(alpha_zeta, B)), _ptr(_frs(alpha_zeta_omega, B)), _ptr(out)))
        return out

    def buffer(self, which: int):
        """(device address, elements per proof) of a prover buffer (PB_*)."""
        p, ln = ctypes.c_void_p(0), ctypes.c_uint64(0)
        check(lib.uzk_prover_buffer(self.handle, which, ctypes.byref(p), ctypes.byref(ln)))
        return p.value or 0, ln.value

    def download(self, which: int, proof: int = 0) -> np.ndarray:
        p, ln = self.buffer(which)
        return dev_download(p + 32 * ln * proof, (ln, 4))

    def destroy(self) -> None:
        if self.handle:
            check(lib.uzk_prover_destroy(self.handle))
            self.handle = 0


def _frs(a, count: int) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(count, 4)
