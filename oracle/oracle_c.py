"""ctypes loader for the C oracle (oracle/bn254_oracle.c).  TEST INFRASTRUCTURE ONLY:
import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
product path.  All buffers are numpy uint64 arrays in the C-ABI wire format
(Montgomery 4x64 LE limbs; affine = 8 words, Jacobian = 12 words)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# UZK_ORACLE_LIB: another build of the same source (oracle/Makefile `asan`: tests/test_oracle_sanitized.py runs the checks under
# AddressSanitizer / UBSan in a child process)
_LIB = os.environ.get("UZK_ORACLE_LIB") or os.path.join(_HERE, "liboracle_bn254.so")


def build(force: bool = False) -> str:
    if os.environ.get("UZK_ORACLE_LIB"):
        return _LIB
    src = os.path.join(_HERE, "bn254_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_bn254.so"])
    return _LIB


def _load():
    build()
    lib = ctypes.CDLL(_LIB)
    P = ctypes.c_void_p
    sigs = {
        "oracle_fq_mul": [P, P, P], "oracle_fr_mul": [P, P, P], "oracle_fr_add": [P, P, P],
        "oracle_fr_sub": [P, P, P], "oracle_fr_inv": [P, P], "oracle_fq_inv": [P, P],
        "oracle_fr_to_mont": [P, P], "oracle_fr_from_mont": [P, P],
        "oracle_fq_to_mont": [P, P], "oracle_fq_from_mont": [P, P],
        "oracle_g1_add": [P, P, P], "oracle_g1_double": [P, P], "oracle_g1_add_affine": [P, P, P],
        "oracle_g1_to_affine": [P, P], "oracle_g1_mul": [P, P, P],
        "oracle_msm_naive": [P, P, ctypes.c_size_t, P],
        "oracle_msm_pippenger": [P, P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, P],
        "oracle_root_of_unity": [ctypes.c_uint64, P],
        "oracle_mul_var": [P, ctypes.c_uint64, P],
        "oracle_poly_eval": [P, ctypes.c_uint64, P, P],
    }
    for name, args in sigs.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    lib.oracle_z_poly.argtypes = [P, P, P, P, P, P, ctypes.c_uint32, ctypes.c_uint32, P]
    lib.oracle_z_poly.restype = None
    lib.oracle_open_quotient.argtypes = [P, ctypes.c_uint64, ctypes.c_uint32, P, P, P, P]
    lib.oracle_open_quotient.restype = ctypes.c_int
    lib.oracle_t_quotient.argtypes = [P, P]
    lib.oracle_t_quotient.restype = None
    lib.oracle_z_h_inv.argtypes = [P, ctypes.c_uint32, ctypes.c_uint32, P]
    lib.oracle_z_h_inv.restype = None
    lib.oracle_ntt.argtypes = [P, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
    lib.oracle_ntt.restype = ctypes.c_int
    lib.oracle_domain_supported.argtypes = [ctypes.c_uint64]
    lib.oracle_domain_supported.restype = ctypes.c_int
    lib.oracle_g1_is_on_curve.argtypes = [P]
    lib.oracle_g1_is_on_curve.restype = ctypes.c_int
    return lib


lib = _load()


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _binop(fn, a, b):
    out = np.empty(4, dtype=np.uint64)
    fn(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def fq_mul(a, b): return _binop(lib.oracle_fq_mul, a, b)
def fr_mul(a, b): return _binop(lib.oracle_fr_mul, a, b)
def fr_add(a, b): return _binop(lib.oracle_fr_add, a, b)
def fr_sub(a, b): return _binop(lib.oracle_fr_sub, a, b)


def fr_inv(a):
    out = np.empty(4, dtype=np.uint64); lib.oracle_fr_inv(_p(np.ascontiguousarray(a)), _p(out)); return out


def root_of_unity(n: int) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64); lib.oracle_root_of_unity(n, _p(out)); return out


def g1_to_affine(jac: np.ndarray) -> np.ndarray:
    out = np.empty(8, dtype=np.uint64); lib.oracle_g1_to_affine(_p(np.ascontiguousarray(jac)), _p(out)); return out


def g1_add(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    out = np.empty(12, dtype=np.uint64)
    lib.oracle_g1_add(_p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(q)), _p(out)); return out


def g1_mul(p_affine: np.ndarray, scalar_mont: np.ndarray) -> np.ndarray:
    out = np.empty(12, dtype=np.uint64)
    lib.oracle_g1_mul(_p(np.ascontiguousarray(p_affine)), _p(np.ascontiguousarray(scalar_mont)), _p(out)); return out


def msm_naive(points: np.ndarray, scalars: np.ndarray) -> np.ndarray:
    """points [n,8] u64, scalars [n,4] u64 -> Jacobian [12]."""
    n = scalars.shape[0]
    out = np.empty(12, dtype=np.uint64)
    lib.oracle_msm_naive(_p(np.ascontiguousarray(points)), _p(np.ascontiguousarray(scalars)), n, _p(out))
    return out


def msm_pippenger(points: np.ndarray, scalars: np.ndarray, c: int = 0, threads: int = 1) -> np.ndarray:
    n = scalars.shape[0]
    out = np.empty(12, dtype=np.uint64)
    lib.oracle_msm_pippenger(_p(np.ascontiguousarray(points)), _p(np.ascontiguousarray(scalars)), n, c, threads, _p(out))
    return out


def ntt(data: np.ndarray, inverse: bool = False, threads: int = 1) -> np.ndarray:
    """data [n,4] u64 (Montgomery), n = 2^k or 3*2^k; returns a new array."""
    a = np.ascontiguousarray(data).copy()
    rc = lib.oracle_ntt(_p(a), a.shape[0], int(inverse), threads)
    if rc != 0:
        raise ValueError(f"unsupported domain size {a.shape[0]}")
    return a


def mul_var(data: np.ndarray, k_mont: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(data).copy()
    lib.oracle_mul_var(_p(a), a.shape[0], _p(np.ascontiguousarray(k_mont)))
    return a


def poly_eval(coefs: np.ndarray, x_mont: np.ndarray) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    lib.oracle_poly_eval(_p(np.ascontiguousarray(coefs)), coefs.shape[0], _p(np.ascontiguousarray(x_mont)), _p(out))
    return out


def z_poly(w: np.ndarray, perm: np.ndarray, group: np.ndarray, k: np.ndarray, beta: np.ndarray, gamma: np.ndarray) -> np.ndarray:
    """helpers.rs:160-220 on the CPU.  w [n_wires, n, 4], perm [n_wires, n] uint32."""
    wv = np.ascontiguousarray(w, dtype=np.uint64)
    n_wires, n = wv.shape[0], wv.shape[1]
    pm = np.ascontiguousarray(perm, dtype=np.uint32)
    out = np.zeros((n, 4), dtype=np.uint64)
    lib.oracle_z_poly(_p(wv), pm.ctypes.data_as(ctypes.c_void_p), _p(np.ascontiguousarray(group)), _p(np.ascontiguousarray(k)),
                      _p(np.ascontiguousarray(beta)), _p(np.ascontiguousarray(gamma)), n, n_wires, _p(out))
    return out


class _QuotientArgs(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint32), ("factor", ctypes.c_uint32),
        ("vec", ctypes.c_void_p * 56),
        ("alpha", ctypes.c_uint64 * 4), ("beta", ctypes.c_uint64 * 4), ("gamma", ctypes.c_uint64 * 4),
        ("k", (ctypes.c_uint64 * 4) * 5),
        ("anemoi_g", ctypes.c_uint64 * 4), ("anemoi_g_inv", ctypes.c_uint64 * 4), ("edwards_a", ctypes.c_uint64 * 4),
        ("z_h_inv", (ctypes.c_uint64 * 4) * 16),
    ]


def t_quotient(n: int, factor: int, vecs: np.ndarray, alpha, beta, gamma, k, anemoi_g, anemoi_g_inv, edwards_a, z_h_inv) -> np.ndarray:
    """helpers.rs:284-656 on the CPU.  vecs [56, n*factor, 4] in UZK_TQ_* slot order."""
    v = np.ascontiguousarray(vecs, dtype=np.uint64)
    m = n * factor
    assert v.shape == (56, m, 4)
    a = _QuotientArgs()
    a.n, a.factor = n, factor
    for i in range(56):
        a.vec[i] = v[i].ctypes.data
    def put(dst, src):
        s = np.ascontiguousarray(src, dtype=np.uint64).reshape(4)
        for j in range(4):
            dst[j] = int(s[j])
    put(a.alpha, alpha); put(a.beta, beta); put(a.gamma, gamma)
    put(a.anemoi_g, anemoi_g); put(a.anemoi_g_inv, anemoi_g_inv); put(a.edwards_a, edwards_a)
    kk = np.ascontiguousarray(k, dtype=np.uint64).reshape(5, 4)
    for j in range(5):
        put(a.k[j], kk[j])
    zz = np.ascontiguousarray(z_h_inv, dtype=np.uint64).reshape(-1, 4)
    for j in range(zz.shape[0]):
        put(a.z_h_inv[j], zz[j])
    out = np.zeros((m, 4), dtype=np.uint64)
    lib.oracle_t_quotient(ctypes.byref(a), _p(out))
    return out


def open_quotient(polys: np.ndarray, z: np.ndarray, alpha: np.ndarray):
    """pcs.rs:119-135 on the CPU.  polys [batch, n, 4] -> (q [n, 4] with q[n-1] = 0, evals [batch, 4], remainder_is_zero)."""
    p = np.ascontiguousarray(polys, dtype=np.uint64)
    batch, n = p.shape[0], p.shape[1]
    q = np.zeros((n, 4), dtype=np.uint64); ev = np.zeros((batch, 4), dtype=np.uint64)
    ok = lib.oracle_open_quotient(_p(p), n, batch, _p(np.ascontiguousarray(z, dtype=np.uint64).reshape(4)),
                                  _p(np.ascontiguousarray(alpha, dtype=np.uint64).reshape(4)), _p(q), _p(ev))
    return q, ev, bool(ok)


def z_h_inv(k1: np.ndarray, n: int, factor: int) -> np.ndarray:
    """helpers.rs:242-252: 1 / (k1^n g_m^(n i) - 1), i < factor."""
    out = np.zeros((factor, 4), dtype=np.uint64)
    lib.oracle_z_h_inv(_p(np.ascontiguousarray(k1, dtype=np.uint64).reshape(4)), n, factor, _p(out))
    return out


# ---- numpy <-> python-int helpers (wire format) ----
def fr_from_ints(xs, mod=None) -> np.ndarray:
    """canonical ints -> [n,4] Montgomery limbs (Fr unless mod given)."""
    from bn254_py import R, to_mont  # same directory; test infra only
    m = R if mod is None else mod
    out = np.empty((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        v = to_mont(x % m, m)
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def fr_to_ints(a: np.ndarray, mod=None):
    from bn254_py import R, from_mont
    m = R if mod is None else mod
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [from_mont(sum(int(a[i, k]) << (64 * k) for k in range(4)), m) for i in range(a.shape[0])]


def points_from_affine(pts) -> np.ndarray:
    """list of canonical affine tuples / None -> [n,8] wire array."""
    from bn254_py import affine_to_wire
    buf = b"".join(affine_to_wire(p) for p in pts)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).copy()


def jac_to_affine_ints(jac: np.ndarray):
    from bn254_py import jac_wire_to_affine
    return jac_wire_to_affine(np.ascontiguousarray(jac).tobytes())
