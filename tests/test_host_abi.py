"""CPU-side checks of the product: the C-ABI library loads, exports every symbol
include/uzkge_gpu.h declares, its host-only entry points (domain queries, fold, to_affine)
agree with the oracle, and compute entry points fail loudly without a GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import load_srs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    return set(re.findall(r"\b(uzk_[a-z0-9_]+)\s*\(", hdr)) - {"uzk_g1_affine", "uzk_g1_jac"}


def test_every_declared_symbol_is_exported_and_bound():
    """include/uzkge_gpu.h (the drop-in ABI) and include/uzkge_gpu_test.h (test hooks, outside it) against the library's dynamic
    symbol table: every declared function is exported, every exported uzk_* function is declared in exactly one of the two."""
    from uzkge_amd import _native as N
    declared, hooks = _declared("uzkge_gpu.h"), _declared("uzkge_gpu_test.h")
    assert declared and hooks and not (declared & hooks), "header parse failed"
    for name in sorted(declared | hooks):
        assert hasattr(N.lib, name), f"{name} is declared in include/ but not exported"
    assert declared == set(N.PROTOTYPES), (declared ^ set(N.PROTOTYPES))
    assert hooks == set(N.TEST_PROTOTYPES) and all(h.startswith("uzk_test_") for h in hooks), hooks
    assert not any(n.startswith("uzk_test_") for n in declared)
    nm = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("uzk_") and " T " in ln}
    assert exported == declared | hooks, exported ^ (declared | hooks)


def test_the_product_field_entry_point_refuses_the_kat_opcodes():
    """uzk_field_op_device is what the host mirrors use (format conversion, a few O(n) field operations): seven plain operations.
    The known-answer opcodes of the arithmetic cores are test hooks (uzk_test_field_kat) -- not reachable through the product ABI.
    (Argument checks come before the device is touched, so this runs without a GPU.)"""
    from uzkge_amd import _native as N
    a = np.zeros((1, 4), dtype=np.uint64)
    P = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    for op in (3, 8, 9, 10, 20, 24, 27, 28, -1):
        assert N.lib.uzk_field_op_device(1, op, P(a), P(a), P(a), 1) == N.UZK_ERR_PARAMETER, op


def test_domain_queries_match_oracle():
    """uzk_domain_supported = "this library transforms it" (bounded by the largest oracle-checked sizes, the header's
    UZK_NTT_MAX_LOG2 / _MIXED); uzk_domain_group_gen answers for every domain Fr has."""
    from uzkge_amd import backend as b
    hdr = open(os.path.join(ROOT, "include", "uzkge_gpu.h")).read()
    kmax = int(re.search(r"#define UZK_NTT_MAX_LOG2 (\d+)", hdr).group(1))
    kmix = int(re.search(r"#define UZK_NTT_MAX_LOG2_MIXED (\d+)", hdr).group(1))
    assert (kmax, kmix) == (25, 22)         # == the sizes tests/test_gpu_ntt_large.py compares with the oracle
    for n in (1, 2, 3, 4, 48, 1 << 14, 98304, 1 << 22, 1 << kmax, 3 << kmix):
        assert b.domain_supported(n)
        assert np.array_equal(b.domain_group_gen(n), oc.root_of_unity(n))
    for n in (0, 5, 9, 1 << (kmax + 1), 3 << (kmix + 1), 1 << 28, 1 << 29):
        assert not b.domain_supported(n)
    for n in (1 << 28, 3 << 28):
        assert np.array_equal(b.domain_group_gen(n), oc.root_of_unity(n))
    from uzkge_amd import UzkgeError
    with pytest.raises(UzkgeError):
        b.domain_group_gen(1 << 29)


def test_host_fold_and_to_affine_match_oracle():
    from uzkge_amd import backend as b
    wire, pts = load_srs("lagrange-srs-4096.bin")
    one_q = oc.fr_from_ints([1], mod=opy.P)[0]
    parts = np.stack([np.concatenate([wire[i], one_q]) for i in range(8)])
    parts[3, 8:12] = 0                                   # an infinity partial (z == 0)
    want = None
    for i in range(8):
        if i != 3:
            want = opy.g1_add(want, pts[i])
    folded = b.g1_fold(parts)
    assert oc.jac_to_affine_ints(folded) == want
    aff = b.g1_to_affine(folded)
    assert opy.wire_to_affine(aff.tobytes()) == want
    # P + P and P + (-P) through the fold
    assert oc.jac_to_affine_ints(b.g1_fold(np.stack([parts[0], parts[0]]))) == opy.g1_add(pts[0], pts[0])
    neg = oc.points_from_affine([opy.g1_neg(pts[0])])[0]
    assert oc.jac_to_affine_ints(b.g1_fold(np.stack([parts[0], np.concatenate([neg, one_q])]))) is None
    assert not b.g1_to_affine(np.zeros(12, dtype=np.uint64)).any()


def test_compute_fails_loudly_without_gpu():
    from uzkge_amd import UzkgeError, backend as b
    if b.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(UzkgeError) as e:
        b.ntt(np.zeros((4, 4), dtype=np.uint64))
    assert e.value.kind == "DeviceError"
    with pytest.raises(UzkgeError):
        b.msm_raw(np.zeros((1, 8), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))
    with pytest.raises(UzkgeError) as e:
        b.ntt(np.zeros((5, 4), dtype=np.uint64))          # bad domain is reported before the device
    assert e.value.kind == "FFTError"


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle: no source under uzkge_amd/ or
    include/ may import, link, dlopen or name the oracle modules."""
    banned = re.compile(r"oracle_c|bn254_py|bn254_oracle|liboracle|import\s+oracle|from\s+oracle")
    for top in ("uzkge_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".inc", "Makefile")):
                    src = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not banned.search(src), f"{top}/{f} references the oracle"


def test_max_power_of_2_follows_the_reference_loop():
    """pcs.rs:139-145 / helpers.rs:1367-1373 in release arithmetic (usize wraps): degree 0 leaves the loop with 0, and the
    fold then fails with the reference's FFT error instead of an assertion."""
    from uzkge_amd import UzkgeError
    from uzkge_amd.poly_commit import commit_folded_lagrange, max_power_of_2

    def reference_loop(degree):
        mp = degree
        for i in range(degree, -1, -1):
            if (i & ((i - 1) & (2 ** 64 - 1))) == 0:
                mp = i
                break
        return mp

    for d in list(range(0, 70)) + [4095, 4096, 4097, 16386, 98304]:
        assert max_power_of_2(d) == reference_loop(d), d
    with pytest.raises(UzkgeError) as e:
        commit_folded_lagrange(None, None, np.zeros((1, 4), dtype=np.uint64), 0)
    assert e.value.kind == "FFTError"


def test_circuit_and_prover_entry_points_check_their_arguments_before_the_device():
    """uzk_circuit_create / uzk_prover_create / uzk_prove_round*: argument errors are reported as such with or without a GPU (the
    checks run before anything touches the device), unknown handles are ParameterError, and without a GPU a well-formed call fails
    with DeviceError -- there is no CPU fallback behind the round API either."""
    import ctypes
    from uzkge_amd import UzkgeError, _native as N, backend as b
    wire, _ = load_srs("lagrange-srs-4096.bin")
    n = 4096
    perm = np.arange(5 * n, dtype=np.uint32)
    k = oc.fr_from_ints([1, 2, 3, 4, 5])
    one = oc.fr_from_ints([1])[0]
    polys = [oc.fr_from_ints([i + 1, 1]) for i in range(46)]
    args = dict(lagrange_bases=wire, blind_bases=wire[:6], permutation=perm, k=k, anemoi_g=one, anemoi_g_inv=one, edwards_a=one, polys=polys)
    for bad_n in (0, 8, 4097, 1 << 21):                              # not a power of two in 16 .. 2^20
        d = N.CircuitDesc(); d.n = bad_n
        h = ctypes.c_uint64(0)
        assert N.lib.uzk_circuit_create(ctypes.byref(d), ctypes.byref(h)) == N.UZK_ERR_PARAMETER
    with pytest.raises(UzkgeError) as e:                              # a foreign root of unity: FFTError, before any upload
        b.Circuit(n, group_gen=oc.fr_from_ints([7])[0], **args)
    assert e.value.kind == "FFTError"
    bad_perm = perm.copy(); bad_perm[17] = 5 * n
    with pytest.raises(UzkgeError) as e:
        b.Circuit(n, **dict(args, permutation=bad_perm))
    assert e.value.kind == "ParameterError" and "permutation[17]" in str(e.value)
    h = ctypes.c_uint64(0)
    assert N.lib.uzk_prover_create(100, 1, ctypes.byref(h)) == N.UZK_ERR_PARAMETER        # n not a power of two
    assert N.lib.uzk_prover_create(4096, 0, ctypes.byref(h)) == N.UZK_ERR_PARAMETER       # empty batch
    z = np.zeros(64, dtype=np.uint64)
    p = z.ctypes.data_as(ctypes.c_void_p)
    assert N.lib.uzk_prove_round2(12345, p, p, p, p) == N.UZK_ERR_PARAMETER               # unknown prover
    assert N.lib.uzk_prove_round5(12345, p, 43, p, p, p) == N.UZK_ERR_PARAMETER
    assert N.lib.uzk_prove_round4((1 << 62) | 77, p, p, 19) == N.UZK_ERR_PARAMETER        # ... also one shaped like a shared prover's handle
    assert N.lib.uzk_prover_create_private(4096, 65, ctypes.byref(h)) == N.UZK_ERR_PARAMETER   # more lanes than a prover can hold
    assert N.lib.uzk_coalesce_config(65, 50, 0, 0) == N.UZK_ERR_PARAMETER
    assert N.lib.uzk_coalesce_config(8, 0, 0, 0) == N.UZK_OK
    assert N.lib.uzk_circuit_info(999, None, None, None, None) == N.UZK_ERR_PARAMETER
    # several devices behind the ABI: argument errors before anything touches a device
    dev = (ctypes.c_int * 2)(0, 0)
    assert N.lib.uzk_srs_register_sharded(p, 4, dev, 0, -1, ctypes.byref(h)) == N.UZK_ERR_PARAMETER       # no chunk
    assert N.lib.uzk_srs_register_sharded(p, 4, dev, 2, 3, ctypes.byref(h)) == N.UZK_ERR_PARAMETER        # a window width that does not exist
    assert N.lib.uzk_srs_register_sharded(p, 4, None, 2, -1, ctypes.byref(h)) == N.UZK_ERR_PARAMETER
    assert N.lib.uzk_msm_g1_sharded((1 << 61) | 5, p, 4, None, p) == N.UZK_ERR_PARAMETER                  # unknown handle
    assert N.lib.uzk_srs_release_sharded(12345) == N.UZK_ERR_PARAMETER
    d = ctypes.c_int(-7)
    assert N.lib.uzk_ctx_device(999, ctypes.byref(d)) == N.UZK_ERR_PARAMETER
    if b.device_count() == 0:
        assert N.lib.uzk_ctx_create_on(0, ctypes.byref(h)) == N.UZK_ERR_DEVICE
        assert N.lib.uzk_srs_register_sharded(p, 4, dev, 2, -1, ctypes.byref(h)) == N.UZK_ERR_DEVICE
    assert N.lib.uzk_test_circuit_truncate_t(999, 1) == N.UZK_ERR_PARAMETER
    assert N.lib.uzk_circuit_release(999) == N.UZK_ERR_PARAMETER
    assert N.lib.uzk_circuit_update_tables(999, 21, 12, p, p) == N.UZK_ERR_PARAMETER
    if b.device_count() == 0:
        with pytest.raises(UzkgeError) as e:
            b.Circuit(n, **args)
        assert e.value.kind == "DeviceError"
        with pytest.raises(UzkgeError) as e:
            b.Prover(n, 1)
        assert e.value.kind == "DeviceError"


def test_no_cpp_exception_can_leave_an_entry_point():
    """The callers of the C ABI cannot unwind C++ (Rust builds with panic = "abort"; ctypes): every `int uzk_*` definition of the
    library is a function-try-block that ends in uzk::on_exception (message + UZK_ERR_DEVICE)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    defined = set()
    for name in ("api.cpp", "prover.cpp", "coalesce.cpp", "sharded.cpp"):
        src = open(os.path.join(root, "uzkge_amd", "csrc", name)).read()
        for m in re.finditer(r"^int (uzk_[a-z0-9_]+)\([^;{]*\)\s*(try\s*)?\{", src, re.M):
            assert m.group(2), f"{name}: {m.group(1)} is not a function-try-block"
            assert f'return uzk::on_exception("{m.group(1)}");' in src, m.group(1)
            defined.add(m.group(1))
    header = open(os.path.join(root, "include", "uzkge_gpu.h")).read()
    declared = set(re.findall(r"^int (uzk_[a-z0-9_]+)\(", header, re.M))
    assert declared <= defined, sorted(declared - defined)
