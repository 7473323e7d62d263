"""ctypes binding of libuzkge_gpu.so (C ABI: include/uzkge_gpu.h).

The library is the product; this module only loads it and declares the prototypes.  There is no
Python or CPU fallback: if the shared object is missing (not built) the import fails loudly, and
if no gfx950 device is usable every compute call raises `UzkgeError(DEVICE)`.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuzkge_gpu.so")

UZK_OK, UZK_ERR_PARAMETER, UZK_ERR_DEGREE, UZK_ERR_FFT, UZK_ERR_COMMITMENT, UZK_ERR_DEVICE = range(6)

# symbol -> (restype, argtypes); every symbol include/uzkge_gpu.h declares
_P = ctypes.c_void_p
_SZ = ctypes.c_size_t
_U64 = ctypes.c_uint64
_I = ctypes.c_int
PROTOTYPES = {
    "uzk_init": (_I, [_I]),
    "uzk_shutdown": (_I, []),
    "uzk_device_count": (_I, []),
    "uzk_last_error": (ctypes.c_char_p, []),
    "uzk_version": (ctypes.c_char_p, []),
    "uzk_ctx_create": (_I, [ctypes.POINTER(_U64)]),
    "uzk_ctx_create_on": (_I, [_I, ctypes.POINTER(_U64)]),
    "uzk_ctx_device": (_I, [_U64, ctypes.POINTER(_I)]),
    "uzk_ctx_set_current": (_I, [_U64]),
    "uzk_ctx_destroy": (_I, [_U64]),
    "uzk_ctx_wait": (_I, [_U64]),
    "uzk_ctx_current": (_I, [ctypes.POINTER(_U64)]),
    "uzk_dev_alloc": (_I, [_SZ, ctypes.POINTER(_P)]),
    "uzk_dev_free": (_I, [_P]),
    "uzk_host_alloc": (_I, [_SZ, ctypes.POINTER(_P)]),
    "uzk_host_free": (_I, [_P]),
    "uzk_dev_copy": (_I, [_P, _P, _SZ, _I]),
    "uzk_dev_copy2d": (_I, [_P, _SZ, _P, _SZ, _SZ, _SZ, _I]),
    "uzk_dev_memset": (_I, [_P, _I, _SZ]),
    "uzk_dev_memset2d": (_I, [_P, _SZ, _I, _SZ, _SZ]),
    "uzk_srs_register": (_I, [_P, _SZ, ctypes.POINTER(_U64)]),
    "uzk_srs_register_device": (_I, [_P, _SZ, ctypes.POINTER(_U64)]),
    "uzk_srs_release": (_I, [_U64]),
    "uzk_srs_precompute": (_I, [_U64, _I]),
    "uzk_srs_len": (_I, [_U64, ctypes.POINTER(_SZ)]),
    "uzk_srs_register_sharded": (_I, [_P, _SZ, ctypes.POINTER(_I), ctypes.c_uint32, _I, ctypes.POINTER(_U64)]),
    "uzk_srs_release_sharded": (_I, [_U64]),
    "uzk_msm_g1_sharded": (_I, [_U64, _P, _SZ, _P, _P]),
    "uzk_srs_sharded_info": (_I, [_U64, ctypes.POINTER(_SZ), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(_I), ctypes.POINTER(_SZ)]),
    "uzk_msm_g1": (_I, [_U64, _SZ, _P, _SZ, _P]),
    "uzk_msm_g1_device": (_I, [_U64, _SZ, _P, _SZ, _P]),
    "uzk_msm_g1_batch": (_I, [_U64, _SZ, _P, _SZ, ctypes.c_uint32, _P]),
    "uzk_msm_g1_batch_device": (_I, [_U64, _SZ, _P, _SZ, ctypes.c_uint32, _P]),
    "uzk_msm_g1_batch_tail_device": (_I, [_U64, _SZ, _P, _SZ, _SZ, ctypes.c_uint32, _P, ctypes.c_uint32, _I, _P]),
    "uzk_msm_g1_raw": (_I, [_P, _P, _SZ, _P]),
    "uzk_g1_fold": (_I, [_P, _SZ, _P]),
    "uzk_g1_to_affine": (_I, [_P, _P]),
    "uzk_domain_supported": (_I, [_U64]),
    "uzk_domain_group_gen": (_I, [_U64, _P]),
    "uzk_ntt_fr": (_I, [_P, _U64, _I, _P]),
    "uzk_ntt_fr_device": (_I, [_P, _P, _U64, _I, _P, _I]),
    "uzk_ntt_fr_batch": (_I, [_P, _U64, ctypes.c_uint32, _I, _P]),
    "uzk_ntt_fr_batch_device": (_I, [_P, _P, _U64, ctypes.c_uint32, _I, _P, _I]),
    "uzk_ntt_fr_batch_strided_device": (_I, [_P, _U64, _P, _U64, _U64, ctypes.c_uint32, _I, _P, _I]),
    "uzk_hide_polynomial_batch_device": (_I, [_P, _U64, _U64, ctypes.c_uint32, _P, ctypes.c_uint32, _U64]),
    "uzk_fold_blinds_batch_device": (_I, [_P, _U64, _P, _U64, ctypes.c_uint32, _P, _U64, _P, ctypes.c_uint32, _P]),
    "uzk_poly_trimmed_len_device": (_I, [_P, _U64, _P, ctypes.c_uint32, _P, _I]),
    "uzk_split_t_device": (_I, [_P, _U64, _U64, ctypes.c_uint32, _P, _P, _U64, _P]),
    "uzk_poly_eval_ptrs_device": (_I, [_P, _P, _P, ctypes.c_uint32, _P, ctypes.c_uint32, _P]),
    "uzk_open_quotient_ptrs_device": (_I, [_P, _P, ctypes.c_uint32, _P, _P, _P, _U64, _P]),
    "uzk_poly_eval_batch": (_I, [_P, _U64, ctypes.c_uint32, _P, _P]),
    "uzk_poly_eval_batch_device": (_I, [_P, _U64, ctypes.c_uint32, _P, _P]),
    "uzk_z_poly": (_I, [_P, _P, _P, _P, _P, _P, ctypes.c_uint32, ctypes.c_uint32, _P]),
    "uzk_z_poly_device": (_I, [_P, _P, _P, _P, _P, _P, ctypes.c_uint32, ctypes.c_uint32, _P]),
    "uzk_t_quotient_device": (_I, [_P, _P, _I]),
    "uzk_open_quotient_device": (_I, [_P, ctypes.c_uint64, ctypes.c_uint32, _P, _P, _P, _P]),
    "uzk_open_quotient": (_I, [_P, ctypes.c_uint64, ctypes.c_uint32, _P, _P, _P, _P]),
    "uzk_fold_blinds_device": (_I, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    "uzk_poly_lincomb_device": (_I, [_P, _P, _P, ctypes.c_uint32, _P, ctypes.c_uint64]),
    "uzk_hide_polynomial_device": (_I, [_P, ctypes.c_uint64, _P, ctypes.c_uint32, ctypes.c_uint64]),
    "uzk_circuit_create": (_I, [_P, ctypes.POINTER(_U64)]),
    "uzk_circuit_update_tables": (_I, [_U64, ctypes.c_uint32, ctypes.c_uint32, _P, _P]),
    "uzk_circuit_refresh_tables": (_I, [_U64, ctypes.c_uint32, ctypes.c_uint32, _P, _P, _P, _P, _P]),
    "uzk_preprocess_tables": (_I, [_U64, ctypes.c_uint32, ctypes.c_uint32, _P, _P, _P, _P, _P, _P]),
    "uzk_circuit_table": (_I, [_U64, ctypes.c_uint32, _I, ctypes.POINTER(_P), ctypes.POINTER(_U64)]),
    "uzk_circuit_release": (_I, [_U64]),
    "uzk_circuit_info": (_I, [_U64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(_I)]),
    "uzk_prover_create": (_I, [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_U64)]),
    "uzk_prover_create_private": (_I, [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_U64)]),
    "uzk_coalesce_config": (_I, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "uzk_coalesce_stats": (_I, [ctypes.POINTER(_U64)]),
    "uzk_prover_destroy": (_I, [_U64]),
    "uzk_prove_round1": (_I, [_U64, _U64, _P, _P, _I, _P, _P, ctypes.c_uint32, _P, _P, _P]),
    "uzk_prove_round2": (_I, [_U64, _P, _P, _P, _P]),
    "uzk_prove_round3": (_I, [_U64, _P, _P, _P]),
    "uzk_prove_round4": (_I, [_U64, _P, _P, _SZ]),
    "uzk_prove_round5": (_I, [_U64, _P, _SZ, _P, _P, _P]),
    "uzk_prover_buffer": (_I, [_U64, _I, ctypes.POINTER(_P), ctypes.POINTER(_U64)]),
    "uzk_synth_points_arith": (_I, [_P, _SZ, _P]),
    "uzk_synth_points_random": (_I, [_P, _SZ, _U64]),
    "uzk_synth_scalars": (_I, [_P, _SZ, _U64]),
    "uzk_synth_scalars_mix": (_I, [_P, _SZ, _U64]),
    "uzk_field_op_device": (_I, [_I, _I, _P, _P, _P, _SZ]),
    "uzk_profile_enable": (_I, [_I]),
    "uzk_profile_reset": (_I, []),
    "uzk_profile_get": (_I, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_U64)]),
    "uzk_profile_dump": (_I, [ctypes.c_char_p, _SZ]),
    "uzk_sync": (_I, []),
    "uzk_stream": (_P, []),
    "uzk_msm_set_window_bits": (_I, [_I]),
    "uzk_msm_plan_info": (_I, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "uzk_tune": (_I, [ctypes.c_char_p, _I]),
}


# include/uzkge_gpu_test.h: test hooks exported by the same library (known-answer entry points of the arithmetic cores, the
# synthetic-circuit switch).  Not part of the drop-in ABI; bound here for tests/ and tools/ only.
TEST_PROTOTYPES = {
    "uzk_test_field_kat": (_I, [_I, _I, _P, _P, _P, _SZ]),
    "uzk_test_g1_kat": (_I, [_I, _P, _P, _P, _SZ]),
    "uzk_test_circuit_truncate_t": (_I, [_U64, _I]),
}


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.  Two HIP runtimes in one process do not share
    devices ("No HIP GPUs are available" in whichever initialises second), so when torch is installed
    but not yet imported, load ITS runtime first: libuzkge_gpu.so then binds to the same copy torch
    will use, whatever the import order.  Opt out with UZK_USE_SYSTEM_HIP=1."""
    import sys
    if os.environ.get("UZK_USE_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:   # plumbing only: fall back to the system runtime
        pass


def load() -> ctypes.CDLL:
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C uzkge_amd/csrc` (hipcc, gfx950). The MI355X backend has no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in list(PROTOTYPES.items()) + list(TEST_PROTOTYPES.items()):
        fn = getattr(lib, name)   # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


TQ_NVEC = 56


class QuotientArgs(ctypes.Structure):
    """uzk_quotient_args (include/uzkge_gpu.h)."""
    _fields_ = [
        ("n", ctypes.c_uint32), ("factor", ctypes.c_uint32),
        ("vec", ctypes.c_void_p * TQ_NVEC),
        ("alpha", ctypes.c_uint64 * 4), ("beta", ctypes.c_uint64 * 4), ("gamma", ctypes.c_uint64 * 4),
        ("k", (ctypes.c_uint64 * 4) * 5),
        ("anemoi_g", ctypes.c_uint64 * 4), ("anemoi_g_inv", ctypes.c_uint64 * 4), ("edwards_a", ctypes.c_uint64 * 4),
        ("z_h_inv", (ctypes.c_uint64 * 4) * 16),
    ]


CIRCUIT_SLOTS = 46


class CircuitDesc(ctypes.Structure):
    """uzk_circuit_desc (include/uzkge_gpu.h)."""
    _fields_ = [
        ("n", ctypes.c_uint32), ("shuffle", ctypes.c_uint32), ("precompute", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
        ("lagrange_bases", ctypes.c_void_p), ("blind_bases", ctypes.c_void_p), ("permutation", ctypes.c_void_p),
        ("k", (ctypes.c_uint64 * 4) * 5),
        ("anemoi_g", ctypes.c_uint64 * 4), ("anemoi_g_inv", ctypes.c_uint64 * 4), ("edwards_a", ctypes.c_uint64 * 4),
        ("group_gen", ctypes.c_uint64 * 4),
        ("polys", ctypes.c_void_p * CIRCUIT_SLOTS),
        ("poly_lens", ctypes.c_uint64 * CIRCUIT_SLOTS),
    ]
