"""ctypes binding of libcrabml_amd.so (the C ABI in include/crabml_amd.h).

There is no CPU fallback: if the shared library is missing the import fails,
and every compute entry point returns CRAB_ERR_DEVICE without a gfx950 GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRAB_AMD_LIB") or os.path.join(_HERE, "libcrabml_amd.so")   # the override is for same-box A/B timing of two builds

CRAB_OK, CRAB_ERR_ARG, CRAB_ERR_NO_ROOT, CRAB_ERR_EIGEN, CRAB_ERR_DEVICE, CRAB_ERR_STATE = range(6)
CRAB_FLAG_NO_CLAMP = 1
CRAB_FLAG_BEB_PARTIAL, CRAB_FLAG_BEB_GIVEN_LNFX = 2, 4

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
lp = C.POINTER(C.c_int64)
bp = C.POINTER(C.c_uint8)
vp = C.c_void_p

# name -> (restype, argtypes); the complete export list of include/crabml_amd.h
SIGNATURES = {
    "crab_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "crab_create_multi": (C.c_int, [ip, C.c_int, C.POINTER(vp)]),
    "crab_n_devices": (C.c_int, [vp]),
    "crab_set_eigensolver": (C.c_int, [vp, C.c_int]),
    "crab_set_clade_tables": (C.c_int, [vp, C.c_int]),
    "crab_last_prune_launches": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "crab_last_clade_tables": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "crab_destroy": (None, [vp]),
    "crab_last_error": (C.c_char_p, [vp]),
    "crab_device_available": (C.c_int, []),
    "crab_set_tree": (C.c_int, [vp, C.c_int, ip, ip, ip, C.c_int]),
    "crab_set_alignment": (C.c_int, [vp, ip, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "crab_n_patterns": (C.c_int64, [vp]),
    "crab_n_patterns_total": (C.c_int64, [vp]),
    "crab_n_sites": (C.c_int64, [vp]),
    "crab_eval_site_classes": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, C.c_uint, dp, dp]),
    "crab_eval_site_classes_batch": (C.c_int, [vp, C.c_int, C.c_int, dp, dp, dp, dp, C.c_uint, dp]),
    "crab_eval_site_lnl_by_class": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_uint, dp]),
    "crab_eval_beb_grid": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int, C.c_int, ip, dp, C.c_uint, dp, dp, dp]),
    "crab_eval_branch_site": (C.c_int, [vp, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp, bp, C.c_int,
                                        C.c_uint, dp, dp]),
    "crab_eval_branch": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int, C.c_uint, dp, dp]),
    "crab_eval_codon_site_classes": (C.c_int, [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_uint, dp]),
    "crab_eval_codon_class_posteriors": (C.c_int, [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_uint, dp]),
    "crab_eval_codon_branch": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, C.c_uint, dp]),
    "crab_eval_codon_branch_site": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp, dp, dp, bp, C.c_int, C.c_uint, dp]),
    "crab_build_site_class_q_matrices": (C.c_int, [C.c_int, C.c_double, dp, dp, dp, dp]),
    "crab_discretise_omega": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_int, dp]),
    "crab_plan_clade_work": (C.c_int, [C.c_int, ip, ip, ip, C.c_int, ip, C.c_int64, C.c_int, C.POINTER(C.c_int), lp, lp, lp]),
    "crab_discretise_omega_mixture": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, dp]),
    "crab_model_a_quantities": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp]),
    "crab_last_timing": (C.c_int, [vp, dp, dp, dp]),
    "crab_last_work": (C.c_int, [vp, lp, lp]),
    "crab_last_partial_updates": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "crab_last_tree_parts": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "crab_eig_pool_stats": (C.c_int, [vp, lp, lp, lp]),
    "crab_export_transition_matrices": (C.c_int, [vp, C.c_int, dp]),
    "crab_matrix_exponential": (C.c_int, [C.c_int, C.c_int, dp, dp, C.c_double, dp]),
    "crab_matrix_exponential_noalloc": (C.c_int, [C.c_int, C.c_int, dp, dp, C.c_double, dp]),
    "crab_build_codon_q_matrix": (C.c_int, [C.c_double, C.c_double, dp, C.c_int, C.c_double, dp]),
    "crab_compile_tree": (C.c_int, [C.c_int, ip, ip, ip, C.c_int, ip, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crab_compile_path_program": (C.c_int, [C.c_int, ip, ip, ip, C.c_int, C.c_int, ip, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crab_compile_tree_part": (C.c_int, [C.c_int, ip, ip, ip, C.c_int, C.c_int, C.c_int, ip, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crab_compile_clade_tables": (C.c_int, [C.c_int, ip, ip, ip, C.c_int, ip, C.c_int64, C.c_int, C.c_int, ip, C.c_int, C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint32), ip]),
    "crab_compress_patterns": (C.c_int, [ip, C.c_int, C.c_int64, C.c_int, lp, dp, lp, lp]),
    "crab_host_pool_selftest": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "crab_spectral_decompose": (C.c_int, [C.c_int, dp, dp, dp, dp, dp]),
    "crab_compute_log_likelihood": (C.c_int, [C.c_int, C.c_int, dp, dp, C.c_int, ip, ip, dp, ip, C.c_int, C.c_int64, ip, dp]),
    "crab_compute_site_class_log_likelihood": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_int, ip, ip, dp, ip,
                                                         C.c_int, C.c_int64, ip, dp]),
    "crab_compute_site_log_likelihoods_by_class": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, ip, ip, dp, ip,
                                                             C.c_int, C.c_int64, ip, dp]),
    "crab_compute_branch_site_log_likelihood": (C.c_int, [C.c_int, C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double,
                                                          C.c_double, dp, dp, C.c_int, ip, ip, dp, bp, C.c_int, ip, C.c_int,
                                                          C.c_int64, ip, dp]),
    "crab_compute_log_likelihood_branch": (C.c_int, [C.c_int, C.c_int, ip, C.c_int, C.c_int64, ip, C.c_int, ip, dp, C.c_int,
                                                     dp, C.c_int, dp, dp]),
    "crab_clear_cache": (None, []),
}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes. Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). crabml_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class CrabError(ValueError):
    """Raised for every non-zero status; a ValueError like the reference's PyO3 errors (lib.rs: PyValueError)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def check(status, handle=None):
    if status != CRAB_OK:
        msg = load().crab_last_error(handle)
        raise CrabError(status, (msg or b"").decode() or f"crabml_amd error {status}")


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ip)


def u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(bp)
