"""crab_matrix_exponential_noalloc (the entry point that allocates nothing) against crab_matrix_exponential.

Through the C ABI, two calls in a row, the second with a smaller matrix: both entry points run the same host
decomposition and the same kernel, so the results must be bit-identical (the bar for this path is 1e-10 relative /
2e-14 absolute on P entries; bit equality is stricter).  Collected last (file name) so that a failure here hides nothing.
"""
import ctypes as C

import numpy as np
import pytest

from crabml_amd import _lib as L

pytestmark = pytest.mark.gpu


def _reversible_q(n, seed):
    rng = np.random.default_rng(seed)
    pi = rng.dirichlet(5.0 * np.ones(n))
    S = rng.random((n, n))
    S = 0.5 * (S + S.T)
    Q = S * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q /= -(pi * np.diag(Q)).sum()
    return np.ascontiguousarray(Q), np.ascontiguousarray(pi)


def _expm(fn, Q, pi, t):
    n = Q.shape[0]
    out = np.full((n, n), np.nan)
    L.check(fn(0, n, Q.ctypes.data_as(L.dp), pi.ctypes.data_as(L.dp), float(t), out.ctypes.data_as(L.dp)))
    return out


def test_noalloc_matrix_exponential_equals_the_allocating_entry_point(gpu_engine_cls):
    lib = L.load()
    # a codon-sized matrix first, then a smaller one (stale entries of the first call must not leak into the second),
    # then the largest size the array holds
    for n, t, seed in ((61, 0.37, 1), (20, 1.9, 2), (63, 0.05, 3), (4, 0.0, 4)):
        Q, pi = _reversible_q(n, seed)
        want = _expm(lib.crab_matrix_exponential, Q, pi, t)
        got = _expm(lib.crab_matrix_exponential_noalloc, Q, pi, t)
        assert np.all(np.isfinite(got))
        assert np.array_equal(got, want), f"n={n}: max |d| = {np.abs(got - want).max():.3e}"
        np.testing.assert_allclose(got.sum(axis=1), 1.0, rtol=0, atol=1e-12)


def test_noalloc_entry_point_allocates_no_device_memory(gpu_engine_cls):
    """After the first call on the device (which makes the runtime load the code object that holds the array), further calls
    leave the device's free memory where it was -- read with hipMemGetInfo through the HIP runtime the library itself uses."""
    lib = L.load()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    Q, pi = _reversible_q(61, 7)
    _expm(lib.crab_matrix_exponential_noalloc, Q, pi, 0.2)   # first call: symbol lookup, code object
    before = free_bytes()
    for n, seed in ((61, 8), (33, 9), (61, 10), (7, 11)):
        Qn, pin = _reversible_q(n, seed)
        _expm(lib.crab_matrix_exponential_noalloc, Qn, pin, 0.3)
        assert free_bytes() == before


def test_noalloc_rejects_sizes_beyond_its_array(gpu_engine_cls):
    lib = L.load()
    Q, pi = _reversible_q(61, 5)
    big = np.zeros((64, 64))
    rc = lib.crab_matrix_exponential_noalloc(0, 64, big.ctypes.data_as(L.dp), np.full(64, 1 / 64).ctypes.data_as(L.dp), 0.1,
                                             big.ctypes.data_as(L.dp))
    assert rc == L.CRAB_ERR_ARG
    assert b"[1, 63]" in lib.crab_last_error(None)
    out = np.zeros((61, 61))
    assert lib.crab_matrix_exponential_noalloc(0, 61, Q.ctypes.data_as(L.dp), pi.ctypes.data_as(L.dp), 0.1, None) == L.CRAB_ERR_ARG
    assert lib.crab_matrix_exponential_noalloc(99, 61, Q.ctypes.data_as(L.dp), pi.ctypes.data_as(L.dp), 0.1,
                                               out.ctypes.data_as(L.dp)) == L.CRAB_ERR_ARG
