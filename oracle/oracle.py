"""ctypes front-end of the CPU oracle (oracle/crab_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (crabml_amd/) never imports this.
Function names mirror the crabml_rust entry points (rust/src/lib.rs:468-478).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "crab_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _check(rc, what):
    if rc != 0:
        msg = {1: "invalid argument", 2: "No root node found", 3: "Matrix inversion failed",
               4: "allocation failed", 5: "Eigendecomposition failed"}.get(rc, f"error {rc}")
        raise ValueError(f"{what}: {msg}")


def _tree_args(tree_structure):
    ids = [int(v) for v, _ in tree_structure]
    par = [-1 if p is None else int(p) for _, p in tree_structure]
    return _i(ids), _i(par)


def matrix_exponential(q, pi, t):
    q_, qp = _d(q)
    pi_, pp = _d(pi)
    n = q_.shape[0]
    out = np.empty((n, n))
    _check(lib().orc_matrix_exponential(n, qp, pp, C.c_double(t), out.ctypes.data_as(_dp)), "matrix_exponential")
    return out


def cached_q(q, pi):
    q_, qp = _d(q)
    pi_, pp = _d(pi)
    n = q_.shape[0]
    U, lam, Uinv = np.empty((n, n)), np.empty(n), np.empty((n, n))
    _check(lib().orc_cached_q(n, qp, pp, U.ctypes.data_as(_dp), lam.ctypes.data_as(_dp), Uinv.ctypes.data_as(_dp)), "cached_q")
    return U, lam, Uinv


def compute_log_likelihood(q, pi, tree_structure, branch_lengths, leaf_names, sequences, leaf_node_ids,
                           scaled=False, return_sites=False):
    q_, qp = _d(q)
    pi_, pp = _d(pi)
    (ids, idp), (par, parp) = _tree_args(tree_structure)
    bl, blp = _d(branch_lengths)
    seq, sp = _i(sequences)
    leaf, lp = _i(leaf_node_ids)
    S = seq.shape[1]
    lnL = C.c_double()
    site = np.empty(S)
    _check(lib().orc_log_likelihood(len(pi_), qp, pp, len(ids), idp, parp, blp, sp, len(leaf), S, lp, int(scaled),
                                    C.byref(lnL), site.ctypes.data_as(_dp)), "compute_log_likelihood")
    return (lnL.value, site) if return_sites else lnL.value


def compute_site_class_log_likelihood(q_matrices, proportions, pi, tree_structure, branch_lengths, leaf_names,
                                      sequences, leaf_node_ids, scaled=False, threads=None, return_sites=False):
    Q, Qp = _d(np.stack([np.asarray(q) for q in q_matrices]))
    pr, prp = _d(proportions)
    pi_, pp = _d(pi)
    (ids, idp), (par, parp) = _tree_args(tree_structure)
    bl, blp = _d(branch_lengths)
    seq, sp = _i(sequences)
    leaf, lp = _i(leaf_node_ids)
    S, K = seq.shape[1], Q.shape[0]
    threads = min(K, os.cpu_count() or 1) if threads is None else threads
    lnL = C.c_double()
    site = np.empty(S)
    _check(lib().orc_site_class_log_likelihood(K, len(pi_), Qp, prp, pp, len(ids), idp, parp, blp, sp, len(leaf), S, lp,
                                               int(scaled), int(threads), C.byref(lnL), site.ctypes.data_as(_dp)),
           "compute_site_class_log_likelihood")
    return (lnL.value, site) if return_sites else lnL.value


def compute_site_log_likelihoods_by_class(q_matrices, pi, tree_structure, branch_lengths, leaf_names, sequences,
                                          leaf_node_ids, scaled=False, threads=None):
    Q, Qp = _d(np.stack([np.asarray(q) for q in q_matrices]))
    pi_, pp = _d(pi)
    (ids, idp), (par, parp) = _tree_args(tree_structure)
    bl, blp = _d(branch_lengths)
    seq, sp = _i(sequences)
    leaf, lp = _i(leaf_node_ids)
    S, K = seq.shape[1], Q.shape[0]
    threads = min(K, os.cpu_count() or 1) if threads is None else threads
    out = np.empty((S, K))
    _check(lib().orc_site_log_likelihoods_by_class(K, len(pi_), Qp, pp, len(ids), idp, parp, blp, sp, len(leaf), S, lp,
                                                   int(scaled), int(threads), out.ctypes.data_as(_dp)),
           "compute_site_log_likelihoods_by_class")
    return out


def compute_branch_site_log_likelihood(q_omega0, q_omega1, q_omega2, qfactor_back, qfactor_fore, omega0, omega2,
                                       site_class_freqs, pi, tree_structure, branch_lengths, branch_labels,
                                       leaf_names, sequences, leaf_node_ids, scaled=False, threads=4,
                                       return_sites=False):
    Q, Qp = _d(np.stack([np.asarray(q_omega0), np.asarray(q_omega1), np.asarray(q_omega2)]))
    fr, frp = _d(site_class_freqs)
    pi_, pp = _d(pi)
    (ids, idp), (par, parp) = _tree_args(tree_structure)
    bl, blp = _d(branch_lengths)
    lab = np.ascontiguousarray(branch_labels, dtype=np.uint8)
    seq, sp = _i(sequences)
    leaf, lp = _i(leaf_node_ids)
    S = seq.shape[1]
    lnL = C.c_double()
    site = np.empty(S)
    _check(lib().orc_branch_site_log_likelihood(len(pi_), Qp, C.c_double(qfactor_back), C.c_double(qfactor_fore),
                                                C.c_double(omega0), C.c_double(omega2), frp, pp, len(ids), idp, parp,
                                                blp, lab.ctypes.data_as(_bp), len(lab), sp, len(leaf), S, lp,
                                                int(scaled), int(threads), C.byref(lnL), site.ctypes.data_as(_dp)),
           "compute_branch_site_log_likelihood")
    return (lnL.value, site) if return_sites else lnL.value


def compute_log_likelihood_branch(sequences, parent_indices, leaf_node_ids, Q_matrices, branch_lengths, pi,
                                  scaled=False, threads=None, return_sites=False):
    Q, Qp = _d(np.stack([np.asarray(q) for q in Q_matrices]))
    pi_, pp = _d(pi)
    par, parp = _i(parent_indices)
    bl, blp = _d(branch_lengths)
    seq, sp = _i(sequences)
    leaf, lp = _i(leaf_node_ids)
    S = seq.shape[1]
    threads = (os.cpu_count() or 1) if threads is None else threads
    lnL = C.c_double()
    site = np.empty(S)
    _check(lib().orc_log_likelihood_branch(len(pi_), Q.shape[0], Qp, pp, len(par), parp, blp, len(bl), sp, len(leaf), S,
                                           lp, int(scaled), int(threads), C.byref(lnL), site.ctypes.data_as(_dp)),
           "compute_log_likelihood_branch")
    return (lnL.value, site) if return_sites else lnL.value


def site_lnf_from_transition_matrices(P, pi, tree_structure, sequences, leaf_node_ids, naive=False, threads=None):
    """ln f_{k,s} (no clamp; evaluated with rescaling so it never underflows) from P[K][n_nodes][n][n]."""
    P_, Pp = _d(P)
    if P_.ndim == 3:
        P_ = P_[None]
    pi_, pp = _d(pi)
    (ids, idp), (par, parp) = _tree_args(tree_structure)
    seq, sp = _i(sequences)
    leaf, lp = _i(leaf_node_ids)
    K, S = P_.shape[0], seq.shape[1]
    threads = min(K, os.cpu_count() or 1) if threads is None else threads
    f = np.empty((K, S))
    e = np.empty((K, S), dtype=np.int32)
    _check(lib().orc_site_likelihoods_from_p(K, len(pi_), P_.ctypes.data_as(_dp), pp, len(ids), idp, parp, sp, len(leaf), S, lp,
                                             int(naive), 1, int(threads), f.ctypes.data_as(_dp), e.ctypes.data_as(_ip)),
           "site_lnf_from_transition_matrices")
    with np.errstate(divide="ignore", invalid="ignore"):
        # f <= 0 (a class that forbids the data, up to rounding noise of either sign) is reported as -inf
        return np.where(f > 0, np.log(np.where(f > 0, f, 1.0)) + e * np.log(2.0), -np.inf).T  # [S, K]


def build_codon_q_matrix_py(kappa, omega, pi, normalization_factor=None):
    pi_, pp = _d(pi)
    out = np.empty((61, 61))
    has = normalization_factor is not None
    lib().orc_build_codon_q(C.c_double(kappa), C.c_double(omega), pp, int(has),
                            C.c_double(normalization_factor if has else 0.0), out.ctypes.data_as(_dp))
    return out
