"""Drop-in for the PyO3 extension module ``crabml.crabml_rust``.

Same seven function names, argument names/orders and return types as
rust/src/lib.rs:468-478, served by libcrabml_amd.so through ctypes:

    import sys, crabml_amd.crabml_rust as shim
    sys.modules["crabml.crabml_rust"] = shim        # before crabml is imported

after which unmodified crabML (optimize_model, positive_selection, the branch /
branch-site optimisers, BEB) evaluates every likelihood on the MI355X.
The boundary stays stateless for the caller; behind it the C library memoises one
engine per (alignment contents, tree), so the alignment is uploaded once.

Errors: every failure raises ValueError (the reference raises ValueError for its
``Err`` paths, lib.rs:40,72,114...; its Rust panics -- wrong Q count, bad label --
surface as PanicException there and as ValueError here).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .engine import tree_arrays

__all__ = ["matrix_exponential", "compute_log_likelihood", "compute_site_class_log_likelihood",
           "compute_site_log_likelihoods_by_class", "compute_branch_site_log_likelihood",
           "compute_log_likelihood_branch", "build_codon_q_matrix_py"]


def _device() -> int:
    """CRAB_DEVICES=0,1,...: every listed GPU behind one handle (-1 tells the C library to read the list);
    else CRAB_DEVICE or LOCAL_RANK; else device 0."""
    if os.environ.get("CRAB_DEVICES", "").strip():
        return -1
    for var in ("CRAB_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v.strip().lstrip("-").isdigit():
            return int(v)
    return 0


def _single_device() -> int:
    """For the calls that use one GPU whatever the list says (matrix_exponential): the first of CRAB_DEVICES."""
    first = os.environ.get("CRAB_DEVICES", "").split(",")[0].strip()
    return int(first) if first.lstrip("-").isdigit() else max(_device(), 0)


def _square(q, what="q"):
    a = np.ascontiguousarray(q, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError(f"Q matrix must be square, got {'x'.join(map(str, a.shape))}")
    return a


def _pi_for(pi, n):
    p = np.ascontiguousarray(pi, dtype=np.float64)
    if p.ndim != 1 or p.shape[0] != n:
        raise ValueError(f"pi must have length {n}, got {p.shape[0] if p.ndim == 1 else p.shape}")
    return p


def _stack(q_matrices):
    mats = [_square(q) for q in q_matrices]
    if not mats:
        raise ValueError("q_matrices is empty")
    n = mats[0].shape[0]
    for k, m in enumerate(mats):
        if m.shape[0] != n:
            raise ValueError(f"Q matrix {k} has shape {m.shape}, expected ({n}, {n})")
    return np.ascontiguousarray(np.stack(mats)), n


def _seq(sequences):
    s = np.ascontiguousarray(sequences, dtype=np.int32)
    if s.ndim != 2:
        raise ValueError("sequences must be a 2-D int32 array [n_leaves, n_sites]")
    return s


def _leaf_ids(leaf_node_ids, n_rows):
    """One node id per alignment row: the C library reads exactly n_rows of them."""
    leaf = np.ascontiguousarray(leaf_node_ids, dtype=np.int32).reshape(-1)
    if len(leaf) != n_rows:
        raise ValueError(f"leaf_node_ids has {len(leaf)} entries but sequences has {n_rows} rows")
    return leaf


def _branch_lengths(branch_lengths, n_nodes):
    bl = np.ascontiguousarray(branch_lengths, dtype=np.float64).reshape(-1)
    if len(bl) < n_nodes:
        raise ValueError(f"branch_lengths has {len(bl)} entries for {n_nodes} nodes")
    return bl


def matrix_exponential(q, pi, t):
    """exp(Q t) via the cached-eigensystem route (lib.rs:61-79)."""
    Q = _square(q)
    n = Q.shape[0]
    p = _pi_for(pi, n)
    out = np.empty((n, n))
    L.check(L.load().crab_matrix_exponential(_single_device(), n, Q.ctypes.data_as(L.dp), p.ctypes.data_as(L.dp), float(t),
                                             out.ctypes.data_as(L.dp)))
    return out


def compute_log_likelihood(q, pi, tree_structure, branch_lengths, leaf_names, sequences, leaf_node_ids):
    """Single-Q (M0) log-likelihood (lib.rs:97-127)."""
    Q = _square(q)
    n = Q.shape[0]
    p = _pi_for(pi, n)
    ids, par = tree_arrays(tree_structure)
    bl = _branch_lengths(branch_lengths, len(ids))
    seq = _seq(sequences)
    leaf = _leaf_ids(leaf_node_ids, seq.shape[0])
    out = C.c_double()
    L.check(L.load().crab_compute_log_likelihood(_device(), n, Q.ctypes.data_as(L.dp), p.ctypes.data_as(L.dp), len(ids),
                                                 ids.ctypes.data_as(L.ip), par.ctypes.data_as(L.ip), bl.ctypes.data_as(L.dp),
                                                 seq.ctypes.data_as(L.ip), seq.shape[0], seq.shape[1],
                                                 leaf.ctypes.data_as(L.ip), C.byref(out)))
    return out.value


def compute_site_class_log_likelihood(q_matrices, proportions, pi, tree_structure, branch_lengths, leaf_names, sequences,
                                      leaf_node_ids):
    """Site-class mixture log-likelihood, M1a...M9 (lib.rs:146-191)."""
    Q, n = _stack(q_matrices)
    pr = np.ascontiguousarray(proportions, dtype=np.float64)
    if len(pr) < Q.shape[0]:
        raise ValueError(f"Number of proportions ({len(pr)}) must match number of Q matrices ({Q.shape[0]})")
    p = _pi_for(pi, n)
    ids, par = tree_arrays(tree_structure)
    bl = _branch_lengths(branch_lengths, len(ids))
    seq = _seq(sequences)
    leaf = _leaf_ids(leaf_node_ids, seq.shape[0])
    out = C.c_double()
    L.check(L.load().crab_compute_site_class_log_likelihood(
        _device(), Q.shape[0], n, Q.ctypes.data_as(L.dp), pr.ctypes.data_as(L.dp), p.ctypes.data_as(L.dp), len(ids),
        ids.ctypes.data_as(L.ip), par.ctypes.data_as(L.ip), bl.ctypes.data_as(L.dp), seq.ctypes.data_as(L.ip), seq.shape[0],
        seq.shape[1], leaf.ctypes.data_as(L.ip), C.byref(out)))
    return out.value


def compute_site_log_likelihoods_by_class(q_matrices, pi, tree_structure, branch_lengths, leaf_names, sequences,
                                          leaf_node_ids):
    """[n_sites, n_classes] array of ln max(f_{k,s}, 1e-300) for BEB (lib.rs:214-258)."""
    Q, n = _stack(q_matrices)
    p = _pi_for(pi, n)
    ids, par = tree_arrays(tree_structure)
    bl = _branch_lengths(branch_lengths, len(ids))
    seq = _seq(sequences)
    leaf = _leaf_ids(leaf_node_ids, seq.shape[0])
    out = np.empty((seq.shape[1], Q.shape[0]))
    L.check(L.load().crab_compute_site_log_likelihoods_by_class(
        _device(), Q.shape[0], n, Q.ctypes.data_as(L.dp), p.ctypes.data_as(L.dp), len(ids), ids.ctypes.data_as(L.ip),
        par.ctypes.data_as(L.ip), bl.ctypes.data_as(L.dp), seq.ctypes.data_as(L.ip), seq.shape[0], seq.shape[1],
        leaf.ctypes.data_as(L.ip), out.ctypes.data_as(L.dp)))
    return out


def compute_branch_site_log_likelihood(q_omega0, q_omega1, q_omega2, qfactor_back, qfactor_fore, omega0, omega2,
                                       site_class_freqs, pi, tree_structure, branch_lengths, branch_labels, leaf_names,
                                       sequences, leaf_node_ids):
    """Branch-site Model A log-likelihood (lib.rs:287-345)."""
    Q0, Q1, Q2 = _square(q_omega0), _square(q_omega1), _square(q_omega2)
    n = Q0.shape[0]
    if Q1.shape[0] != n or Q2.shape[0] != n:
        raise ValueError("the three Q matrices must have the same shape")
    fr = np.ascontiguousarray(site_class_freqs, dtype=np.float64)
    if fr.shape != (4,):
        raise ValueError("site_class_freqs must have exactly 4 entries")
    p = _pi_for(pi, n)
    ids, par = tree_arrays(tree_structure)
    bl = _branch_lengths(branch_lengths, len(ids))
    lab = np.ascontiguousarray(branch_labels, dtype=np.uint8)
    seq = _seq(sequences)
    leaf = _leaf_ids(leaf_node_ids, seq.shape[0])
    out = C.c_double()
    L.check(L.load().crab_compute_branch_site_log_likelihood(
        _device(), n, Q0.ctypes.data_as(L.dp), Q1.ctypes.data_as(L.dp), Q2.ctypes.data_as(L.dp), float(qfactor_back),
        float(qfactor_fore), float(omega0), float(omega2), fr.ctypes.data_as(L.dp), p.ctypes.data_as(L.dp), len(ids),
        ids.ctypes.data_as(L.ip), par.ctypes.data_as(L.ip), bl.ctypes.data_as(L.dp), lab.ctypes.data_as(L.bp), len(lab),
        seq.ctypes.data_as(L.ip), seq.shape[0], seq.shape[1], leaf.ctypes.data_as(L.ip), C.byref(out)))
    return out.value


def compute_log_likelihood_branch(sequences, parent_indices, leaf_node_ids, Q_matrices, branch_lengths, pi):
    """Branch (free-ratio / multi-ratio) model log-likelihood, one Q per node (lib.rs:363-430)."""
    Q, n = _stack(Q_matrices)
    p = _pi_for(pi, n)
    par = np.ascontiguousarray([-1 if int(x) < 0 else int(x) for x in parent_indices], dtype=np.int32)
    bl = np.ascontiguousarray(branch_lengths, dtype=np.float64).reshape(-1)
    seq = _seq(sequences)
    leaf = _leaf_ids(leaf_node_ids, seq.shape[0])
    out = C.c_double()
    L.check(L.load().crab_compute_log_likelihood_branch(
        _device(), n, seq.ctypes.data_as(L.ip), seq.shape[0], seq.shape[1], par.ctypes.data_as(L.ip), len(par),
        leaf.ctypes.data_as(L.ip), Q.ctypes.data_as(L.dp), Q.shape[0], bl.ctypes.data_as(L.dp), len(bl),
        p.ctypes.data_as(L.dp), C.byref(out)))
    return out.value


def build_codon_q_matrix_py(kappa, omega, pi, normalization_factor=None):
    """61x61 codon rate matrix (lib.rs:445-463 -> qmatrix.rs:22-80). Host arithmetic."""
    p = _pi_for(pi, 61)
    out = np.empty((61, 61))
    has = normalization_factor is not None
    L.check(L.load().crab_build_codon_q_matrix(float(kappa), float(omega), p.ctypes.data_as(L.dp), int(has),
                                               float(normalization_factor) if has else 0.0, out.ctypes.data_as(L.dp)))
    return out


def clear_cache():
    L.load().crab_clear_cache()
