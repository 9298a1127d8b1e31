"""BLAS-backed CPU baseline: the reference's algorithm with its per-edge product done by dgemm.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): used by bench.py's cpu_baseline leg and by
tests/test_oracle.py.  The product (crabml_amd/) never imports this.

crabML's CPU path spends its time in OpenBLAS dgemm: compute_internal_node multiplies the child's [n_sites, 61]
conditional likelihoods by P^T with ndarray's BLAS-backed `dot` and multiplies the result into the node's row block
(rust/src/likelihood.rs:176-195), one Rayon task per site class (likelihood.rs:267), each with a fresh
[n_nodes, n_sites, 61] workspace (likelihood.rs:97-106) and one-hot / all-ones leaf rows (likelihood.rs:120-132).
The C oracle restates that with a hand-written loop; this module restates it with the same library call the reference
makes -- numpy's `@` on float64 is dgemm from the OpenBLAS numpy ships (vendor string reported) -- one Python thread per
class with BLAS pinned to one thread per call (numpy releases the GIL inside dgemm), which is the reference's
parallelism.  Unscaled, clamps as the reference's.  P(t) comes from numpy's eigh + inv (matrix.rs:41-67,88-102).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

LN_TINY = float(np.log(1e-300))


def blas_vendor() -> str:
    try:
        from threadpoolctl import threadpool_info
        for lib in threadpool_info():
            if lib.get("user_api") == "blas":
                return f"{lib.get('internal_api')} {lib.get('version')} ({lib.get('architecture', '?')}, {lib.get('threading_layer', '?')})"
    except Exception:
        pass
    return "unknown BLAS"


def transition_matrices(Q, pi, branch_lengths):
    """[n_nodes, n, n]: P = (U e^{lam t}) U^-1 with U from eigh of the pi-symmetrised matrix and U^-1 by LU inverse."""
    sq = np.sqrt(pi)
    S = Q * sq[:, None] / sq[None, :]
    lam, V = np.linalg.eigh(S, UPLO="L")
    U = V / sq[:, None]
    Ui = np.linalg.inv(U)
    return np.stack([(U * np.exp(lam * t)[None, :]) @ Ui for t in branch_lengths])


def _site_likelihoods(P, pi, parent, order, kids, codes):
    """f_s for one class: dense one-hot leaves, one dgemm per child edge, Hadamard product per node."""
    n_nodes, n = len(parent), len(pi)
    n_leaves, S = codes.shape
    L = [None] * n_nodes
    for r in range(n_leaves):
        row = np.zeros((S, n))
        c = codes[r]
        ok = (c >= 0) & (c < n)
        row[np.flatnonzero(ok), c[ok]] = 1.0
        row[~ok] = 1.0
        L[r] = row
    for v in order:
        if L[v] is not None:
            continue
        acc = np.ones((S, n))
        for c in kids[v]:
            acc *= L[c] @ P[c].T            # dgemm M = S, N = K = 61 (likelihood.rs:189), then zip_mut_with (:192-193)
            L[c] = None                     # (the reference keeps its whole workspace; freeing only bounds our memory)
        L[v] = acc
    return L[order[-1]] @ pi


def site_class_log_likelihood(Q, proportions, pi, parent, branch_lengths, codes, threads=None, return_sites=False):
    """compute_site_class_log_likelihood (likelihood.rs:256-292,349-370) with BLAS products."""
    from threadpoolctl import threadpool_limits
    Q = np.asarray(Q, dtype=np.float64)
    pi = np.asarray(pi, dtype=np.float64)
    parent = np.asarray(parent)
    codes = np.ascontiguousarray(codes)
    K, n_nodes = len(Q), len(parent)
    kids = [[] for _ in range(n_nodes)]
    root = -1
    for v, p in enumerate(parent):
        if p >= 0:
            kids[p].append(v)
        else:
            root = v
    order, stack = [], [(root, 0)]
    while stack:                                         # iterative post-order
        v, i = stack.pop()
        if i < len(kids[v]):
            stack.append((v, i + 1))
            stack.append((kids[v][i], 0))
        else:
            order.append(v)
    bl = np.asarray(branch_lengths, dtype=np.float64)
    threads = min(K, os.cpu_count() or 1) if threads is None else threads

    def one(k):
        P = transition_matrices(Q[k], pi, bl)
        return _site_likelihoods(P, pi, parent, order, kids, codes)

    with threadpool_limits(limits=1, user_api="blas"):
        if threads <= 1:
            f = [one(k) for k in range(K)]
        else:
            with ThreadPoolExecutor(max_workers=threads) as ex:
                f = list(ex.map(one, range(K)))
    f = np.stack(f, axis=1)                              # [S, K]
    with np.errstate(divide="ignore"):
        x = np.log(np.maximum(f, 1e-300)) + np.log(np.asarray(proportions, dtype=np.float64))[None, :]
    mx = x.max(axis=1)
    site = mx + np.log(np.exp(x - mx[:, None]).sum(axis=1))
    lnL = float(site.sum())
    return (lnL, site) if return_sites else lnL
