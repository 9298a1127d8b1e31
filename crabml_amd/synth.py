"""Seeded synthetic trees and codon alignments for benchmarks and large parity tests.

The reference's simulators (src/crabml/simulate/) never touch the native core and are out
of scope; this is a compact generator of our own, following SURVEY.md section 8(d): random binary
topology by random joins with a trifurcating root (bushy, never a caterpillar), branch lengths
Exp(mean), pi ~ Dirichlet(5), kappa = 2.5, sequences evolved down the tree under the model that
is then evaluated, 1 % gap codons.  Output is already in boundary numbering: leaves are nodes
0..N-1 (row i of ``codes`` belongs to node i), internal nodes follow children-before-parents,
the root is last.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import models
from .io import GAP_CODE


def random_tree(n_taxa: int, rng, mean_branch: float):
    """parent[v] (-1 root), branch_lengths[v] (0 at root); trifurcating root; needs n_taxa >= 3."""
    if n_taxa < 3:
        raise ValueError("n_taxa must be >= 3")
    n_nodes = 2 * n_taxa - 2
    parent = np.full(n_nodes, -1, dtype=np.int32)
    active = list(range(n_taxa))
    nxt = n_taxa
    while len(active) > 3:
        i, j = rng.choice(len(active), size=2, replace=False)
        a, b = active[i], active[j]
        parent[a] = parent[b] = nxt
        active = [x for k, x in enumerate(active) if k != i and k != j] + [nxt]
        nxt += 1
    root = nxt
    for a in active:
        parent[a] = root
    assert root == n_nodes - 1
    bl = rng.exponential(mean_branch, size=n_nodes)
    bl = np.maximum(bl, 1e-4)
    bl[root] = 0.0
    return parent, bl


def _transition_matrix(Q, pi, t, eig_cache=None):
    sq = np.sqrt(pi)
    key = (Q.__array_interface__["data"][0], Q.shape)   # the class matrices are views of one stack: same memory, same matrix
    if eig_cache is not None and key in eig_cache:
        lam, V = eig_cache[key]
    else:
        S = Q * sq[:, None] / sq[None, :]
        S = 0.5 * (S + S.T)
        lam, V = np.linalg.eigh(S)
        if eig_cache is not None:
            eig_cache[key] = (lam, V)
    P = (V * np.exp(lam * t)) @ V.T
    P = P * sq[None, :] / sq[:, None]
    P = np.clip(P, 0.0, None)
    return P / P.sum(axis=1, keepdims=True)


def _sample_children(P, parent_states, rng):
    """Child state per site: inverse-CDF draw from row parent_state of P (the number of cumulative sums below u)."""
    cum = np.cumsum(P, axis=1)
    cum[:, -1] = 1.0
    u = rng.random(parent_states.shape[0])
    out = np.empty_like(parent_states)
    order = np.argsort(parent_states, kind="stable")          # sites grouped by parent state: one searchsorted per row of P
    bounds = np.searchsorted(parent_states[order], np.arange(P.shape[0] + 1))
    for i in range(P.shape[0]):
        idx = order[bounds[i]:bounds[i + 1]]
        if len(idx):
            out[idx] = np.searchsorted(cum[i], u[idx], side="left")
    return np.minimum(out, P.shape[0] - 1)


def evolve(parent, bl, pi, site_Q_of_node, site_class, n_sites, rng, gap_fraction=0.01):
    """site_Q_of_node(v, k) -> (matrix id, rate matrix, effective time) for the branch above v and class k."""
    n_nodes = len(parent)
    n_taxa = (n_nodes + 2) // 2
    root = n_nodes - 1
    states = np.zeros((n_nodes, n_sites), dtype=np.int64)
    states[root] = rng.choice(61, size=n_sites, p=pi)
    classes = np.unique(site_class)
    eig_cache = {}   # one decomposition per rate matrix, not per (branch, class): 500 taxa x 50 000 codons in seconds
    class_sites = {int(k): np.flatnonzero(site_class == k) for k in classes}
    for v in range(n_nodes - 2, -1, -1):  # parents have larger ids than children
        p = parent[v]
        cache = {}  # classes may share a matrix on one branch (Model A)
        for k, idx in class_sites.items():
            qid, Q, teff = site_Q_of_node(v, k)
            if qid not in cache:
                cache[qid] = _transition_matrix(Q, pi, teff, eig_cache)
            states[v, idx] = _sample_children(cache[qid], states[p, idx], rng)
    codes = states[:n_taxa].astype(np.int32)
    if gap_fraction > 0:
        mask = rng.random(codes.shape) < gap_fraction
        codes[mask] = GAP_CODE
    return np.ascontiguousarray(codes)


def make_case(model: str, n_taxa: int, n_sites: int, seed: int, mean_branch: float | None = None, gap_fraction=0.01,
              block: int = 0):
    """model in {"M0", "M7", "M8", "ModelA", "free-ratio"}.  Returns a namespace with everything the
    boundary needs (parent, leaf_node_ids, tree_structure, branch_lengths, codes, pi, Q, proportions, ...).

    ``seed`` fixes the tree and the model parameters; ``block`` selects an independent block of ``n_sites``
    columns evolved on that same tree (rank r of a multi-GPU run evaluates block r: weak scaling)."""
    rng = np.random.default_rng(seed)
    n_branches = 2 * n_taxa - 3
    if mean_branch is None:
        mean_branch = min(0.1, 10.0 / n_branches)
    parent, bl = random_tree(n_taxa, rng, mean_branch)
    n_nodes = len(parent)
    pi = rng.dirichlet(5.0 * np.ones(61))
    kappa = 2.5
    tree_rng = rng
    rng = np.random.default_rng([seed, 7919 + block])  # sequences: independent of the tree stream, one per block
    case = SimpleNamespace(model=model, n_taxa=n_taxa, n_sites=n_sites, seed=seed, parent=parent, branch_lengths=bl, pi=pi,
                           kappa=kappa, n_nodes=n_nodes, leaf_node_ids=np.arange(n_taxa, dtype=np.int32),
                           tree_structure=[(v, None if parent[v] < 0 else int(parent[v])) for v in range(n_nodes)],
                           tree_length=float(bl.sum()))
    if model in ("M0", "M7", "M8"):
        if model == "M0":
            m = models.M0CodonModel(kappa=kappa, omega=0.3, pi=pi)
            Q, props = np.stack([m.get_Q_matrix()]), np.array([1.0])
        elif model == "M7":
            m = models.M7CodonModel(kappa=kappa, p_beta=0.7, q_beta=1.5, ncatG=10, pi=pi)
            Q, props = np.stack(m.get_Q_matrices()), np.array(m.get_site_classes()[0])
        else:
            m = models.M8CodonModel(kappa=kappa, p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=pi)
            Q, props = np.stack(m.get_Q_matrices()), np.array(m.get_site_classes()[0])
        site_class = rng.choice(len(props), size=n_sites, p=props / props.sum())
        case.Q, case.proportions = Q, props
        case.class_omegas = np.array([0.3]) if model == "M0" else np.array(m.get_site_classes()[1])
        case.codes = evolve(parent, bl, pi, lambda v, k: (k, Q[k], bl[v]), site_class, n_sites, rng, gap_fraction)
    elif model == "ModelA":
        p0, p1, w0, w2 = 0.5, 0.3, 0.1, 3.0
        labels = np.zeros(n_nodes - 1, dtype=np.uint8)
        internals = np.arange(n_taxa, n_nodes - 1)
        fg_root = int(tree_rng.choice(internals)) if len(internals) else 0
        below = np.zeros(n_nodes, dtype=bool)
        below[fg_root] = True
        for v in range(n_nodes - 2, -1, -1):
            if parent[v] >= 0 and below[parent[v]]:
                below[v] = True
        labels[below[: n_nodes - 1]] = 1
        if labels.all() or not labels.any():
            labels[:] = 0
            labels[0] = 1
        ma = models.BranchSiteModelA(pi, np.concatenate([labels.astype(int)]))
        freqs = ma.compute_site_class_frequencies(p0, p1)
        qb, qf = ma.compute_qfactors(kappa, p0, p1, w0, w2)
        Q3 = np.stack(ma.unnormalised_Qs(kappa, w0, w2))
        site_class = rng.choice(4, size=n_sites, p=freqs)

        def q_of(v, k):
            lab = labels[v] if v < n_nodes - 1 else 0
            w = ma.get_omega_for_branch_class(int(lab), k, w0, w2)
            q = 0 if w == w0 else (1 if w == 1.0 else 2)
            return q, Q3[q], bl[v] * (qf if lab else qb)

        case.Q3, case.qfactors, case.site_class_freqs, case.branch_labels = Q3, (qb, qf), freqs, labels
        case.omega0, case.omega2, case.p0, case.p1 = w0, w2, p0, p1
        case.codes = evolve(parent, bl, pi, q_of, site_class, n_sites, rng, gap_fraction)
    elif model == "free-ratio":
        omegas = np.exp(tree_rng.normal(np.log(0.3), 0.5, size=n_nodes - 1))
        bm = models.CodonBranchModel(pi, np.zeros(n_nodes, dtype=int), free_ratio=True, n_nodes=n_nodes)
        Qn = bm.get_Q_matrices(kappa, omegas)
        case.Q_per_node, case.omegas = Qn, omegas
        case.omega_per_node = np.concatenate([omegas, omegas[:1]])   # the root's matrix is built (from omega 0) and ignored
        case.codes = evolve(parent, bl, pi, lambda v, k: (0, Qn[v], bl[v]), np.zeros(n_sites, dtype=int), n_sites, rng, gap_fraction)
    else:
        raise ValueError(f"unknown model {model}")
    return case


def as_alignment_and_tree(case):
    """(Alignment, Tree) objects for a synthetic case: what the adapter and the optimiser mirrors take."""
    from .io import Alignment, Tree
    parent = np.asarray(case.parent)
    kids = [[] for _ in parent]
    for v, p in enumerate(parent):
        if p >= 0:
            kids[p].append(v)
    leaf_name = {int(v): f"t{i}" for i, v in enumerate(case.leaf_node_ids)}

    def newick(v):
        s = leaf_name[v] if not kids[v] else "(" + ",".join(newick(c) for c in kids[v]) + ")"
        return s + (f":{float(case.branch_lengths[v])!r}" if parent[v] >= 0 else "")

    root = int(np.flatnonzero(parent < 0)[0])
    return Alignment.from_codes([leaf_name[int(v)] for v in case.leaf_node_ids], case.codes), Tree.from_newick(newick(root) + ";")
