"""Starting branch lengths for a tree that has none: least-squares fit to pairwise distances.

Mirror of the reference's ``optimize/distance_init.py`` (:15-268), which M0Optimizer falls back on when every branch
of the input tree is shorter than 0.001 (optimizer.py:170-188).  Host logic, no likelihoods involved.  The tree distance
between two leaves is linear in the branch lengths, so the objective is ||d - A |b|||^2 with A the pair-by-branch path
incidence matrix; the reference minimises it with L-BFGS-B (numerical gradient, 100 iterations, ftol 1e-4) from
mean(d)/10, and so does this, on the same numbers: the result is a starting point, not an optimum.

Kept as the reference has it: observed distances are indexed by alignment row, tree distances by the order of the
leaves in ``tree.postorder()`` (distance_init.py:198-204 pairs ``pairwise_dists[i, j]`` with ``leaf_nodes[i], leaf_nodes[j]``).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import minimize


def compute_pairwise_distances(alignment) -> np.ndarray:
    """Jukes-Cantor-form distance on codon mismatches over sites valid in both rows (distance_init.py:15-62)."""
    seq = np.asarray(alignment.sequences)
    valid = (seq >= 0) & (seq < 61)
    n = seq.shape[0]
    d = np.zeros((n, n))
    for i in range(n):
        both = valid[i][None, :] & valid[i + 1:]
        n_valid = both.sum(axis=1)
        n_diff = ((seq[i][None, :] != seq[i + 1:]) & both).sum(axis=1)
        for off, (nv, nd) in enumerate(zip(n_valid, n_diff)):
            if nv > 0:
                p = nd / nv
                v = -0.75 * np.log(1.0 - (4.0 * p / 3.0)) if p < 0.74 else 5.0
            else:
                v = 0.1
            d[i, i + 1 + off] = d[i + 1 + off, i] = v
    return d


def path_incidence(tree, branch_nodes) -> np.ndarray:
    """[n_pairs, n_branches]: 1 where the branch above branch_nodes[b] lies on the path between leaves i < j, leaves in
    ``tree.postorder()`` order, pairs in row-major order of the upper triangle."""
    col = {int(v): b for b, v in enumerate(branch_nodes)}
    leaves = [v for v in tree.postorder() if tree.is_leaf(v)]
    chains = []
    for v in leaves:
        chain = set()
        while tree.parent[v] >= 0:
            chain.add(int(v))
            v = int(tree.parent[v])
        chains.append(chain)
    rows = []
    for i in range(len(leaves)):
        for j in range(i + 1, len(leaves)):
            row = np.zeros(len(branch_nodes))
            for v in chains[i] ^ chains[j]:
                if v in col:
                    row[col[v]] = 1.0
            rows.append(row)
    return np.asarray(rows)


def initialize_branch_lengths_ls(alignment, tree, branch_nodes) -> np.ndarray:
    d = compute_pairwise_distances(alignment)
    A = path_incidence(tree, branch_nodes)
    obs = d[np.triu_indices(d.shape[0], 1)]
    start = np.ones(len(branch_nodes)) * (np.mean(d[d > 0]) / 10.0)

    def sse(b):
        r = obs - A @ np.abs(b)
        return float(r @ r)

    res = minimize(sse, start, method="L-BFGS-B", bounds=[(0.0001, 5.0)] * len(branch_nodes), options={"maxiter": 100, "ftol": 1e-4})
    return np.abs(res.x)
