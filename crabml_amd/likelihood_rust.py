"""Python adapter in front of the engine: the mirror of crabML's ``RustLikelihoodCalculator``.

Same constructor, attributes and three methods as the reference adapter
(src/crabml/core/likelihood_rust.py:21-364), so callers written against it --
the optimisers, BEB -- work unchanged.  Differences behind the interface: the tree
and the alignment are handed to a persistent :class:`crabml_amd.engine.Engine` once,
in ``__init__``; each call only ships rate matrices and branch lengths.

Node numbering contract (likelihood_rust.py:98-142): leaves first, in post-order,
get ids 0..N-1; internal nodes follow in post-order, so the root is last;
``tree_structure[j] == (j, parent_id or None)``; ``sequences[i]`` belongs to node i.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine
from .io import Alignment, Tree


class RustLikelihoodCalculator:
    def __init__(self, alignment: Alignment, tree: Tree, device: int = 0, engine: Engine | None = None,
                 pattern_slice: tuple | None = None):
        self.alignment = alignment
        self.tree = tree
        if len(alignment.names) != tree.n_leaves:
            raise ValueError(f"Alignment has {len(alignment.names)} sequences but tree has {tree.n_leaves} leaves")
        a, t = set(alignment.names), set(tree.leaf_names)
        if a != t:
            raise ValueError("Alignment and tree have different species. "
                             f"In alignment but not tree: {a - t}. In tree but not alignment: {t - a}")
        if alignment.seqtype != "codon":
            raise ValueError(f"Unknown sequence type: {alignment.seqtype}")
        self.n_states = 61
        self.n_sites = alignment.n_sites
        self.leaf_to_seq_idx = {name: i for i, name in enumerate(alignment.names)}
        self._prepare_tree_structure()
        self._prepare_sequences()

        self.engine = engine if engine is not None else Engine(device)
        self.engine.set_tree_structure(self.tree_structure, self.leaf_node_ids)
        begin, end = pattern_slice if pattern_slice is not None else (0, -1)
        self.engine.set_alignment(self.sequences, n_states=self.n_states, compress=True, pattern_begin=begin, pattern_end=end)

    # -- boundary inputs ------------------------------------------------------------------
    def _prepare_tree_structure(self):
        post = self.tree.postorder()
        leaves = [v for v in post if self.tree.is_leaf(v)]
        internals = [v for v in post if not self.tree.is_leaf(v)]
        self.postorder_nodes = leaves + internals
        self.node_id_map = {v: i for i, v in enumerate(self.postorder_nodes)}
        self.tree_structure = []
        self.branch_lengths_template = []
        for v in self.postorder_nodes:
            p = int(self.tree.parent[v])
            self.tree_structure.append((self.node_id_map[v], self.node_id_map[p] if p >= 0 else None))
            self.branch_lengths_template.append(float(self.tree.branch_length[v]) if p >= 0 else 0.0)
        self.leaf_names_ordered = [self.tree.name[v] for v in leaves]
        self.leaf_node_ids = list(range(len(leaves)))

    def _prepare_sequences(self):
        rows = [self.leaf_to_seq_idx[name] for name in self.leaf_names_ordered]
        self.sequences = np.ascontiguousarray(np.asarray(self.alignment.sequences)[rows, :], dtype=np.int32)

    def _get_current_branch_lengths(self, scale_branch_lengths: float = 1.0) -> np.ndarray:
        bl = np.array([self.tree.branch_length[v] if self.tree.parent[v] >= 0 else 0.0 for v in self.postorder_nodes],
                      dtype=np.float64)
        return bl * scale_branch_lengths

    @property
    def parent_indices(self):
        return [-1 if p is None else p for _, p in self.tree_structure]

    def renumber_branch_labels(self) -> list:
        """Per-node 0/1 labels in boundary numbering, length n_nodes-1 (optimize/branch_site.py:79-90)."""
        out = [0] * (len(self.tree_structure) - 1)
        for v in range(self.tree.n_nodes):
            if self.tree.parent[v] < 0:
                continue
            rid = self.node_id_map[v]
            if rid < len(out):
                out[rid] = int(self.tree.label[v])
        return out

    # -- the three reference methods ----------------------------------------------------------
    def compute_log_likelihood(self, Q: np.ndarray, pi: np.ndarray, scale_branch_lengths: float = 1.0) -> float:
        Q = np.asarray(Q)
        pi = np.asarray(pi)
        if Q.shape != (self.n_states, self.n_states):
            raise ValueError(f"Q matrix has shape {Q.shape}, expected ({self.n_states}, {self.n_states})")
        if pi.shape != (self.n_states,):
            raise ValueError(f"pi has shape {pi.shape}, expected ({self.n_states},)")
        return self.engine.log_likelihood(Q, pi, self._get_current_branch_lengths(scale_branch_lengths))

    def _check_classes(self, Q_matrices, proportions):
        if len(proportions) != len(Q_matrices):
            raise ValueError(f"Number of proportions ({len(proportions)}) must match number of Q matrices ({len(Q_matrices)})")
        for k, Q in enumerate(Q_matrices):
            if np.shape(Q) != (self.n_states, self.n_states):
                raise ValueError(f"Q matrix {k} has shape {np.shape(Q)}, expected ({self.n_states}, {self.n_states})")

    def compute_log_likelihood_site_classes(self, Q_matrices, pi, proportions, scale_branch_lengths: float = 1.0,
                                            use_scaling: bool = False) -> float:
        self._check_classes(Q_matrices, proportions)
        return self.engine.site_classes(np.stack([np.asarray(q) for q in Q_matrices]), proportions, pi,
                                        self._get_current_branch_lengths(scale_branch_lengths))

    def compute_site_log_likelihoods(self, Q_matrices, pi, proportions, scale_branch_lengths: float = 1.0) -> np.ndarray:
        self._check_classes(Q_matrices, proportions)
        return self.engine.site_lnl_by_class(np.stack([np.asarray(q) for q in Q_matrices]), pi,
                                             self._get_current_branch_lengths(scale_branch_lengths))

    # -- the two paths the reference reaches through crabml_rust directly ---------------------------
    def compute_log_likelihood_branch_site(self, Q_omega0, Q_omega1, Q_omega2, qfactor_back, qfactor_fore, omega0, omega2,
                                           site_class_freqs, pi, branch_labels=None, branch_lengths=None) -> float:
        labels = self.renumber_branch_labels() if branch_labels is None else branch_labels
        bl = self._get_current_branch_lengths() if branch_lengths is None else branch_lengths
        return self.engine.branch_site(np.stack([Q_omega0, Q_omega1, Q_omega2]), qfactor_back, qfactor_fore, omega0, omega2,
                                       site_class_freqs, pi, bl, labels)

    def compute_log_likelihood_branch(self, Q_matrices, pi, branch_lengths=None) -> float:
        bl = self._get_current_branch_lengths() if branch_lengths is None else branch_lengths
        return self.engine.branch(Q_matrices, pi, bl)
