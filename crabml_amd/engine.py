"""Handle API: one Engine keeps a tree program and a compressed alignment resident on one MI355X.

Thin, allocation-free-per-call wrapper over the crab_* handle functions of include/crabml_amd.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def tree_arrays(tree_structure):
    """[(node_id, parent_or_None), ...] -> (node_ids int32, parents int32 with -1 for None)."""
    ids = np.fromiter((int(v) for v, _ in tree_structure), dtype=np.int32, count=len(tree_structure))
    par = np.fromiter((-1 if p is None else int(p) for _, p in tree_structure), dtype=np.int32, count=len(tree_structure))
    return ids, par


class Engine:
    """``Engine(0)``: one GPU.  ``Engine([0, 1, 2, 3])``: one handle over several GPUs of this process
    (crab_create_multi): patterns sliced over them, every evaluation on all of them at once, same methods."""

    def __init__(self, device=0):
        self._lib = L.load()
        self._h = L.vp()
        if np.ndim(device) == 0:
            L.check(self._lib.crab_create(int(device), C.byref(self._h)))
            self.device = int(device)
        else:
            devs, dp_ = L.i32(list(device))
            L.check(self._lib.crab_create_multi(dp_, len(devs), C.byref(self._h)))
            self.device = [int(d) for d in devs]
        self.n_nodes = 0
        self.n_states = 0
        self.n_sites = 0

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.crab_destroy(self._h)
            self._h = L.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, status):
        L.check(status, self._h)

    # -- state ------------------------------------------------------------------------------
    def set_tree(self, parents, leaf_node_ids, node_ids=None):
        par, parp = L.i32(parents)
        leaf, leafp = L.i32(leaf_node_ids)
        if node_ids is None:
            idp = None
        else:
            ids, idp = L.i32(node_ids)
        self._check(self._lib.crab_set_tree(self._h, len(par), idp, parp, leafp, len(leaf)))
        self.n_nodes = len(par)

    def set_tree_structure(self, tree_structure, leaf_node_ids):
        ids, par = tree_arrays(tree_structure)
        self.set_tree(par, leaf_node_ids, ids)

    def set_alignment(self, codes, n_states=61, compress=True, pattern_begin=0, pattern_end=-1):
        codes_, cp = L.i32(codes)
        if codes_.ndim != 2:
            raise ValueError("codes must be [n_leaves, n_sites]")
        self._check(self._lib.crab_set_alignment(self._h, cp, codes_.shape[0], codes_.shape[1], int(n_states),
                                                 1 if compress else 0, int(pattern_begin), int(pattern_end)))
        self.n_states = int(n_states)
        self.n_sites = codes_.shape[1]

    def set_eigensolver(self, where: str):
        """"host" (default: shortest latency for a few matrices) or "device" (many matrices per call: free-ratio fits,
        gene scans); crab_set_eigensolver."""
        self._check(self._lib.crab_set_eigensolver(self._h, {"host": 0, "device": 1}[where]))

    @property
    def n_devices(self) -> int:
        return int(self._lib.crab_n_devices(self._h))

    @property
    def n_patterns(self) -> int:
        return int(self._lib.crab_n_patterns(self._h))

    @property
    def n_patterns_total(self) -> int:
        return int(self._lib.crab_n_patterns_total(self._h))

    # -- evaluations ------------------------------------------------------------------------
    def _q_stack(self, Q, K=None):
        Q_ = np.ascontiguousarray(np.asarray(Q, dtype=np.float64))
        if Q_.ndim == 2:
            Q_ = Q_[None]
        n = self.n_states
        if Q_.shape[1:] != (n, n):
            raise ValueError(f"Q matrix has shape {Q_.shape[1:]}, expected ({n}, {n})")
        return Q_, Q_.ctypes.data_as(L.dp)

    def _pi(self, pi):
        pi_, pp = L.f64(pi)
        if pi_.shape != (self.n_states,):
            raise ValueError(f"pi must have length {self.n_states}, got {pi_.shape}")
        return pi_, pp

    def _bl(self, branch_lengths):
        bl, blp = L.f64(branch_lengths)
        if bl.shape != (self.n_nodes,):
            raise ValueError(f"branch_lengths must have one entry per node ({self.n_nodes}), got {bl.shape}")
        return bl, blp

    def site_classes(self, Q, proportions, pi, branch_lengths, no_clamp=False, return_sites=False):
        Q_, Qp = self._q_stack(Q)
        pr, prp = L.f64(proportions)
        if len(pr) != Q_.shape[0]:
            raise ValueError(f"Number of proportions ({len(pr)}) must match number of Q matrices ({Q_.shape[0]})")
        pi_, pp = self._pi(pi)
        bl, blp = self._bl(branch_lengths)
        lnL = C.c_double()
        site = np.full(self.n_sites, np.nan) if return_sites else None
        self._check(self._lib.crab_eval_site_classes(self._h, Q_.shape[0], Qp, prp, pp, blp,
                                                     L.CRAB_FLAG_NO_CLAMP if no_clamp else 0, C.byref(lnL),
                                                     site.ctypes.data_as(L.dp) if return_sites else None))
        return (lnL.value, site) if return_sites else lnL.value

    def site_classes_batch(self, Q, proportions, pi, branch_lengths, no_clamp=False):
        """lnL for B parameter vectors at once: Q [B, K, n, n], proportions [B, K], branch_lengths [B, n_nodes]
        (pi shared).  Element b equals ``site_classes(Q[b], proportions[b], pi, branch_lengths[b])`` bit for bit."""
        Q_ = np.ascontiguousarray(np.asarray(Q, dtype=np.float64))
        n = self.n_states
        if Q_.ndim != 4 or Q_.shape[2:] != (n, n):
            raise ValueError(f"Q must have shape (B, K, {n}, {n}), got {Q_.shape}")
        B, K = Q_.shape[:2]
        pr = np.ascontiguousarray(np.asarray(proportions, dtype=np.float64))
        bl = np.ascontiguousarray(np.asarray(branch_lengths, dtype=np.float64))
        if pr.shape != (B, K):
            raise ValueError(f"proportions must have shape ({B}, {K}), got {pr.shape}")
        if bl.shape != (B, self.n_nodes):
            raise ValueError(f"branch_lengths must have shape ({B}, {self.n_nodes}), got {bl.shape}")
        pi_, pp = self._pi(pi)
        out = np.empty(B)
        self._check(self._lib.crab_eval_site_classes_batch(self._h, B, K, Q_.ctypes.data_as(L.dp), pr.ctypes.data_as(L.dp), pp,
                                                           bl.ctypes.data_as(L.dp), L.CRAB_FLAG_NO_CLAMP if no_clamp else 0,
                                                           out.ctypes.data_as(L.dp)))
        return out

    def log_likelihood(self, Q, pi, branch_lengths, **kw):
        return self.site_classes(np.asarray(Q)[None], [1.0], pi, branch_lengths, **kw)

    def site_lnl_by_class(self, Q, pi, branch_lengths, no_clamp=False):
        Q_, Qp = self._q_stack(Q)
        pi_, pp = self._pi(pi)
        bl, blp = self._bl(branch_lengths)
        out = np.full((self.n_sites, Q_.shape[0]), np.nan)
        self._check(self._lib.crab_eval_site_lnl_by_class(self._h, Q_.shape[0], Qp, pp, blp,
                                                          L.CRAB_FLAG_NO_CLAMP if no_clamp else 0, out.ctypes.data_as(L.dp)))
        return out

    def beb_grid(self, Q, pi, branch_lengths, cols, props, partial=False, lnfx=None):
        """Site posteriors integrated over a parameter grid (analysis/beb.py:575-661 on the device).
        Q: [K, n, n]; cols int [G, n_mix] and props [G, n_mix] describe each grid point's mixture.
        Returns (posteriors [n_sites, n_mix], lnfX per grid point [G], ln f(X)).
        Over pattern slices (one engine per GPU): ``partial=True`` returns (None, this slice's lnfX, 0.0); add the
        slices' lnfX vectors and pass the sum as ``lnfx=`` to get the posteriors of this engine's sites."""
        Q_, Qp = self._q_stack(Q)
        pi_, pp = self._pi(pi)
        bl, blp = self._bl(branch_lengths)
        cols_ = np.ascontiguousarray(cols, dtype=np.int32)
        props_ = np.ascontiguousarray(props, dtype=np.float64)
        if cols_.ndim != 2 or props_.shape != cols_.shape:
            raise ValueError("cols and props must both have shape (G, n_mix)")
        G, n_mix = cols_.shape
        flags = 0
        if partial:
            flags, post = L.CRAB_FLAG_BEB_PARTIAL, None
            lnfx_ = np.empty(G)
        else:
            post = np.full((self.n_sites, n_mix), np.nan)
            if lnfx is None:
                lnfx_ = np.empty(G)
            else:
                flags = L.CRAB_FLAG_BEB_GIVEN_LNFX
                lnfx_ = np.ascontiguousarray(lnfx, dtype=np.float64).copy()
                if lnfx_.shape != (G,):
                    raise ValueError(f"lnfx must have one entry per grid point ({G})")
        ln_fx = C.c_double()
        self._check(self._lib.crab_eval_beb_grid(self._h, Q_.shape[0], Qp, pp, blp, G, n_mix, cols_.ctypes.data_as(L.ip),
                                                 props_.ctypes.data_as(L.dp), flags,
                                                 post.ctypes.data_as(L.dp) if post is not None else None,
                                                 lnfx_.ctypes.data_as(L.dp), C.byref(ln_fx)))
        return post, lnfx_, ln_fx.value

    def branch_site(self, Q3, qfactor_back, qfactor_fore, omega0, omega2, site_class_freqs, pi, branch_lengths, labels,
                    return_sites=False):
        Q_, Qp = self._q_stack(Q3)
        if Q_.shape[0] != 3:
            raise ValueError("Q3 must hold Q(omega0), Q(1), Q(omega2)")
        fr, frp = L.f64(site_class_freqs)
        if len(fr) != 4:
            raise ValueError("site_class_freqs must have 4 entries")
        pi_, pp = self._pi(pi)
        bl, blp = self._bl(branch_lengths)
        lab, labp = L.u8(labels)
        lnL = C.c_double()
        site = np.full(self.n_sites, np.nan) if return_sites else None
        self._check(self._lib.crab_eval_branch_site(self._h, Qp, float(qfactor_back), float(qfactor_fore), float(omega0),
                                                    float(omega2), frp, pp, blp, labp, len(lab), 0, C.byref(lnL),
                                                    site.ctypes.data_as(L.dp) if return_sites else None))
        return (lnL.value, site) if return_sites else lnL.value

    def branch(self, Q_per_node, pi, branch_lengths, no_clamp=False, return_sites=False):
        Q_, Qp = self._q_stack(Q_per_node)
        pi_, pp = self._pi(pi)
        bl, blp = L.f64(branch_lengths)
        lnL = C.c_double()
        site = np.full(self.n_sites, np.nan) if return_sites else None
        self._check(self._lib.crab_eval_branch(self._h, Q_.shape[0], Qp, pp, blp, len(bl),
                                               L.CRAB_FLAG_NO_CLAMP if no_clamp else 0, C.byref(lnL),
                                               site.ctypes.data_as(L.dp) if return_sites else None))
        return (lnL.value, site) if return_sites else lnL.value

    # -- evaluations from model parameters (rate matrices built behind the boundary) ----------
    def _batch(self, x, B=None, inner=None, name="parameter"):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
        if inner is not None:
            a = np.ascontiguousarray(a.reshape(-1, inner)) if a.ndim == 1 and a.size == inner else a
            if a.ndim != 2 or a.shape[1] != inner:
                raise ValueError(f"{name} must have shape (B, {inner}), got {a.shape}")
        elif a.ndim != 1:
            raise ValueError(f"{name} must be a scalar or a vector of B values, got shape {a.shape}")
        if B is not None and a.shape[0] != B:
            raise ValueError(f"{name} has {a.shape[0]} entries, expected {B}")
        return a

    def codon_site_classes(self, kappa, omegas, proportions, pi, branch_lengths, no_clamp=False):
        """lnL of site-class mixtures given (kappa, class omegas, class proportions); scalars/vectors evaluate one
        parameter vector and return a float, leading batch dimensions evaluate B vectors and return an array."""
        single = np.ndim(kappa) == 0
        ka = self._batch(kappa, name="kappa")
        B = len(ka)
        om = np.ascontiguousarray(np.asarray(omegas, dtype=np.float64)).reshape(B, -1)
        K = om.shape[1]
        pr = self._batch(proportions, B, K, "proportions")
        bl = self._batch(branch_lengths, B, self.n_nodes, "branch_lengths")
        pi_, pp = self._pi(pi)
        out = np.empty(B)
        self._check(self._lib.crab_eval_codon_site_classes(self._h, B, K, ka.ctypes.data_as(L.dp), om.ctypes.data_as(L.dp),
                                                           pr.ctypes.data_as(L.dp), pp, bl.ctypes.data_as(L.dp),
                                                           L.CRAB_FLAG_NO_CLAMP if no_clamp else 0, out.ctypes.data_as(L.dp)))
        return float(out[0]) if single else out

    def codon_class_posteriors(self, kappa, omegas, proportions, weights, pi, branch_lengths):
        """[n_sites, K] class posteriors averaged over B weighted parameter vectors (analysis/beb.py:698-729):
        kappa, weights [B]; omegas, proportions [B, K]; branch_lengths [n_nodes] shared by all vectors."""
        ka = self._batch(kappa, name="kappa")
        B = len(ka)
        om = np.ascontiguousarray(np.asarray(omegas, dtype=np.float64)).reshape(B, -1)
        K = om.shape[1]
        pr = self._batch(proportions, B, K, "proportions")
        w = self._batch(weights, B, name="weights")
        bl, blp = self._bl(branch_lengths)
        pi_, pp = self._pi(pi)
        out = np.full((self.n_sites, K), np.nan)
        self._check(self._lib.crab_eval_codon_class_posteriors(self._h, B, K, ka.ctypes.data_as(L.dp), om.ctypes.data_as(L.dp),
                                                               pr.ctypes.data_as(L.dp), w.ctypes.data_as(L.dp), pp, blp, 0,
                                                               out.ctypes.data_as(L.dp)))
        return out

    def codon_branch(self, kappa, omega_per_node, pi, branch_lengths, no_clamp=False):
        """Branch model from (kappa, one omega per node); batched like codon_site_classes."""
        single = np.ndim(kappa) == 0
        ka = self._batch(kappa, name="kappa")
        B = len(ka)
        om = self._batch(omega_per_node, B, self.n_nodes, "omega_per_node")
        bl = self._batch(branch_lengths, B, self.n_nodes, "branch_lengths")
        pi_, pp = self._pi(pi)
        out = np.empty(B)
        self._check(self._lib.crab_eval_codon_branch(self._h, B, ka.ctypes.data_as(L.dp), om.ctypes.data_as(L.dp), pp,
                                                     bl.ctypes.data_as(L.dp), L.CRAB_FLAG_NO_CLAMP if no_clamp else 0,
                                                     out.ctypes.data_as(L.dp)))
        return float(out[0]) if single else out

    def codon_branch_site(self, kappa, omega0, omega2, p0, p1, pi, branch_lengths, labels):
        """Branch-site Model A from (kappa, omega0, omega2, p0, p1); batched like codon_site_classes."""
        single = np.ndim(kappa) == 0
        ka = self._batch(kappa, name="kappa")
        B = len(ka)
        w0, w2, p0_, p1_ = (self._batch(x, B, name=n) for x, n in ((omega0, "omega0"), (omega2, "omega2"), (p0, "p0"), (p1, "p1")))
        bl = self._batch(branch_lengths, B, self.n_nodes, "branch_lengths")
        pi_, pp = self._pi(pi)
        lab, labp = L.u8(labels)
        out = np.empty(B)
        self._check(self._lib.crab_eval_codon_branch_site(self._h, B, ka.ctypes.data_as(L.dp), w0.ctypes.data_as(L.dp),
                                                          w2.ctypes.data_as(L.dp), p0_.ctypes.data_as(L.dp), p1_.ctypes.data_as(L.dp),
                                                          pp, bl.ctypes.data_as(L.dp), labp, len(lab), 0, out.ctypes.data_as(L.dp)))
        return float(out[0]) if single else out

    # -- measurement ------------------------------------------------------------------------
    def last_timing(self):
        """(device_ms, prune_kernel_ms, host_eigen_ms) of the last evaluation; HIP events on the engine's stream."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.crab_last_timing(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def export_transition_matrices(self, k=0):
        """[n_nodes, n, n] transition matrices the last evaluation built on the device for class k."""
        out = np.empty((self.n_nodes, self.n_states, self.n_states))
        self._check(self._lib.crab_export_transition_matrices(self._h, int(k), out.ctypes.data_as(L.dp)))
        return out

    def eig_pool_stats(self):
        """(hits, misses, resident entries) of this engine's eigensystem pool since creation."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.crab_eig_pool_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_clade_tables(self, on: bool):
        """Clade tables (subtree site repeats; crab_set_clade_tables): on by default, bit-identical to whole walks."""
        self._check(self._lib.crab_set_clade_tables(self._h, 1 if on else 0))

    def last_clade_tables(self):
        """(clades the last evaluation walked once per distinct sub-pattern, sum of their distinct sub-pattern counts)."""
        n, u = C.c_int(), C.c_int64()
        self._check(self._lib.crab_last_clade_tables(self._h, C.byref(n), C.byref(u)))
        return n.value, u.value

    def last_prune_launches(self) -> int:
        """Launches of the pruning kernel in the last evaluation (crab_last_prune_launches)."""
        n = C.c_int()
        self._check(self._lib.crab_last_prune_launches(self._h, C.byref(n)))
        return n.value

    def last_partial_updates(self) -> int:
        """Vectors of the last batched evaluation that re-walked only a branch-to-root path (crab_last_partial_updates)."""
        n = C.c_int()
        self._check(self._lib.crab_last_partial_updates(self._h, C.byref(n)))
        return n.value

    def last_tree_parts(self) -> int:
        """Subtree programs the last single evaluation was cut into (crab_last_tree_parts); 0: one whole-tree walk per item."""
        n = C.c_int()
        self._check(self._lib.crab_last_tree_parts(self._h, C.byref(n)))
        return n.value

    def last_work(self):
        """(dense MFMA updates, gathered leaf updates) of the last evaluation, in pattern x branch x class units."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.crab_last_work(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


# -- host-only helpers (no GPU needed) ----------------------------------------------------------
OP_FIELDS = ("kind", "node", "flags", "arg", "slot")
OP_GEMM, OP_LEAF, OP_CONST = 0, 1, 2
F_MUL_PROD, F_POP_MUL, F_PUSH, F_FIN, F_RESCALE = 1, 2, 4, 8, 16


def compile_tree(parents, leaf_node_ids, node_ids=None):
    """The device program for a tree: int32 array [n_ops, 5] (kind, node, flags, arg, slot) and the stack depth."""
    lib = L.load()
    par, parp = L.i32(parents)
    leaf, leafp = L.i32(leaf_node_ids)
    idp = None
    if node_ids is not None:
        ids, idp = L.i32(node_ids)
    n_ops, depth = C.c_int(), C.c_int()
    L.check(lib.crab_compile_tree(len(par), idp, parp, leafp, len(leaf), None, 0, C.byref(n_ops), C.byref(depth)))
    ops = np.zeros((n_ops.value, 5), dtype=np.int32)
    L.check(lib.crab_compile_tree(len(par), idp, parp, leafp, len(leaf), ops.ctypes.data_as(L.ip), n_ops.value,
                                  C.byref(n_ops), C.byref(depth)))
    return ops, depth.value


def compile_path_program(parents, leaf_node_ids, node=-1, node_ids=None):
    """int32 [n_ops, 6] (kind, node, flags, arg, slot, save slot): the program that recomputes the root when only the
    branch above ``node`` moved (node = -1: the whole-tree program in the same form), and the number of save slots."""
    lib = L.load()
    par, parp = L.i32(parents)
    leaf, leafp = L.i32(leaf_node_ids)
    idp = None
    if node_ids is not None:
        ids, idp = L.i32(node_ids)
    n_ops, n_slots = C.c_int(), C.c_int()
    L.check(lib.crab_compile_path_program(len(par), idp, parp, leafp, len(leaf), int(node), None, 0, C.byref(n_ops), C.byref(n_slots)))
    ops = np.zeros((n_ops.value, 6), dtype=np.int32)
    L.check(lib.crab_compile_path_program(len(par), idp, parp, leafp, len(leaf), int(node), ops.ctypes.data_as(L.ip), n_ops.value,
                                          C.byref(n_ops), C.byref(n_slots)))
    return ops, n_slots.value


def compile_tree_parts(parents, leaf_node_ids, target, node_ids=None):
    """(root program, [subtree programs]) of a tree cut into parts of at most ``target`` products, each int32 [n_ops, 6]."""
    lib = L.load()
    par, parp = L.i32(parents)
    leaf, leafp = L.i32(leaf_node_ids)
    idp = None
    if node_ids is not None:
        ids, idp = L.i32(node_ids)

    def one(part):
        n_ops, n_parts = C.c_int(), C.c_int()
        L.check(lib.crab_compile_tree_part(len(par), idp, parp, leafp, len(leaf), int(target), part, None, 0, C.byref(n_ops), C.byref(n_parts)))
        ops = np.zeros((n_ops.value, 6), dtype=np.int32)
        L.check(lib.crab_compile_tree_part(len(par), idp, parp, leafp, len(leaf), int(target), part, ops.ctypes.data_as(L.ip), n_ops.value,
                                           C.byref(n_ops), C.byref(n_parts)))
        return ops, n_parts.value

    root, n_parts = one(-1)
    return root, [one(i)[0] for i in range(n_parts)]


def compile_clade_tables(parents, leaf_node_ids, codes, n_states=61, node_ids=None):
    """Clade tables as programs (crab_compile_clade_tables; host only): (main program, [clade dicts], inner clades first) for
    the patterns in ``codes`` [n_leaves, n_patterns]; a clade dict holds its ``ops`` (int32 [n_ops, 6]), ``root`` node,
    ``parent`` clade (-1: the main program), ``row`` (its wide row in its consumer's programs), ``index`` (its sub-pattern for
    every pattern of its consumer) and ``sub_codes`` ([n_leaves, U])."""
    lib = L.load()
    par, parp = L.i32(parents)
    leaf, leafp = L.i32(leaf_node_ids)
    cod, codp = L.i32(codes)
    idp = None
    if node_ids is not None:
        ids, idp = L.i32(node_ids)
    n_leaves, n_pat = cod.shape

    def one(part):
        n_ops, n_cl, root, parent, row, U = C.c_int(), C.c_int(), C.c_int(-1), C.c_int(-1), C.c_int(0), C.c_int64(0)
        args = (len(par), idp, parp, leafp, len(leaf), codp, n_pat, int(n_states), part)
        refs = (C.byref(n_ops), C.byref(n_cl), C.byref(root), C.byref(parent), C.byref(row), C.byref(U))
        L.check(lib.crab_compile_clade_tables(*args, None, 0, *refs, None, None))
        ops = np.zeros((n_ops.value, 6), dtype=np.int32)
        index = np.zeros(n_pat, dtype=np.uint32)
        sub = np.zeros((n_leaves, max(1, U.value)), dtype=np.int32)
        L.check(lib.crab_compile_clade_tables(*args, ops.ctypes.data_as(L.ip), n_ops.value, *refs,
                                              index.ctypes.data_as(C.POINTER(C.c_uint32)) if part >= 0 else None,
                                              sub.ctypes.data_as(L.ip) if part >= 0 else None))
        return ops, n_cl.value, dict(ops=ops, root=root.value, parent=parent.value, row=row.value, index=index, sub_codes=sub[:, : U.value])

    main, n_cl, _ = one(-1)
    return main, [one(j)[2] for j in range(n_cl)]


def plan_clade_work(parents, leaf_node_ids, codes, n_states=61, node_ids=None):
    """(clades, distinct sub-patterns, products per class with clade tables, products per class of whole walks) for the
    patterns in ``codes`` [n_leaves, n_patterns] (crab_plan_clade_work; host only)."""
    lib = L.load()
    par, parp = L.i32(parents)
    leaf, leafp = L.i32(leaf_node_ids)
    cod, codp = L.i32(codes)
    idp = None
    if node_ids is not None:
        ids, idp = L.i32(node_ids)
    n_cl, sub, prod, whole = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64()
    L.check(lib.crab_plan_clade_work(len(par), idp, parp, leafp, len(leaf), codp, cod.shape[1], int(n_states), C.byref(n_cl),
                                     C.byref(sub), C.byref(prod), C.byref(whole)))
    return n_cl.value, sub.value, prod.value, whole.value


def compress_patterns(codes, n_states=61):
    lib = L.load()
    codes_, cp = L.i32(codes)
    n_leaves, n_sites = codes_.shape
    s2p = np.zeros(n_sites, dtype=np.int64)
    w = np.zeros(n_sites, dtype=np.float64)
    first = np.zeros(n_sites, dtype=np.int64)
    n = C.c_int64()
    L.check(lib.crab_compress_patterns(cp, n_leaves, n_sites, int(n_states), s2p.ctypes.data_as(L.lp), w.ctypes.data_as(L.dp),
                                       first.ctypes.data_as(L.lp), C.byref(n)))
    return s2p, w[: n.value].copy(), first[: n.value].copy()


def spectral_decompose(Q, pi):
    lib = L.load()
    Q_, Qp = L.f64(Q)
    pi_, pp = L.f64(pi)
    n = Q_.shape[0]
    U, lam, Uinv = np.empty((n, n)), np.empty(n), np.empty((n, n))
    L.check(lib.crab_spectral_decompose(n, Qp, pp, U.ctypes.data_as(L.dp), lam.ctypes.data_as(L.dp), Uinv.ctypes.data_as(L.dp)))
    return U, lam, Uinv


def build_site_class_q_matrices(kappa, omegas, proportions, pi):
    """[K, 61, 61] rate matrices exactly as crab_eval_codon_site_classes builds them (host only)."""
    lib = L.load()
    om, omp = L.f64(omegas)
    pr, prp = L.f64(proportions)
    pi_, pp = L.f64(pi)
    if len(pr) != len(om) or pi_.shape != (61,):
        raise ValueError("omegas and proportions must have equal length and pi 61 entries")
    Q = np.empty((len(om), 61, 61))
    L.check(lib.crab_build_site_class_q_matrices(len(om), float(kappa), omp, prp, pp, Q.ctypes.data_as(L.dp)))
    return Q


def discretise_omega(kind, a, b, K):
    """Median-quantile class omegas: kind "beta" (M7/M8/M8a) or "gamma" (M5, shape a, rate b); host only."""
    lib = L.load()
    out = np.empty(int(K))
    L.check(lib.crab_discretise_omega({"beta": 0, "gamma": 1}[kind], float(a), float(b), int(K), out.ctypes.data_as(L.dp)))
    return out


def discretise_omega_mixture(kind, params, K):
    """Median-quantile class omegas of a two-component mixture: kind "M6" (p0, alpha1, beta1, alpha2) or "M9" (p0, p_beta,
    q_beta, alpha, beta_gamma); Brent's method as scipy.optimize.brentq runs it (crab_discretise_omega_mixture; host only)."""
    lib = L.load()
    pr, prp = L.f64(params)
    out = np.empty(int(K))
    L.check(lib.crab_discretise_omega_mixture({"M6": 2, "M9": 3}[kind], prp, len(pr), int(K), out.ctypes.data_as(L.dp)))
    return out


def model_a_quantities(kappa, omega0, omega2, p0, p1, pi):
    """(site_class_freqs[4], (qfactor_back, qfactor_fore)) as crab_eval_codon_branch_site computes them (host only)."""
    lib = L.load()
    pi_, pp = L.f64(pi)
    fr, qf = np.empty(4), np.empty(2)
    L.check(lib.crab_model_a_quantities(float(kappa), float(omega0), float(omega2), float(p0), float(p1), pp,
                                        fr.ctypes.data_as(L.dp), qf.ctypes.data_as(L.dp)))
    return fr, (float(qf[0]), float(qf[1]))


def device_available() -> bool:
    return bool(L.load().crab_device_available())
