"""Shared helpers for the test-suite: fixture loading, tolerances, a numpy emulator of the device program."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
DATA = os.path.join(GOLDEN, "data")

LYSOZYME_MODELS = ["m0", "m1a", "m2a", "m3", "m4", "m5", "m6", "m7", "m8", "m8a", "m9"]
SITE_MODEL_CASES = [f"lysozyme_{m}" for m in LYSOZYME_MODELS] + ["hiv_m8a", "lysin_m2a_gaps", "synthetic_edge"]

# Tolerances of BASELINE.json's north_star: per-pattern log-likelihoods within 1e-10 relative,
# total lnL within 1e-6 absolute.
SITE_RTOL = 1e-10
TOTAL_ATOL = 1e-6


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def tree_structure(parent):
    return [(i, None if p < 0 else int(p)) for i, p in enumerate(parent)]


def leaf_ids(case):
    return list(range(case["sequences"].shape[0]))


def assert_by_class_close(got, ref, what="", rtol=SITE_RTOL):
    """Compare [S, K] tables of ln f_{k,s}.

    Entries within e^-7 of their site's best class must agree to 1e-10 relative in ln f.  Entries far
    below the best class are differences of O(1e-16)-accurate transition probabilities and carry the
    rounding noise of whichever eigen-solver / expm produced them (the reference's Rust and Python
    paths differ there too); those are compared in probability space against the site's best class.
    """
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    best = ref.max(axis=1, keepdims=True)
    near = ref >= best - 7.0
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)  # ln f = 0 on an all-missing column
    assert np.all(rel[near] <= rtol), f"{what}: well-conditioned entries differ, max rel {rel[near].max():.3e}"
    d = np.abs(np.exp(got - best) - np.exp(ref - best))
    assert np.all(d[~near] <= 1e-9), f"{what}: far-tail entries differ by {d[~near].max():.3e} of the site's best class"


# -----------------------------------------------------------------------------------------------
# numpy emulator of the device tree program (crabml_amd/csrc/crab_kernels.h: TreeOp)
# -----------------------------------------------------------------------------------------------
OP_GEMM, OP_LEAF, OP_CONST = 0, 1, 2
F_MUL_PROD, F_POP_MUL, F_PUSH, F_FIN, F_RESCALE = 1, 2, 4, 8, 16


OP_LOAD = 3
F_POP_SAVED, F_SAVE = 512, 1024


F_TABLE = 2048


def emulate_program(ops, P_of_node, codes, pi, n_states=61, saved=None, save=False, tables=None, wide=None):
    """Run the op list exactly as prune_kernel does (minus rescaling, which is exact) and return f[s].

    P_of_node[v] is the n x n transition matrix on the branch above node v; codes [n_rows, S].  Ops have 5 fields
    (crab_compile_tree) or 6 (crab_compile_path_program: the sixth is the save slot).  ``save=True`` fills the dict
    ``saved`` the way a base walk with F_SAVE fills the save area (slot sv: operand of each product, sv+1: the parent's
    running product after the op unless it completes the parent, sv+2: the message before a pop); path programs (OP_LOAD,
    F_POP_SAVED) read it.  Clade tables: a clade program ends with OP_LOAD | F_SAVE | F_TABLE, at which the emulator returns
    the current tile [U, n_states] (the clade's table); in the main program a leaf op whose row is a key of ``wide`` gathers
    rows wide[row] of tables[node] instead of columns of P."""
    S = codes.shape[1]
    cur = prod = None
    stack = {}
    for op in ops:
        kind, node, flags, arg, slot = (int(x) for x in op[:5])
        sv = int(op[5]) if len(op) > 5 else 0
        if kind == OP_GEMM:
            if save:
                saved[sv] = prod
            t = prod @ P_of_node[node].T          # the current tile: a product always follows its operand's last op
            if save and flags & F_POP_MUL:
                saved[sv + 2] = t
        elif kind == OP_LEAF and wide is not None and arg in wide:
            t = tables[node][wide[arg]]
        elif kind == OP_LOAD and (flags & (F_SAVE | F_TABLE)) == (F_SAVE | F_TABLE):
            return prod
        elif kind == OP_LEAF:
            c = codes[arg]
            valid = (c >= 0) & (c < n_states)
            t = np.where(valid[:, None], P_of_node[node].T[np.where(valid, c, 0)], P_of_node[node].sum(axis=1)[None, :])
        elif kind == OP_LOAD and flags & F_SAVE:          # a tree part hands its tile over
            saved[sv] = prod
            continue
        elif kind == OP_LOAD:
            t = saved[sv]
        else:
            t = np.full((S, n_states), float(arg))
        if flags & F_MUL_PROD:
            t = t * prod
        if flags & F_POP_SAVED:
            t = t * saved[slot]
        elif flags & F_POP_MUL:
            t = t * stack.pop(slot)
        prod = t
        if save and kind == OP_GEMM and not flags & F_FIN:
            saved[sv + 1] = prod
        if flags & F_PUSH:
            assert slot not in stack, "stack slot overwritten while live"
            stack[slot] = prod
        if flags & F_FIN:
            cur = prod
    assert not stack, "stack not empty at the end of the program"
    assert cur is prod, "the last op must complete the node"
    return prod @ pi


# -----------------------------------------------------------------------------------------------
# how far apart may two correct evaluations of one site be?
# -----------------------------------------------------------------------------------------------
# Every eigen route to P(t) = exp(Qt) -- the reference's LAPACK eigh + inverse, the oracle's Jacobi, the engine's
# Householder + QL with one re-orthogonalisation step -- leaves an ABSOLUTE error of a few 1e-16 ... 1e-15 on each entry
# of P.  tests/golden/parity_tail.npz (made by make_parity_tail.py against 60-digit arithmetic) records the largest
# such error of each route over every matrix of four cases: none exceeds 5e-15; tests/test_oracle.py asserts
# EPS_P from the fixture.  Two routes therefore differ by at most 2 EPS_P per entry.
EPS_P = 1e-14


def eigen_route_bound(oracle, P, pi, ts, codes, leaf, mix, naive=False):
    """Per-site bound on |ln f(P') - ln f(P'')| over all families P', P'' of transition matrices within 2 EPS_P of P
    entry by entry.  The site likelihood is a polynomial with non-negative coefficients in the entries of the P's
    (products and sums of non-negative numbers only), hence monotone in every entry: both values lie between the
    likelihood of P - 2 EPS_P (floored at 0) and that of P + 2 EPS_P.  P: [K][n_nodes][n][n]; mix maps the [S, K]
    table of ln f_{k,s} to per-site ln f (monotone too).  On ordinary sites the bound is ~1e-12; it grows exactly where
    a site leans on entries of P that are themselves tiny, which is where two eigensolvers legitimately disagree
    (+inf where the floored matrices make the site impossible: such a site is decided by rounding noise alone)."""
    P = np.asarray(P)
    lo = mix(oracle.site_lnf_from_transition_matrices(np.maximum(P - 2 * EPS_P, 0.0), pi, ts, codes, leaf, naive=naive))
    hi = mix(oracle.site_lnf_from_transition_matrices(P + 2 * EPS_P, pi, ts, codes, leaf, naive=naive))
    with np.errstate(invalid="ignore"):
        return np.where(np.isfinite(lo), hi - lo, np.inf)


def check_work(work, n_clades, dense, gather):
    """Products and gathers the engine reports for its last evaluation: the nominal counts of a whole walk per pattern,
    or fewer of both when clade tables (subtree site repeats) walked some clades once per distinct sub-pattern."""
    if n_clades == 0:
        assert tuple(work) == (dense, gather), (work, dense, gather)
    else:
        assert 0 < work[0] < dense and 0 < work[1] < gather, (work, dense, gather)
