#!/usr/bin/env python3
"""Evidence for the tail of the per-site parity gate: where two evaluations of the SAME
algorithm disagree by more than 1e-10 relative per site, how far is each from the truth?

Every route to P(t) = exp(Qt) through a symmetric eigendecomposition -- the reference's LAPACK eigh + inverse
(rust/src/matrix.rs:41-67,88-102), the oracle's Jacobi (oracle/crab_oracle.c), the engine's Householder + QL
(crabml_amd/csrc/crab_eigen.cpp) -- carries an ABSOLUTE error of a few 1e-16 per entry, whatever its summation order.
On most sites that is far below 1e-10 relative in ln f.  On a site whose likelihood leans on entries of P that are
themselves 1e-8 ... 1e-12 (several substitutions on one short branch) it is not, and any two routes differ there.

This script, run in the build container only (CPU, ~10 min on 8 cores; mpmath 1.3 in pure Python):
  1. computes P(t) for every (class, branch) of a case three ways in fp64 (LAPACK route via numpy, oracle Jacobi,
     engine QL via crab_spectral_decompose) and once at 60 decimal digits (mpmath.eigsy on the same lower triangle
     the reference's eigh(UPLO::Lower) reads; truth kept as double-double);
  2. runs the oracle's pruning on each fp64 route's matrices, lists the sites on which any two routes differ by more
     than 1e-10 relative (the tail), and prunes exactly those sites at 60 digits;
  3. writes tests/golden/parity_tail.npz: per case the tail sites, ln f per route and the 60-digit truth, the largest
     absolute error of each route's P entries, and a handful of control sites.

tests/test_oracle.py (CPU) asserts from the fixture that all three fp64 routes sit within one and the same bound of the
truth; tests/test_gpu_parity.py (-m gpu) asserts that the device result is no farther from the truth than the LAPACK
route on these sites, and replaces its former bare 1e-7 allowances by the per-site bound described in tests/util.py.

    python tests/golden/make_parity_tail.py [case ...]      # cases: c2 lysin synthetic_edge free_ratio_60
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DPS = 60
MAX_TAIL = 24          # worst sites kept per case
N_CONTROL = 6          # well-conditioned sites kept per case
LN_TINY = float(np.log(1e-300))


# ------------------------------------------------------------------------------------------------------------------
# fp64 routes
# ------------------------------------------------------------------------------------------------------------------
def route_lapack(Q, pi, ts):
    """The reference's route: S = Pi^1/2 Q Pi^-1/2, eigh on the lower triangle, U = Pi^-1/2 V, U^-1 by LU inverse,
    P = (U * exp(lam t)) U^-1 (matrix.rs:41-67,88-102)."""
    sq = np.sqrt(pi)
    S = Q * sq[:, None] / sq[None, :]
    lam, V = np.linalg.eigh(S, UPLO="L")
    U = V / sq[:, None]
    Ui = np.linalg.inv(U)
    return np.stack([(U * np.exp(lam * t)[None, :]) @ Ui for t in ts])


def route_jacobi(Q, pi, ts):
    from oracle import oracle as orc
    return np.stack([orc.matrix_exponential(Q, pi, float(t)) for t in ts])


def route_ql(Q, pi, ts):
    """The engine's host decomposition; the product is kernel 1's on the device (same operands, MFMA summation order)."""
    from crabml_amd.engine import spectral_decompose
    U, lam, Ui = spectral_decompose(Q, pi)
    return np.stack([(U * np.exp(lam * t)[None, :]) @ Ui for t in ts])


ROUTES = (("lapack", route_lapack), ("jacobi", route_jacobi), ("ql", route_ql))


# ------------------------------------------------------------------------------------------------------------------
# 60-digit truth (worker processes)
# ------------------------------------------------------------------------------------------------------------------
def _mp():
    import mpmath
    mpmath.mp.dps = DPS
    return mpmath


def _to_obj(a):
    mp = _mp()
    return np.array([mp.mpf(float(x)) for x in np.asarray(a).ravel()], dtype=object).reshape(np.shape(a))


def truth_p_worker(args):
    """One eigensystem, many times: P(t) at DPS digits as (hi, lo) doubles, and each fp64 route's largest absolute
    entry error.  args = (Q, pi, ts, routes_P [R][len(ts)][n][n])."""
    mp = _mp()
    Q, pi, ts, routes_P = args
    n = len(pi)
    sq = [mp.sqrt(mp.mpf(float(x))) for x in pi]
    S = mp.matrix(n, n)
    for i in range(n):
        for j in range(i + 1):                      # lower triangle only, mirrored: eigh(UPLO::Lower)
            v = mp.mpf(float(Q[i, j])) * sq[i] / sq[j]
            S[i, j] = v
            S[j, i] = v
    lam, V = mp.eigsy(S)
    Vo = np.array([[V[i, k] for k in range(n)] for i in range(n)], dtype=object)
    hi = np.empty((len(ts), n, n))
    lo = np.empty((len(ts), n, n))
    err = np.zeros(len(routes_P))
    sqo = np.array(sq, dtype=object)
    for a, t in enumerate(ts):
        e = np.array([mp.exp(lam[k] * mp.mpf(float(t))) for k in range(n)], dtype=object)
        P = (Vo * e[None, :]).dot(Vo.T)
        P = P * (sqo[None, :] / sqo[:, None])       # P = Pi^-1/2 (V e V^T) Pi^1/2
        for i in range(n):
            for j in range(n):
                h = float(P[i, j])
                hi[a, i, j] = h
                lo[a, i, j] = float(P[i, j] - mp.mpf(h))
                for r in range(len(routes_P)):
                    d = abs(float(P[i, j] - mp.mpf(float(routes_P[r][a][i, j]))))
                    if d > err[r]:
                        err[r] = d
    return hi, lo, err


def truth_prune_worker(args):
    """ln f_{k,s} at DPS digits for one class and a list of sites.  args = (P_hi, P_lo [n_nodes][n][n], pi, parent,
    leaf_row_of_node, codes [n_rows][n_sel])."""
    mp = _mp()
    P_hi, P_lo, pi, parent, leaf_row, codes = args
    n_nodes, n = len(parent), len(pi)
    P = [None] * n_nodes
    for v in range(n_nodes):
        if parent[v] >= 0:
            P[v] = _to_obj(P_hi[v]) + _to_obj(P_lo[v])
    kids = [[] for _ in range(n_nodes)]
    root = -1
    for v in range(n_nodes):
        if parent[v] >= 0:
            kids[parent[v]].append(v)
        else:
            root = v
    order, stack = [], [root]
    while stack:                                     # parents before children; reversed below
        v = stack.pop()
        order.append(v)
        stack.extend(kids[v])
    pio = _to_obj(pi)
    ones = np.array([mp.mpf(1)] * n, dtype=object)
    out = []
    for s in range(codes.shape[1]):
        L = [None] * n_nodes
        for v in reversed(order):
            if leaf_row[v] >= 0:
                c = int(codes[leaf_row[v], s])
                if 0 <= c < n:
                    L[v] = np.array([mp.mpf(1) if i == c else mp.mpf(0) for i in range(n)], dtype=object)
                else:
                    L[v] = ones.copy()
                continue
            acc = ones.copy()
            for c in kids[v]:
                if leaf_row[c] >= 0:
                    code = int(codes[leaf_row[c], s])
                    m = P[c][:, code] if 0 <= code < n else P[c].dot(ones)
                else:
                    m = P[c].dot(L[c])
                acc = acc * m
            L[v] = acc
        f = (pio * L[root]).sum()
        out.append(mp.nstr(f, 40))
    return out


# ------------------------------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------------------------------
def load_cases(which):
    from crabml_amd import synth
    from util import load_case
    cases = {}
    if "c2" in which:
        c = synth.make_case("M8", 100, 10000, seed=2)
        cases["c2"] = dict(kind="classes", Q=c.Q, props=np.asarray(c.proportions), pi=c.pi, parent=np.asarray(c.parent), bl=c.branch_lengths,
                           codes=c.codes, note="synth.make_case('M8', 100, 10000, seed=2): BASELINE configs[1]")
    for name, key in (("lysin_m2a_gaps", "lysin"), ("synthetic_edge", "synthetic_edge")):
        if key in which:
            c = load_case(name)
            cases[key] = dict(kind="classes", Q=c["Q"], props=c["proportions"], pi=c["pi"], parent=c["parent"], bl=c["branch_lengths"],
                              codes=c["sequences"], note=f"tests/golden/{name}.npz")
    if "free_ratio_60" in which:
        c = synth.make_case("free-ratio", 60, 1500, seed=22)
        cases["free_ratio_60"] = dict(kind="branch", Q=c.Q_per_node, props=np.array([1.0]), pi=c.pi, parent=np.asarray(c.parent),
                                      bl=c.branch_lengths, codes=c.codes, note="synth.make_case('free-ratio', 60, 1500, seed=22)")
    return cases


def mixture(lnf, props):
    """Site-class mixture as the reference takes it (likelihood.rs:349-370), on [S, K] ln f."""
    x = np.maximum(lnf, LN_TINY)
    with np.errstate(divide="ignore"):
        x = x + np.log(props)[None, :]
    mx = x.max(axis=1)
    return mx + np.log(np.exp(x - mx[:, None]).sum(axis=1))


def run_case(name, c, pool):
    from oracle import oracle as orc
    mp = _mp()
    pi, parent, bl, codes, props = c["pi"], c["parent"], c["bl"], np.asarray(c["codes"]), np.asarray(c["props"], dtype=float)
    n_nodes, n = len(parent), len(pi)
    ts = [(v, None if parent[v] < 0 else int(parent[v])) for v in range(n_nodes)]
    leaf = np.arange(codes.shape[0], dtype=np.int32)
    K = 1 if c["kind"] == "branch" else len(c["Q"])
    # eigensystems and the branches each one serves: (class k, nodes)
    if c["kind"] == "classes":
        units = [(k, c["Q"][k], list(range(n_nodes))) for k in range(K)]
    else:
        units = [(0, c["Q"][v], [v]) for v in range(n_nodes)]
    print(f"[{name}] {K} class(es), {n_nodes} nodes, {codes.shape[1]} sites, {len(units)} eigensystems", flush=True)

    # ---- fp64 routes ----
    P = {r: np.zeros((K, n_nodes, n, n)) for r, _ in ROUTES}
    for k, Q, nodes in units:
        tt = [0.0 if parent[v] < 0 else float(bl[v]) for v in nodes]
        for r, fn in ROUTES:
            P[r][k, nodes] = fn(np.asarray(Q), pi, tt)
    site = {}
    for r, _ in ROUTES:
        lnf = orc.site_lnf_from_transition_matrices(P[r], pi, ts, codes, leaf)
        site[r] = mixture(lnf, props)
    names = [r for r, _ in ROUTES]
    pair = np.zeros(codes.shape[1])
    for i in range(3):
        for j in range(i + 1, 3):
            a, b = site[names[i]], site[names[j]]
            pair = np.maximum(pair, np.abs(a - b) / np.maximum(np.abs(b), 1e-3))
    order = np.argsort(-pair)
    n_tail_all = int((pair > 1e-10).sum())
    tail = order[: min(MAX_TAIL, max(n_tail_all, 4))]
    rng = np.random.default_rng(12345)
    calm = np.flatnonzero(pair <= 1e-12)
    control = rng.choice(calm, size=min(N_CONTROL, len(calm)), replace=False) if len(calm) else np.array([], dtype=int)
    sel = np.concatenate([tail, control]).astype(int)
    print(f"[{name}] sites with any two routes > 1e-10 apart: {n_tail_all} of {codes.shape[1]}; keeping {len(tail)} + {len(control)} controls; "
          f"largest pairwise difference {pair.max():.3e}", flush=True)

    # ---- truth P for every (class, branch) ----
    jobs = []
    for k, Q, nodes in units:
        tt = [0.0 if parent[v] < 0 else float(bl[v]) for v in nodes]
        jobs.append((np.asarray(Q), pi, tt, [P[r][k, nodes] for r in names]))
    P_hi = np.zeros((K, n_nodes, n, n))
    P_lo = np.zeros((K, n_nodes, n, n))
    p_err = np.zeros(3)
    for (k, Q, nodes), (hi, lo, err) in zip(units, pool.map(truth_p_worker, jobs, chunksize=1)):
        P_hi[k, nodes], P_lo[k, nodes] = hi, lo
        p_err = np.maximum(p_err, err)
    print(f"[{name}] largest |P_route - P_truth| per entry: " + ", ".join(f"{r} {e:.2e}" for r, e in zip(names, p_err)), flush=True)

    # ---- truth pruning of the selected sites ----
    leaf_row = np.full(n_nodes, -1)
    leaf_row[: codes.shape[0]] = np.arange(codes.shape[0])
    chunks = np.array_split(np.arange(len(sel)), max(1, min(len(sel), 1 if K >= 8 else 8 // K)))
    jobs, keys = [], []
    for k in range(K):
        for ch in chunks:
            jobs.append((P_hi[k], P_lo[k], pi, parent, leaf_row, codes[:, sel[ch]]))
            keys.append((k, ch))
    f_truth = [[None] * K for _ in sel]
    for (k, ch), res in zip(keys, pool.map(truth_prune_worker, jobs, chunksize=1)):
        for i, s in zip(ch, res):
            f_truth[i][k] = mp.mpf(s)
    truth = np.empty(len(sel))
    tiny = mp.mpf("1e-300")
    for i in range(len(sel)):
        xs = []
        for k in range(K):
            if props[k] > 0:
                fk = f_truth[i][k] if f_truth[i][k] > tiny else tiny
                xs.append(mp.log(fk) + mp.log(mp.mpf(float(props[k]))))
        mx = max(xs)
        truth[i] = float(mx + mp.log(sum(mp.exp(x - mx) for x in xs)))
    out = {f"{name}__sites": sel.astype(np.int64), f"{name}__n_tail": np.int64(len(tail)), f"{name}__n_tail_all": np.int64(n_tail_all),
           f"{name}__n_sites": np.int64(codes.shape[1]), f"{name}__truth": truth, f"{name}__p_abs_err": p_err,
           f"{name}__routes": np.array(names), f"{name}__note": np.array(c["note"])}
    for r in names:
        out[f"{name}__lnf_{r}"] = site[r][sel]
    rel = {r: np.abs(site[r][sel] - truth) / np.abs(truth) for r in names}
    print(f"[{name}] tail sites, |route - truth| / |truth|: " + ", ".join(f"{r} max {rel[r][:len(tail)].max():.2e}" for r in names), flush=True)
    if len(control):
        print(f"[{name}] control sites: " + ", ".join(f"{r} max {rel[r][len(tail):].max():.2e}" for r in names), flush=True)
    return out


def main():
    which = sys.argv[1:] or ["lysin", "synthetic_edge", "free_ratio_60", "c2"]
    from oracle import oracle as orc
    orc.build()
    cases = load_cases(which)
    path = os.path.join(HERE, "parity_tail.npz")
    out = {}
    if os.path.exists(path):
        z = np.load(path, allow_pickle=False)
        out = {k: z[k] for k in z.files}
    with ProcessPoolExecutor(max_workers=8) as pool:
        for name, c in cases.items():
            out.update(run_case(name, c, pool))
            np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
