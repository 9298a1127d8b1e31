"""Pin the CPU oracle (oracle/crab_oracle.c) before anything is checked against it.

Sources of truth, in order:
  * golden vectors produced by the reference's own Python calculator (tests/golden/make_golden.py),
  * the PAML known answers the reference's tests assert
    (tests/test_rust/test_paml_rust_validation.py, tests/test_branch_site_likelihood.py),
  * the per-pattern PAML ``lnf`` files shipped with the reference's test data,
  * exact identities for the two Rust-only paths (branch model, branch-site Model A).
"""
import numpy as np
import pytest
from scipy.linalg import expm

from util import (LYSOZYME_MODELS, SITE_MODEL_CASES, SITE_RTOL, TOTAL_ATOL, assert_by_class_close, leaf_ids, load_case,
                  tree_structure)


@pytest.mark.parametrize("name", SITE_MODEL_CASES)
def test_site_class_total_matches_python_reference(oracle, name):
    c = load_case(name)
    ts = tree_structure(c["parent"])
    lnL, site = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], ts, c["branch_lengths"],
                                                         None, c["sequences"], leaf_ids(c), return_sites=True)
    # The Python reference builds P(t) with scipy's Pade expm, the Rust path (restated by the oracle) with an
    # eigendecomposition.  The eigen route carries ~2e-15 *absolute* error per entry of P(t) whatever the
    # eigensolver (measured against 60-digit mpmath: Jacobi 1.8e-15, LAPACK eigh 1.3e-15, QL 2.3e-15; Pade 1e-16),
    # which is a 1e-8 *relative* error on the ~1e-9 entries a multi-step change across a short branch needs.
    # lysin_m2a_gaps (real sequences, random branch lengths) and synthetic_edge (random codons) have such sites;
    # the cases evaluated at maximum-likelihood estimates do not.
    ill = name in ("lysin_m2a_gaps", "synthetic_edge")
    assert abs(lnL - float(c["lnL_pyref"])) < (TOTAL_ATOL if ill else 1e-9)
    assert abs(site.sum() - lnL) < 1e-9
    # per-site mixture from the reference's per-class values
    ref = c["site_class_lnf_pyref"]
    with np.errstate(divide="ignore"):
        x = ref + np.log(c["proportions"])[None, :]
    mx = x.max(axis=1)
    ref_site = mx + np.log(np.exp(x - mx[:, None]).sum(axis=1))
    np.testing.assert_allclose(site, ref_site, rtol=1e-8 if ill else SITE_RTOL, atol=1e-13)  # atol: the all-missing column has ln f = 0


@pytest.mark.parametrize("model", LYSOZYME_MODELS)
def test_paml_known_answers(oracle, model):
    c = load_case(f"lysozyme_{model}")
    ts = tree_structure(c["parent"])
    lnL = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], ts, c["branch_lengths"], None,
                                                   c["sequences"], leaf_ids(c))
    # tolerances as in the reference's tests: 1e-6 for M0, 1e-5 relative for the mixtures; M6 is where the
    # reference's own discretisation sits 1.05e-4 from PAML (still inside its rtol 1e-5 * 900)
    tol = 1e-6 if model == "m0" else 1e-5 * abs(float(c["lnL_paml"])) + 1e-5
    assert abs(lnL - float(c["lnL_paml"])) < tol


def test_m0_entry_point_equals_single_class(oracle):
    c = load_case("lysozyme_m0")
    ts = tree_structure(c["parent"])
    a, sa = oracle.compute_log_likelihood(c["Q"][0], c["pi"], ts, c["branch_lengths"], None, c["sequences"], leaf_ids(c),
                                          return_sites=True)
    b, sb = oracle.compute_site_class_log_likelihood([c["Q"][0]], [1.0], c["pi"], ts, c["branch_lengths"], None,
                                                     c["sequences"], leaf_ids(c), return_sites=True)
    assert a == b and np.array_equal(sa, sb)
    assert abs(a - (-906.0174402566055)) < 1e-9  # the reference's Python calculator


@pytest.mark.parametrize("name", SITE_MODEL_CASES)
def test_by_class_matches_python_reference(oracle, name):
    c = load_case(name)
    ts = tree_structure(c["parent"])
    got = oracle.compute_site_log_likelihoods_by_class(list(c["Q"]), c["pi"], ts, c["branch_lengths"], None, c["sequences"],
                                                       leaf_ids(c))
    assert got.shape == c["site_class_lnf_pyref"].shape
    # see test_site_class_total_matches_python_reference for why two cases get the eigen-route allowance
    assert_by_class_close(got, c["site_class_lnf_pyref"], name, rtol=1e-7 if name in ("lysin_m2a_gaps", "synthetic_edge") else SITE_RTOL)


def test_hiv_m8a_paml_lnf(oracle):
    """HIV env, M8a at PAML's MLEs: lnL -1114.579213 and PAML's per-pattern lnf table
    (parameters printed with 6 decimals, hence 1e-4-level agreement)."""
    c = load_case("hiv_m8a")
    ts = tree_structure(c["parent"])
    lnL, site = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], ts, c["branch_lengths"],
                                                         None, c["sequences"], leaf_ids(c), return_sites=True)
    assert abs(lnL - float(c["lnL_paml"])) < 2e-3
    # class table of mlc_m8a:229-230
    np.testing.assert_allclose(c["omegas"], [0.00699, 0.01997, 0.03359, 0.04850, 0.06529, 0.08477, 0.10831, 0.13858,
                                             0.18216, 0.26769, 1.0], atol=6e-6)
    rows = _parse_lnf("HIVenvSweden_m8a.lnf")
    assert rows[:, 1].sum() == site.size and abs((rows[:, 1] * rows[:, 2]).sum() - float(c["lnL_paml"])) < 1e-3
    _match_lnf_multiset(site, rows, tol=2e-4)


def _parse_lnf(fname):
    import os
    from util import DATA
    rows = []
    for ln in open(os.path.join(DATA, fname)):
        p = ln.split()
        if len(p) >= 5 and p[0].isdigit() and p[1].isdigit():
            try:
                rows.append((int(p[0]), int(p[1]), float(p[2])))
            except ValueError:
                pass
    return np.array(rows)


def _match_lnf_multiset(site_lnl, rows, tol):
    """PAML orders patterns its own way; compare as multisets: every site value must have a PAML pattern
    value within tol, with multiplicities adding up."""
    ours = np.sort(site_lnl)
    theirs = np.sort(np.repeat(rows[:, 2], rows[:, 1].astype(int)))
    assert ours.size == theirs.size
    assert np.max(np.abs(ours - theirs)) < tol


def test_matrix_exponential(oracle):
    c = load_case("lysozyme_m0")
    Q, pi = c["Q"][0], c["pi"]
    for t in (0.0, 1e-4, 0.05, 0.7, 5.0):
        P = oracle.matrix_exponential(Q, pi, t)
        np.testing.assert_allclose(P, expm(Q * t), atol=2e-14, rtol=0)
        np.testing.assert_allclose(P.sum(axis=1), 1.0, atol=1e-13)
    # JC69 analytic (reference: tests/test_reference/test_matrix.py:21-45)
    a = 0.25
    Qj = np.full((4, 4), a) - 4 * a * np.eye(4) + 0 * np.eye(4)
    np.fill_diagonal(Qj, -3 * a)
    P = oracle.matrix_exponential(Qj, np.full(4, 0.25), 0.3)
    same = 0.25 + 0.75 * np.exp(-4 * a * 0.3)
    diff = 0.25 - 0.25 * np.exp(-4 * a * 0.3)
    np.testing.assert_allclose(P, np.where(np.eye(4) > 0, same, diff), atol=1e-15)


def test_branch_site_model_a_paml(oracle):
    """Rust-only path: PAML lnL -1035.533916 (tests/test_branch_site_likelihood.py:121-128) and the
    per-pattern lnf file (tests/data/paml_examples/lysozyme/lnf: 86 patterns / 130 sites)."""
    c = load_case("lysozymeLarge_modelA")
    ts = tree_structure(c["parent"])
    np.testing.assert_allclose(c["site_class_freqs"], [0.326513, 0.269308, 0.221492, 0.182687], atol=1e-6)
    lnL, site = oracle.compute_branch_site_log_likelihood(c["Q"][0], c["Q"][1], c["Q"][2], c["qfactors"][0], c["qfactors"][1],
                                                          float(c["omega0"]), float(c["omega2"]), c["site_class_freqs"],
                                                          c["pi"], ts, c["branch_lengths"], c["branch_labels"], None,
                                                          c["sequences"], leaf_ids(c), return_sites=True)
    assert abs(lnL - float(c["lnL_paml"])) < 1e-5
    rows = c["paml_lnf"]
    assert rows.shape[0] == 86 and rows[:, 1].sum() == 130
    _match_lnf_multiset(site, rows, tol=1e-4)  # PAML prints its MLEs with 6 decimals


def test_branch_site_all_background_is_two_class_mixture(oracle):
    """Identity: Model A with every label 0 equals the 2-class site mixture with Q(omega0)*qf_back,
    Q(1)*qf_back and weights [p0+p2a, p1+p2b]."""
    c = load_case("lysozymeLarge_modelA")
    ts = tree_structure(c["parent"])
    qb = float(c["qfactors"][0])
    labels = np.zeros_like(c["branch_labels"])
    f = c["site_class_freqs"]
    lnL, site = oracle.compute_branch_site_log_likelihood(c["Q"][0], c["Q"][1], c["Q"][2], qb, float(c["qfactors"][1]),
                                                          float(c["omega0"]), float(c["omega2"]), f, c["pi"], ts,
                                                          c["branch_lengths"], labels, None, c["sequences"], leaf_ids(c),
                                                          return_sites=True)
    ref, ref_site = oracle.compute_site_class_log_likelihood([c["Q"][0] * qb, c["Q"][1] * qb], [f[0] + f[2], f[1] + f[3]],
                                                             c["pi"], ts, c["branch_lengths"], None, c["sequences"],
                                                             leaf_ids(c), return_sites=True)
    np.testing.assert_allclose(site, ref_site, rtol=1e-12)
    assert abs(lnL - ref) < 1e-9


def test_branch_model_with_shared_omega_is_m0(oracle):
    """Identity: one Q per node, all equal, is the M0 likelihood (reference asserts it loosely in
    tests/test_branch_models.py:131-174)."""
    c = load_case("lysozyme_m0")
    ts = tree_structure(c["parent"])
    n_nodes = len(ts)
    Qs = np.repeat(c["Q"][:1], n_nodes, axis=0)
    lnL, site = oracle.compute_log_likelihood_branch(c["sequences"], c["parent"], leaf_ids(c), Qs, c["branch_lengths"],
                                                     c["pi"], return_sites=True)
    ref, ref_site = oracle.compute_log_likelihood(c["Q"][0], c["pi"], ts, c["branch_lengths"], None, c["sequences"],
                                                  leaf_ids(c), return_sites=True)
    assert lnL == ref and np.array_equal(site, ref_site)
    with pytest.raises(ValueError):
        oracle.compute_log_likelihood_branch(c["sequences"], c["parent"], leaf_ids(c), Qs[:-1], c["branch_lengths"], c["pi"])


@pytest.mark.parametrize("name", ["lysozyme_m8", "lysin_m2a_gaps", "synthetic_edge"])
def test_scaled_mode_equals_reference_mode(oracle, name):
    c = load_case(name)
    ts = tree_structure(c["parent"])
    a, sa = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], ts, c["branch_lengths"], None,
                                                     c["sequences"], leaf_ids(c), return_sites=True)
    b, sb = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], ts, c["branch_lengths"], None,
                                                     c["sequences"], leaf_ids(c), scaled=True, return_sites=True)
    np.testing.assert_allclose(sb, sa, rtol=1e-13)
    assert abs(a - b) < 1e-10


def test_zero_proportion_contributes_nothing(oracle):
    """M4 KAT carries two exact-zero proportions (tests/test_rust/test_paml_rust_validation.py:709): ln 0 = -inf
    must drop out of the log-sum-exp without producing NaN."""
    c = load_case("lysozyme_m4")
    assert (c["proportions"] == 0).sum() == 2
    ts = tree_structure(c["parent"])
    keep = c["proportions"] > 0
    full = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], ts, c["branch_lengths"], None,
                                                    c["sequences"], leaf_ids(c))
    part = oracle.compute_site_class_log_likelihood(list(c["Q"][keep]), c["proportions"][keep], c["pi"], ts,
                                                    c["branch_lengths"], None, c["sequences"], leaf_ids(c))
    assert np.isfinite(full) and abs(full - part) < 1e-10


def test_no_root_is_an_error(oracle):
    c = load_case("lysozyme_m0")
    ts = [(i, (i + 1) % len(c["parent"])) for i in range(len(c["parent"]))]
    with pytest.raises(ValueError, match="No root"):
        oracle.compute_log_likelihood(c["Q"][0], c["pi"], ts, c["branch_lengths"], None, c["sequences"], leaf_ids(c))


def test_codon_q_builder(oracle):
    """qmatrix.rs:82-181 properties plus equality with the Python-side builder."""
    from crabml_amd.models import build_codon_Q_matrix
    pi = np.arange(1, 62, dtype=float)
    pi /= pi.sum()
    Q = oracle.build_codon_q_matrix_py(2.0, 0.5, pi, None)
    np.testing.assert_allclose(Q.sum(axis=1), 0, atol=1e-12)
    assert abs(-(pi * np.diag(Q)).sum() - 1.0) < 1e-12
    np.testing.assert_allclose(pi[:, None] * Q, (pi[:, None] * Q).T, atol=1e-12)
    assert (Q[~np.eye(61, dtype=bool)] > 0).sum() == 526
    np.testing.assert_allclose(Q, build_codon_Q_matrix(2.0, 0.5, pi), rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(oracle.build_codon_q_matrix_py(3.1, 1.7, pi, 1.0), build_codon_Q_matrix(3.1, 1.7, pi, 1.0),
                               rtol=1e-13, atol=1e-16)


# ---------------------------------------------------------------------------------------------
# the tail of the per-site parity gate: three eigen routes against 60-digit arithmetic
# (fixture tests/golden/parity_tail.npz, made by tests/golden/make_parity_tail.py in the build container)
# ---------------------------------------------------------------------------------------------
def _tail_case(name):
    from crabml_amd import synth
    if name == "c2":
        c = synth.make_case("M8", 100, 10000, seed=2)
        return c.Q, np.asarray(c.proportions), c.pi, np.asarray(c.parent), c.branch_lengths, c.codes, False
    if name == "free_ratio_60":
        c = synth.make_case("free-ratio", 60, 1500, seed=22)
        return c.Q_per_node, np.array([1.0]), c.pi, np.asarray(c.parent), c.branch_lengths, c.codes, True
    c = load_case({"lysin": "lysin_m2a_gaps", "synthetic_edge": "synthetic_edge"}[name])
    return c["Q"], c["proportions"], c["pi"], c["parent"], c["branch_lengths"], c["sequences"], False


@pytest.mark.parametrize("name", ["c2", "lysin", "synthetic_edge", "free_ratio_60"])
def test_eigen_routes_against_60_digit_truth(oracle, name):
    """The reference's LAPACK route (eigh + inverse, matrix.rs:41-67), the oracle's Jacobi route and the engine's
    Householder + QL route, each followed by the oracle's pruning, against values computed at 60 decimal digits, on the
    sites where they differ most plus control sites.  (1) Every route's P(t) entries lie within EPS_P / 2 of the truth,
    over every matrix of the case: the premise of util.eigen_route_bound.  (2) All three routes AND the truth lie within
    that one per-site bound of each other: where the plain 1e-10 gate cannot hold for two correct evaluations, the
    bound says by how much, and nothing is farther out than it allows.  (3) The engine's route is as close to the truth
    as the reference's own."""
    import os
    from util import EPS_P, GOLDEN, eigen_route_bound
    z = np.load(os.path.join(GOLDEN, "parity_tail.npz"), allow_pickle=False)
    routes = [str(r) for r in z[f"{name}__routes"]]
    assert routes == ["lapack", "jacobi", "ql"]
    assert z[f"{name}__p_abs_err"].max() <= 0.5 * EPS_P                                    # (1)
    sites, truth = z[f"{name}__sites"], z[f"{name}__truth"]
    lnf = {r: z[f"{name}__lnf_{r}"] for r in routes}
    Q, props, pi, parent, bl, codes, per_node = _tail_case(name)
    ts, leaf = tree_structure(parent), np.arange(codes.shape[0], dtype=np.int32)
    K = 1 if per_node else len(Q)
    P = np.zeros((K, len(parent), len(pi), len(pi)))
    for k in range(K):
        for v in range(len(parent)):
            if per_node or v == 0:
                U, lam, Ui = oracle.cached_q(Q[v] if per_node else Q[k], pi)
            P[k, v] = (U * np.exp(lam * (0.0 if parent[v] < 0 else bl[v]))[None, :]) @ Ui
    sub = np.ascontiguousarray(codes[:, sites])

    def mix(x):
        x = np.maximum(x, np.log(1e-300))
        with np.errstate(divide="ignore"):
            x = x + np.log(props)[None, :]
        mx = x.max(axis=1)
        return mx + np.log(np.exp(x - mx[:, None]).sum(axis=1))

    centre = mix(oracle.site_lnf_from_transition_matrices(P, pi, ts, sub, leaf))
    bound = eigen_route_bound(oracle, P, pi, ts, sub, leaf, mix)
    slack = 1e-12 * np.maximum(np.abs(truth), 1e-3)                                        # fp64 pruning itself
    assert np.all(np.abs(centre - lnf["jacobi"]) <= bound + slack)      # same decomposition, numpy's product instead of the C loop's
    for r in routes:
        assert np.all(np.abs(lnf[r] - truth) <= bound + slack), r                          # (2)
    den = np.maximum(np.abs(truth), 1e-3)
    err = {r: (np.abs(lnf[r] - truth) / den).max() for r in routes}
    assert err["ql"] <= max(1.25 * err["lapack"], SITE_RTOL)                               # (3)
    n_tail = int(z[f"{name}__n_tail"])
    if n_tail < len(sites):                                                                # control sites: all routes far inside 1e-10
        for r in routes:
            assert (np.abs(lnf[r] - truth) / den)[n_tail:].max() <= 0.01 * SITE_RTOL
