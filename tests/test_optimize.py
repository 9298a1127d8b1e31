"""The optimiser mirror (crabml_amd/optimize.py): the gradient stencil against scipy's own, on the CPU; the fits on
the device (marked gpu) against PAML's optima and against the sequential form the reference uses."""
import os

import numpy as np
import pytest

from util import DATA

LYSOZYME_START = ("((Hsa_Human: 0.05, Hla_gibbon: 0.05): 0.05, ((Cgu/Can_colobus: 0.05, Pne_langur: 0.05): 0.05, "
                  "Mmu_rhesus: 0.05): 0.05, (Ssc_squirrelM: 0.05, Cja_marmoset: 0.05): 0.05);")


def _rosen(x):
    return float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2) + np.sum(np.cos(3 * x)))


def test_forward_difference_stencil_is_scipys_bit_for_bit():
    """What minimize(method='L-BFGS-B', jac=None) computes (approx_derivative, '2-point', abs_step = eps, bounds) and
    what value_and_gradient computes from one batched call: same points, same steps, same quotients."""
    from scipy.optimize._numdiff import approx_derivative
    from crabml_amd.optimize import forward_difference_points, value_and_gradient
    rng = np.random.default_rng(0)
    bounds = [(-2.0, 2.0), (-1.0, 3.0), (None, 1.5), (0.0, None), (-5e-9, 5e-9), (-4.0, 4.0)]
    lb = np.array([-np.inf if b[0] is None else b[0] for b in bounds])
    ub = np.array([np.inf if b[1] is None else b[1] for b in bounds])
    cases = [rng.uniform(-0.5, 0.5, 6) for _ in range(20)]
    cases += [np.array([2.0, 3.0, 1.5, 0.0, 5e-9, -4.0]),            # on the bounds: steps flip
              np.array([2.0 - 5e-9, -1.0, 1.5 - 1e-9, 1e3, 0.0, 4.0 - 1e-8]),
              np.array([1.0, 1e9, -1e12, 1e17, 0.0, 0.0])]          # steps lost to rounding fall back to relative ones
    cases[-1][3] = 1e17
    for x in cases:
        x = np.clip(x, lb, ub) if not np.any(np.abs(x) > 1e6) else x
        use_bounds = not np.any(np.abs(x) > 1e6)
        seen = []

        def fun(v):
            seen.append(v.copy())
            return _rosen(v)

        ref = approx_derivative(fun, x, method="2-point", abs_step=1e-8, f0=_rosen(x), bounds=(lb, ub) if use_bounds else (-np.inf, np.inf))
        pts, dx = forward_difference_points(x, bounds if use_bounds else None)
        np.testing.assert_array_equal(pts, np.array(seen))
        f, g = value_and_gradient(lambda P: np.array([_rosen(p) for p in P]), x, bounds if use_bounds else None)
        assert f == _rosen(x)
        np.testing.assert_array_equal(g, ref)


def test_batched_gradient_gives_scipys_own_trajectory():
    from scipy.optimize import minimize
    from crabml_amd.optimize import value_and_gradient
    bounds = [(-1.5, 1.5)] * 5 + [(0.2, 0.9)]
    x0 = np.array([-1.0, 1.2, 0.3, -0.2, 1.5, 0.9])
    calls = []

    def batch(P):
        calls.append(len(P))
        return np.array([_rosen(p) for p in P])

    a = minimize(_rosen, x0, method="L-BFGS-B", bounds=bounds, options={"maxiter": 200})
    b = minimize(lambda x: value_and_gradient(batch, x, bounds), x0, jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": 200})
    np.testing.assert_array_equal(a.x, b.x)
    assert a.fun == b.fun and a.nit == b.nit
    assert set(calls) == {7} and a.nfev == 7 * len(calls)          # n + 1 sequential evaluations became one call each


def _lysozyme():
    from crabml_amd.io import Alignment, Tree
    return Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"), seqtype="codon"), Tree.from_newick(LYSOZYME_START)


@pytest.mark.gpu
def test_m0_fit_reaches_pamls_optimum_and_equals_the_sequential_fit():
    """tests/test_reference/test_paml_m0.py:30-72 (the reference accepts |lnL - PAML| < 10; a converged fit is within
    1e-3).  The batched-gradient fit and the reference's form of it (scipy differentiating through n + 1 single calls)
    see bit-identical likelihoods, so they take the same path to the same point."""
    from crabml_amd.optimize import M0Optimizer
    aln, tree = _lysozyme()
    opt = M0Optimizer(aln, tree, use_f3x4=True, optimize_branch_lengths=True)
    kappa, omega, lnl = opt.optimize(init_kappa=2.0, init_omega=0.4)
    assert abs(lnl - (-906.017440)) < 1e-3
    assert abs(kappa - 4.54008) < 5e-2 and abs(omega - 0.80663) < 5e-3          # lysozyme M0 estimates, PAML's mlc
    assert opt.n_vectors == opt.n_calls * (2 + opt.n_branches + 1) and opt.n_branches == 11
    assert opt.history[-1]["log_likelihood"] >= opt.history[0]["log_likelihood"]
    aln2, tree2 = _lysozyme()
    seq = M0Optimizer(aln2, tree2, batched_gradient=False)
    k2, w2, l2 = seq.optimize(init_kappa=2.0, init_omega=0.4)
    assert (k2, w2, l2) == (kappa, omega, lnl)
    np.testing.assert_array_equal(seq.result.x, opt.result.x)
    assert seq.n_calls == opt.n_vectors and seq.result.nit == opt.result.nit
    # the tree keeps the lengths of the last point evaluated: the optimum's neighbourhood for both, but the sequential
    # form (like the reference) ends on a finite-difference neighbour, one length off by its 1e-8 step
    np.testing.assert_allclose(tree.branch_length, tree2.branch_length, rtol=2e-8)
    assert np.abs(np.log(tree.branch_length[np.array(opt.branch_nodes)]) - opt.result.x[2:]).max() < 1e-6


@pytest.mark.gpu
def test_m1a_and_m2a_fits_reach_pamls_optima():
    """tests/test_reference/test_site_class_optimizers.py, with branch lengths free and the M0 start the reference
    defaults to; PAML: M1a -902.503872, M2a -899.998568 (tests/test_rust/test_paml_rust_validation.py:104,164).  L-BFGS-B with scipy's
    default stopping rule and a forward-difference gradient stops a few 1e-3 short of PAML's optimum on the mixtures
    (the reference's driver, unchanged); the reference's own tests only ask for a finite, improved likelihood."""
    from crabml_amd.optimize import M1aOptimizer, M2aOptimizer
    aln, tree = _lysozyme()
    m1 = M1aOptimizer(aln, tree)
    kappa, omega0, p0, l1 = m1.optimize()
    assert abs(l1 - (-902.503872)) < 2e-2 and 0 < omega0 < 1 and 0 < p0 < 1 and kappa > 0
    aln, tree = _lysozyme()
    m2 = M2aOptimizer(aln, tree)
    kappa, omega0, omega2, p0, p1, l2 = m2.optimize()
    assert abs(l2 - (-899.998568)) < 2e-2
    assert 0 < omega0 < 1 < omega2 and 0 < p0 < 1 and 0 < p1 < 1 and 0 < 1 - p0 - p1 < 1
    assert l2 >= l1 - 1e-6
    # mixtures too: batched and sequential gradients walk the same path from the same start (with the M0 start they need
    # not: the sequential M0 fit, like the reference's, leaves the tree on a finite-difference neighbour of its optimum)
    runs = []
    for batched in (True, False):
        aln, tree = _lysozyme()
        o = M2aOptimizer(aln, tree, init_with_m0=False, batched_gradient=batched)
        runs.append((o.optimize(), o.result.x, o.result.nit))
    assert runs[0][0] == runs[1][0] and runs[0][2] == runs[1][2]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    # global branch scale instead of free branch lengths (test_site_class_optimizers.py:74-99)
    aln, tree = _lysozyme()
    m2s = M2aOptimizer(aln, tree, optimize_branch_lengths=False)
    out = m2s.optimize(maxiter=50)
    assert np.isfinite(out[-1]) and out[-1] <= l2 + 1e-6 and out[-1] >= m2s.history[0]["log_likelihood"]
    assert m2s.n_vectors == m2s.n_calls * 7


@pytest.mark.gpu
def test_m7_m8_m8a_fits_and_their_likelihood_ratio_tests():
    """The M7-M8 and M8a-M8 comparisons on lysozyme (BASELINE config 5 runs the first per gene).  PAML's optima:
    M7 -902.510018, M8 -899.999237, M8a -902.503869 (tests/test_rust/test_paml_rust_validation.py:387,445,509)."""
    from crabml_amd.optimize import M7Optimizer, M8aOptimizer, M8Optimizer
    fits = {}
    for cls in (M7Optimizer, M8Optimizer, M8aOptimizer):
        aln, tree = _lysozyme()
        opt = cls(aln, tree)
        fits[cls.__name__] = (opt.optimize(), opt)
    l7, l8, l8a = (fits[k][0][-1] for k in ("M7Optimizer", "M8Optimizer", "M8aOptimizer"))
    assert abs(l7 - (-902.510018)) < 5e-2 and abs(l8 - (-899.999237)) < 5e-2 and abs(l8a - (-902.503869)) < 5e-2
    assert l8 >= l7 - 1e-6 and l8 >= l8a - 1e-6                       # nested models
    assert abs(2 * (l8 - l7) - 2 * (-899.999237 + 902.510018)) < 0.1   # the test statistic PAML's numbers give
    kappa, p0, p_beta, q_beta, omega_s, _ = fits["M8Optimizer"][0]
    assert omega_s > 1 and 0 < p0 < 1 and p_beta > 0 and q_beta > 0 and kappa > 0
    o8 = fits["M8Optimizer"][1]
    assert o8.history and all(np.isfinite(h["log_likelihood"]) for h in o8.history)


def _lysozyme_large():
    from crabml_amd.io import Alignment, Tree
    return (Alignment.from_phylip(os.path.join(DATA, "lysozymeLarge_clean.nuc"), seqtype="codon"),
            Tree.from_newick(open(os.path.join(DATA, "lysozymeLarge.trees")).read()))


@pytest.mark.gpu
def test_branch_site_model_a_fit_against_paml():
    """tests/test_branch_site_optimization.py:23-86 (same start, same tolerances) and the null of
    tests/test_branch_site_null.py.  The reference's objective never reads the branch-length parameters (see the class
    docstring), so the 33 branch rows of each stencil collapse onto the centre row: 6 distinct vectors per gradient."""
    from crabml_amd.optimize import BranchSiteModelAOptimizer
    aln, tree = _lysozyme_large()
    start = tree.branch_length.copy()
    opt = BranchSiteModelAOptimizer(aln, tree, use_f3x4=True, optimize_branch_lengths=True)
    kappa, omega0, omega2, p0, p1, lnl = opt.optimize(init_kappa=3.0, init_omega0=0.05, init_omega2=3.0, init_p0=0.35, init_p1=0.30,
                                                      method="L-BFGS-B", maxiter=500)
    assert abs(lnl - (-1035.533916)) < 0.1
    assert abs(kappa - 4.154927) < 0.1 and abs(p0 - 0.326513) < 0.05 and abs(p1 - 0.269308) < 0.05
    assert 0 < omega0 < 0.01 and omega2 > 1.0 and abs(omega2 - 4.809765) < 0.5
    assert opt.n_branches == 33 and opt.n_rows == opt.n_calls * (5 + 33 + 1) and opt.n_vectors == opt.n_calls * 6
    # the lengths never moved; what the tree holds afterwards is the parameter vector's max(length, 0.001), written but not read
    moved = np.where(np.asarray(tree.parent) >= 0, np.maximum(start, 0.001), start)
    np.testing.assert_allclose(tree.branch_length, moved, rtol=0, atol=2e-8)
    assert np.all(opt.result.x[5:] == np.log(np.maximum(start[np.array(opt.branch_nodes)], 0.001)))
    # the sequential form, as the reference runs it: the same iterates, 39 calls per gradient instead of one
    aln2, tree2 = _lysozyme_large()
    seq = BranchSiteModelAOptimizer(aln2, tree2, batched_gradient=False)
    out2 = seq.optimize(init_kappa=3.0, init_omega0=0.05, init_omega2=3.0, init_p0=0.35, init_p1=0.30, maxiter=500)
    assert out2 == (kappa, omega0, omega2, p0, p1, lnl) and seq.n_calls == opt.n_rows
    # null model and the test between them
    aln3, tree3 = _lysozyme_large()
    null = BranchSiteModelAOptimizer(aln3, tree3, fix_omega=True)
    *_, w2_null, _, _, l0 = null.optimize(init_kappa=3.0, init_omega0=0.05, init_p0=0.35, init_p1=0.30, maxiter=500)
    assert w2_null == 1.0 and l0 <= lnl + 1e-6 and null.n_vectors == null.n_calls * 5
    # branch lengths let through: starts from max(length, 0.001) on the seven zero-length branches, ends near PAML's optimum too
    aln4, tree4 = _lysozyme_large()
    live = BranchSiteModelAOptimizer(aln4, tree4, live_branch_lengths=True)
    l_live = live.optimize(init_kappa=3.0, init_omega0=0.05, init_omega2=3.0, init_p0=0.35, init_p1=0.30, maxiter=500)[-1]
    assert abs(l_live - (-1035.533916)) < 0.2 and live.n_vectors == live.n_rows


@pytest.mark.gpu
def test_branch_model_fits_against_paml():
    """tests/test_branch_paml_validation.py:25-122: the two-ratio model on lysozyme (PAML -903.076551, kappa 4.56831,
    omega0 0.67535, omega1 at PAML's upper bound) with the reference test's tolerances, and the free-ratio model
    (one omega per branch; nests M0 and the two-ratio model)."""
    from crabml_amd.io import Alignment, Tree
    from crabml_amd.optimize import BranchModelOptimizer
    aln = Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"), seqtype="codon")
    two = "((Hsa_Human,Hla_gibbon) #1, ((Cgu/Can_colobus,Pne_langur), Mmu_rhesus), (Ssc_squirrelM,Cja_marmoset));"
    tree = Tree.from_newick(two)
    opt = BranchModelOptimizer(aln, tree, use_f3x4=True, optimize_branch_lengths=True, free_ratio=False)
    kappa, omega, lnl = opt.optimize(init_kappa=3.0, init_omega=0.4, method="L-BFGS-B", maxiter=1000)
    assert abs(lnl - (-903.076551)) < 0.5 and abs(kappa - 4.56831) < 0.1 and abs(omega["omega0"] - 0.67535) < 0.05
    assert omega["omega1"] > 5.0 and set(omega) == {"omega0", "omega1"}
    assert opt.n_vectors == opt.n_calls * (1 + 2 + 11 + 1)
    # the fitted lengths sit on the nodes they were evaluated on: re-evaluating from the tree reproduces the optimum
    again = BranchModelOptimizer(aln, tree, free_ratio=False)
    x = np.concatenate([opt.result.x[:3], np.log(tree.branch_length[np.array(again.calc.postorder_nodes[:-1])])])
    assert abs(-again.compute_log_likelihood(x) - lnl) < 1e-9
    seq = BranchModelOptimizer(aln, Tree.from_newick(two), batched_gradient=False)
    assert seq.optimize(init_kappa=3.0, init_omega=0.4, maxiter=1000) == (kappa, omega, lnl)
    free = BranchModelOptimizer(aln, Tree.from_newick(two), free_ratio=True)
    k_f, omega_f, l_f = free.optimize(maxiter=500)
    assert len(omega_f) == 11 and l_f >= lnl - 0.05 and l_f >= -906.017440 and 2.0 < k_f < 10.0
    with pytest.raises(NotImplementedError):
        BranchModelOptimizer(aln, Tree.from_newick(two), optimize_branch_lengths=False)


def _lysin():
    from crabml_amd.io import Alignment, Tree
    aln = Alignment.from_fasta(os.path.join(DATA, "lysin.fasta"), seqtype="codon")
    lines = [ln.strip() for ln in open(os.path.join(DATA, "lysin.trees")) if ln.strip()]
    text = "".join(ln for ln in lines[1:] if not ln.startswith("#")).replace(": ", ":").replace(" #1", "")   # test_lysin.py:65-74
    return aln, Tree.from_newick(text)


def test_least_squares_starting_lengths_match_the_reference():
    """optimize/distance_init.py on the lysin example (a tree without branch lengths), against the reference's own output
    (tests/golden/make_golden_distance_init.py).  Distances are exact; the lengths come out of an L-BFGS-B run that the
    reference stops at ftol = 1e-4, so they are a starting point and agree to the optimiser's own accuracy."""
    from crabml_amd.distance_init import compute_pairwise_distances, initialize_branch_lengths_ls, path_incidence
    from util import load_case
    c = load_case("lysin_distance_init")
    aln, tree = _lysin()
    np.testing.assert_array_equal(compute_pairwise_distances(aln), c["pairwise"])
    branch_nodes = [v for v in tree.postorder() if tree.parent[v] >= 0]
    assert [tree.name[v] or "" for v in branch_nodes] == [str(s) for s in c["branch_node_names"]]
    A = path_incidence(tree, branch_nodes)
    assert A.shape == (25 * 24 // 2, 47) and A.sum(axis=1).min() >= 2          # every pair is joined through at least two branches
    b = initialize_branch_lengths_ls(aln, tree, branch_nodes)
    np.testing.assert_allclose(b, c["branch_lengths"], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_lysin_fits_from_a_tree_without_branch_lengths():
    """tests/test_paml_examples/test_lysin.py:83-154 -- 25 abalone sperm lysin sequences with alignment gaps, topology
    only: least-squares starting lengths, then M0 / M1a / M2a to within 0.01 of PAML's lnL, as the reference asks."""
    from crabml_amd.optimize import M0Optimizer, M1aOptimizer, M2aOptimizer
    aln, tree = _lysin()
    _, _, l0 = M0Optimizer(aln, tree, use_f3x4=True).optimize(maxiter=1000)
    assert abs(l0 - (-4682.373100)) < 0.01
    aln, tree = _lysin()
    *_, l1 = M1aOptimizer(aln, tree, use_f3x4=True).optimize(maxiter=1000)
    assert abs(l1 - (-4525.413236)) < 0.01
    aln, tree = _lysin()
    *_, l2 = M2aOptimizer(aln, tree, use_f3x4=True).optimize(maxiter=1000)
    assert abs(l2 - (-4464.511376)) < 0.01


@pytest.mark.gpu
def test_m4_m5_m6_m9_fits_reach_pamls_optima():
    """The remaining site-class optimisers (optimizer.py:1415-2100; none of them takes an M0 start) on lysozyme, against
    the likelihoods PAML reports at its optima (tests/test_rust/test_paml_rust_validation.py:569,627,689,745)."""
    from crabml_amd.optimize import M4Optimizer, M5Optimizer, M6Optimizer, M9Optimizer
    paml = {M5Optimizer: -900.222360, M9Optimizer: -899.997270, M4Optimizer: -900.139530, M6Optimizer: -900.167515}
    for cls, want in paml.items():
        aln, tree = _lysozyme()
        opt = cls(aln, tree)
        out = opt.optimize(maxiter=500)
        # M4's optimum has two proportions at zero, i.e. logits at the -10 bound: the driver stops 0.3 short of it from this start
        assert abs(out[-1] - want) < (0.5 if cls is M4Optimizer else 0.05), (cls.__name__, out[-1], want)
        assert out[-1] <= want + 0.01
        assert opt.n_vectors == opt.n_calls * (cls.n_model + 11 + 1)
    kappa, props, _ = out if cls is M4Optimizer else M4Optimizer(*_lysozyme()).optimize(maxiter=5)
    assert len(props) == 5 and abs(sum(props) - 1.0) < 1e-12 and kappa > 0
    with pytest.raises(ValueError, match="exactly 5"):
        M4Optimizer(*_lysozyme()).optimize(init_proportions=[0.5, 0.5])


@pytest.mark.gpu
def test_m3_fit_reaches_pamls_optimum():
    """optimizer.py:604-864 on lysozyme: PAML's M3 (K = 3) optimum is -899.985262
    (tests/test_rust/test_paml_rust_validation.py:223); omegas come back sorted with their proportions (sortwM3)."""
    from crabml_amd.optimize import M3Optimizer
    aln, tree = _lysozyme()
    opt = M3Optimizer(aln, tree, n_classes=3)
    kappa, omegas, props, lnl = opt.optimize()
    assert abs(lnl - (-899.985262)) < 0.05 and lnl <= -899.985262 + 0.01
    assert omegas == sorted(omegas) and len(omegas) == len(props) == 3 and abs(sum(props) - 1.0) < 1e-12 and kappa > 0
    aln, tree = _lysozyme()
    two = M3Optimizer(aln, tree, n_classes=3)
    *_, best = two.optimize(n_restarts=2, maxiter=200)
    assert best >= lnl - 0.05
    # PAML's starting values with rndu() = 0.5, and SetxInitials pushing values off the bounds
    x0, k0, w0, l0 = opt._build_init_params(3, 2.0, 0.4)
    assert k0 == 0.1 + 2.0 and l0 == [0.5, 0.5] and np.allclose(w0, [0.4 * 0.5, 0.4 * (0.5 + 2 / 3), 0.4 * (0.5 + 4 / 3)])
    b = opt._build_bounds(3)
    assert len(b) == len(x0) == 1 + 2 + 3 + 11 and b[0] == (1e-4, 999.0) and b[1] == (-99.0, 99.0) and b[3] == (1e-6, 999.0)
    p = np.array([1e-4, 0.0, 1000.0])
    M3Optimizer._enforce_bounds(p, [(1e-4, 999.0), (-99.0, 99.0), (1e-6, 999.0)])
    assert p[0] == 1e-4 * 1.05 and p[1] == 0.0 and p[2] == 999.0 / 1.05


@pytest.mark.gpu
def test_hiv_nssites_fits_against_paml():
    """tests/test_paml_examples/test_hiv_nssites.py:76-230 -- PAML's HIV-1 env example (13 sequences, 91 codons), the six
    site models the reference checks, each to within 0.01 of PAML's lnL with the default starting values."""
    from crabml_amd.io import Alignment, Tree
    from crabml_amd import optimize as O
    lines = [ln.strip() for ln in open(os.path.join(DATA, "HIVenvSweden.trees")) if ln.strip()]
    text = lines[1].replace(": ", ":")
    paml = {"M0": -1137.688190, "M1a": -1114.641736, "M2a": -1106.445004, "M7": -1115.395312, "M8": -1106.388268, "M8a": -1114.579213}
    got = {}
    for name, want in paml.items():
        aln = Alignment.from_fasta(os.path.join(DATA, "HIVenvSweden.fasta"), seqtype="codon")
        out = getattr(O, name + "Optimizer")(aln, Tree.from_newick(text), use_f3x4=True).optimize()
        got[name] = out[-1]
        assert abs(out[-1] - want) < 0.01, (name, out[-1], want)
    assert got["M0"] < got["M1a"] < got["M2a"] and got["M7"] < got["M8"] and got["M8a"] < got["M8"]


def test_api_loaders_and_model_names():
    """api.py:908-1018, 1156-1165: format detection by suffix with the other format as fallback, a Newick string or a
    file for the tree, objects passed through, and the error for an unknown model name (raised before any device work)."""
    from crabml_amd import api
    from crabml_amd.io import Alignment, Tree
    aln = api._load_alignment(os.path.join(DATA, "lysozymeSmall.txt"))
    assert (aln.n_species, aln.n_sites) == (7, 130) and api._load_alignment(aln) is aln
    assert api._load_alignment(os.path.join(DATA, "lysin.fasta")).n_species == 25
    assert api._load_alignment(os.path.join(DATA, "lysozymeLarge_clean.nuc")).n_species == 19      # unknown suffix: FASTA fails, PHYLIP reads it
    with pytest.raises(FileNotFoundError):
        api._load_alignment(os.path.join(DATA, "no_such_file.fasta"))
    t = api._load_tree(LYSOZYME_START)
    assert t.n_leaves == 7 and api._load_tree(t) is t
    assert api._load_tree(os.path.join(DATA, "lysozymeLarge.trees")).n_leaves == 19
    with pytest.raises(ValueError, match="Failed to parse tree"):
        api._load_tree("((a,b")
    with pytest.raises(ValueError, match="Unknown model: 'M99'"):
        api.optimize_model("M99", aln, t)
    with pytest.raises(ValueError, match="Unknown test: 'M0_vs_M3'"):
        api.positive_selection(aln, t, test="M0_vs_M3")


@pytest.mark.gpu
def test_optimize_model_and_the_tests_for_positive_selection():
    """BASELINE configs[0]: ``optimize_model("M0", ...)`` on the bundled lysozyme files, by path; parameter names and
    counts as api.py:615-837 reports them (PAML's np = 13 for M0 on this tree); the M7-M8 and M8a-M8 tests."""
    from scipy.stats import chi2
    from crabml_amd import api
    res = api.optimize_model("m0", os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START)
    assert res.model_name == "M0" and abs(res.lnL - (-906.017440)) < 1e-3 and abs(res.omega - 0.80663) < 5e-3 and res.n_params == 13
    assert set(res.params) == {"omega"} and res.to_dict()["tree"].count(":") == 11 and res.convergence_info["iterations"] > 0
    m2 = api.optimize_model("M2a", os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START, maxiter=300)
    assert set(m2.params) == {"p0", "p1", "p2", "omega0", "omega2"} and m2.n_params == 16 and m2.omega is None
    assert abs(m2.params["p0"] + m2.params["p1"] + m2.params["p2"] - 1.0) < 1e-12 and abs(m2.lnL - (-899.998568)) < 2e-2
    m3 = api.optimize_model("M3", os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START, n_classes=2, init_with_m0=False)
    assert len(m3.params["omegas"]) == 2 and m3.n_params == 1 + 1 + 2 + 11
    t78 = api.m7_vs_m8(os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START)
    assert abs(t78["LRT"] - 2 * (-899.999237 + 902.510018)) < 0.1 and t78["df"] == 2 and t78["pvalue"] == chi2.sf(t78["LRT"], 2)
    assert set(t78["alternative"].params) == {"p0", "p", "q", "omega_s"} and set(t78["null"].params) == {"p", "q"}
    t8a = api.m8a_vs_m8(os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START)
    assert t8a["pvalue"] == 0.5 * chi2.sf(t8a["LRT"], 1) and t8a["null"].params["omega_s"] == 1.0 and t8a["null"].n_params == 15
    both = api.positive_selection(os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START)
    assert set(both) == {"M1a_vs_M2a", "M7_vs_M8"} and abs(both["M1a_vs_M2a"]["LRT"] - 2 * (-899.998568 + 902.503872)) < 0.1
    assert both["M7_vs_M8"]["LRT"] == t78["LRT"]                                # deterministic from the same start
    assert api.positive_selection(os.path.join(DATA, "lysozymeSmall.txt"), LYSOZYME_START, test="M8a-vs-M8")["df"] == 1


@pytest.mark.gpu
def test_branch_and_branch_site_models_by_name():
    """api.py:1214-1434: ``optimize_branch_model`` (M0 start, then the two-ratio fit) and ``optimize_branch_site_model``
    with its null, on the reference's example files; names, params keys and counts as its parsers report them."""
    from crabml_amd import api
    two = "((Hsa_Human:0.05,Hla_gibbon:0.05) #1:0.05, ((Cgu/Can_colobus:0.05,Pne_langur:0.05):0.05, Mmu_rhesus:0.05):0.05, " \
          "(Ssc_squirrelM:0.05,Cja_marmoset:0.05):0.05);"
    r = api.optimize_branch_model("multi-ratio", os.path.join(DATA, "lysozymeSmall.txt"), two, maxiter=1000)
    assert r.model_name == "Multi-ratio" and abs(r.lnL - (-903.076551)) < 0.5 and set(r.params["omega_dict"]) == {"omega0", "omega1"}
    assert r.n_params == 1 + 2 + 11
    with pytest.raises(ValueError, match="Unknown branch model"):
        api.optimize_branch_model("three-ratio", os.path.join(DATA, "lysozymeSmall.txt"), two)
    kw = dict(init_kappa=3.0, init_omega0=0.05, init_p0=0.35, init_p1=0.30, maxiter=500)
    a = api.optimize_branch_site_model("model-a", os.path.join(DATA, "lysozymeLarge_clean.nuc"), os.path.join(DATA, "lysozymeLarge.trees"),
                                       init_omega2=3.0, **kw)
    n = api.optimize_branch_site_model("modela", os.path.join(DATA, "lysozymeLarge_clean.nuc"), os.path.join(DATA, "lysozymeLarge.trees"),
                                       fix_omega=True, **kw)
    assert a.model_name == "Branch-Site Model A" and abs(a.lnL - (-1035.533916)) < 0.1 and a.n_params == 5 + 33
    assert n.model_name.endswith("(null, ω₂=1)") and n.params["omega2"] == 1.0 and n.lnL <= a.lnL + 1e-6
    assert set(a.params) == {"omega0", "omega2", "p0", "p1"}
    with pytest.raises(ValueError, match="Unknown branch-site model"):
        api.optimize_branch_site_model("model-b", os.path.join(DATA, "lysozymeLarge_clean.nuc"), os.path.join(DATA, "lysozymeLarge.trees"))


@pytest.mark.gpu
def test_gene_scan_on_the_device():
    """crabml_amd/scan.py with real engines: three small genes over two engines sharing the card give, gene for gene,
    what one engine gives alone (fits are deterministic; engines are re-targeted between genes)."""
    import warnings
    from crabml_amd import scan
    from crabml_amd.engine import Engine
    from crabml_amd.io import Alignment, Tree
    aln = Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"), seqtype="codon")
    codes = np.asarray(aln.sequences)
    genes = [(Alignment.from_codes(aln.names, codes[:, lo:lo + 60]), Tree.from_newick(LYSOZYME_START)) for lo in (0, 35, 70)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                               # a negative statistic on 60 codons only warns
        rows = scan.scan(genes, null="M7", workers=2)
        with Engine(0) as eng:
            eng.set_eigensolver("device")                             # what scan's engines use when they share the card
            alone = [scan.lrt_gene(eng, a, t, null="M7") for a, t in genes]
        mixture = scan.scan(genes[:1], null="M8a", workers=1)
    for r, a in zip(rows, alone):
        assert "error" not in r and (r["lnL_null"], r["lnL_M8"], r["LRT"], r["pvalue"]) == (a["lnL_null"], a["lnL_M8"], a["LRT"], a["pvalue"])
        assert r["LRT"] >= 0 and 0 <= r["pvalue"] <= 1 and r["fits"]["M8"]["vectors"] > r["fits"]["M8"]["calls"]
    assert mixture[0]["null"] == "M8a" and 0 <= mixture[0]["pvalue"] <= 1
    assert genes[0][1].branch_length.max() == 0.05                   # the caller's trees are left alone


@pytest.mark.gpu
def test_large_tree_fits_need_no_clamp():
    """500 taxa with long branches: almost every site likelihood lies below 1e-300, so with the reference's clamp the
    objective is the constant n_sites * ln(1e-300), its finite-difference gradient is zero and L-BFGS-B 'converges'
    where it starts.  With ``no_clamp=True`` the optimisers see the device's rescaled likelihood: a real gradient,
    and a fit that moves."""
    from crabml_amd import optimize, synth
    case = synth.make_case("M0", 500, 128, seed=9, mean_branch=0.6)
    aln, tree = synth.as_alignment_and_tree(case)
    import copy
    results = {}
    for no_clamp in (False, True):
        opt = optimize.M0Optimizer(aln, copy.deepcopy(tree), no_clamp=no_clamp)
        b0, bb = opt._initial_branch_parameters()
        x0 = np.array([np.log(2.0), np.log(0.4)] + b0)
        bounds = list(opt.model_bounds) + bb
        f, g = optimize.value_and_gradient(opt.negative_log_likelihoods, x0, bounds)
        results[no_clamp] = (f, g)
        opt.calc.engine.close()
    f_c, g_c = results[False]
    f_x, g_x = results[True]
    assert abs(f_c - 128 * 690.7755278982137) < 1e-6 * f_c and np.count_nonzero(g_c) == 0
    assert f_x > 1.5 * f_c and np.isfinite(f_x) and np.abs(g_x).max() > 1.0
    assert np.count_nonzero(g_x) >= 0.99 * len(g_x)     # (a branch whose effect on lnL ~ 1.6e5 is below one ulp may difference to zero)
    opt = optimize.M0Optimizer(aln, copy.deepcopy(tree), no_clamp=True)
    kappa, omega, lnL = opt.optimize(maxiter=3)
    assert -lnL < f_x - 1.0 and opt.n_vectors > 3 * len(g_x)      # the fit moved, on whole gradients per call
    opt.calc.engine.close()
