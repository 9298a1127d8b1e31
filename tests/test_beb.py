"""BEB grid integration: the CPU restatement against the reference's own output, and the host mirror
of the reference's grid / rate-matrix construction.  CPU only; the device path is in test_gpu_parity.py."""
import numpy as np
import pytest

from util import load_case

PAML_SITES = [15, 17, 37, 41, 50, 62, 114, 126]                      # tests/data/paml_reference/lysozyme_m2a_out.txt:256-263
PAML_PR_W_GT_1 = [0.693, 0.695, 0.744, 0.920, 0.835, 0.578, 0.689, 0.577]


@pytest.fixture(scope="module")
def beb_case():
    return load_case("lysozyme_m2a_beb")


def test_grid_and_rate_matrices_match_the_reference(beb_case):
    from crabml_amd import beb
    c = beb_case
    cols, props = beb.paml_grid_m2a()
    assert cols.shape == props.shape == (10000, 3)
    w = np.array(beb.paml_omega_grid())
    np.testing.assert_array_equal(props[:, 0], c["grid_p0"])
    np.testing.assert_array_equal(props[:, 1], c["grid_p1"])
    np.testing.assert_array_equal(w[cols[:, 0]], c["grid_omega0"])
    np.testing.assert_array_equal(w[cols[:, 2]], c["grid_omega2"])
    assert np.all(cols[:, 1] == 10) and np.all(w[cols[:, 1]] == 1.0)
    np.testing.assert_allclose(props.sum(axis=1), 1.0, atol=1e-15)
    Q = beb.beb_rate_matrices(float(c["kappa"]), c["pi"], c["mle_proportions"], c["mle_omegas"])
    assert np.abs(Q - c["Q"]).max() <= 4e-15 * np.abs(c["Q"]).max()


def test_oracle_reproduces_the_reference_posteriors_and_pamls_table(beb_case):
    from crabml_amd import beb
    from oracle import beb as obeb
    c = beb_case
    cols, props = beb.paml_grid_m2a()
    post, lnfx, ln_fx = obeb.site_posteriors(c["site_log_liks"], cols, props)
    ref = c["posteriors_ref"]
    assert np.abs(post - ref).max() <= 1e-13
    big = ref > 1e-8
    assert np.abs(post[big] / ref[big] - 1.0).max() <= 1e-10
    np.testing.assert_allclose(post.sum(axis=1), 1.0, atol=1e-12)
    assert np.isfinite(ln_fx) and ln_fx >= lnfx.max()
    # PAML's own output, to its three printed decimals
    got = post[np.array(PAML_SITES) - 1, 2]
    assert np.abs(got - PAML_PR_W_GT_1).max() < 5e-4


def test_oracle_loop_form_equals_vectorised_form(beb_case):
    """The literal restatement (running maximum, guards, loop order of beb.py:587-661) on a reduced problem."""
    from crabml_amd import beb
    from oracle import beb as obeb
    c = beb_case
    cols, props = beb.paml_grid_m2a()
    pick = np.arange(0, 10000, 97)
    lnf = c["site_log_liks"][:9].copy()
    lnf[3, :] = np.log(1e-300)          # an all-underflowed column: equal classes after scaling
    lnf[4, 11:] = np.log(1e-300)        # classes that drop below the 1e-300 guard after scaling
    a = obeb.site_posteriors_loops(lnf, cols[pick], props[pick])
    b = obeb.site_posteriors(lnf, cols[pick], props[pick])
    np.testing.assert_allclose(a[0], b[0], rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-13)
    assert abs(a[2] - b[2]) < 1e-11
    # weights = multiplicities: a table of unique patterns gives the same grid likelihoods as the site table
    dup = np.concatenate([lnf, lnf[:4]])
    wts = np.array([2, 2, 2, 2, 1, 1, 1, 1, 1], dtype=float)
    np.testing.assert_allclose(obeb.site_posteriors(dup, cols[pick], props[pick])[1],
                               obeb.site_posteriors(lnf, cols[pick], props[pick], weights=wts)[1], rtol=1e-13)


def test_oracle_reproduces_the_reference_adaptive_grid_accumulation():
    """compute_site_posteriors (beb.py:663-731) on nine M8 grid points: the restatement against the reference's output,
    from the very tables the reference was handed; and the class omegas the device path would be given (quantiles from
    crab_discretise_omega) against the ones the reference's model class produced with scipy."""
    from crabml_amd.engine import discretise_omega
    from oracle import beb as obeb
    c = load_case("lysozyme_m8_beb_grid")
    post = obeb.site_posteriors_weighted(c["tables"], c["grid_proportions"], c["weights"])
    assert np.abs(post - c["posteriors_ref"]).max() <= 1e-15
    np.testing.assert_allclose(post.sum(axis=1), c["weights"].sum(), atol=1e-13)
    for pt, om in zip(c["grid_points"], c["grid_omegas"]):
        got = np.append(discretise_omega("beta", pt[2], pt[3], 10), max(pt[4], 1.001))
        np.testing.assert_allclose(got, om, rtol=1e-12, atol=0)


def _asked_order(single, pair):
    """Row of hessian_points for each likelihood the reference asked for, in its order (beb.py:275-290)."""
    n = len(single)
    return [r for i in range(n) for j in range(i, n) for r in (0, single[i], single[j], pair[i, j])]


def test_hessian_stencil_and_adaptive_grid_match_the_reference():
    """compute_hessian / setup_integration_grid (beb.py:234-387) at the lysozyme M2a estimates.  The reference makes 60
    evaluations for 5 parameters; the mirror evaluates the 21 distinct points once.  Pinned: the class parameters at
    every stencil point (through the first rate matrix and the proportions the reference handed its calculator), the
    Hessian assembled from the reference's own likelihood values (same expression, so bit for bit), and the grid and
    weights derived from the reference's Hessian.  Not pinned, and not reproducible by anyone: the Hessian from
    independently computed likelihoods -- the h = 1e-5 stencil divides rounding noise of |lnL| ~ 900 by 1e-10."""
    from crabml_amd import beb, models
    c = load_case("lysozyme_m2a_adaptive_grid")
    names = [str(s) for s in c["param_names"]]
    mle = dict(zip(names, map(float, c["mle"])))
    pts, single, pair = beb.hessian_points(c["mle"])
    assert pts.shape == (21, 5)
    order = _asked_order(single, pair)
    assert len(order) == len(c["asked_lnl"]) == 60
    f = np.full(len(pts), np.nan)
    for r, q0, pr, lnl in zip(order, c["asked_Q0"], c["asked_proportions"], c["asked_lnl"]):
        assert beb.in_bounds(names, pts[r])
        model = models.M2aCodonModel(**dict(mle, **dict(zip(names, pts[r]))), pi=c["pi"])
        np.testing.assert_array_equal(model.get_site_classes()[0], pr)
        assert np.abs(model.get_Q_matrices()[0] - q0).max() <= 4e-15 * np.abs(q0).max()
        assert np.isnan(f[r]) or f[r] == -lnl           # the reference's repeated evaluations are deterministic
        f[r] = -lnl
    np.testing.assert_array_equal(beb.hessian_from_values(f, single, pair), c["hessian_ref"])
    grid, w = beb.adaptive_grid(c["hessian_ref"], names, c["mle"], 3)
    assert len(grid) == len(c["grid_points_ref"]) == 30
    np.testing.assert_allclose(np.asarray(grid), c["grid_points_ref"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(w, c["weights_ref"], rtol=1e-9)
    # out-of-bounds stencil points are +inf for the reference (beb.py:249-250) and poison only their own entries
    f2 = f.copy()
    f2[pair[3, 4]] = np.inf
    H2 = beb.hessian_from_values(f2, single, pair)
    assert np.isinf(H2[3, 4]) and np.isinf(H2[4, 3]) and np.isfinite(np.delete(H2.ravel(), [3 * 5 + 4, 4 * 5 + 3])).all()
    assert not beb.in_bounds(names, [2.0, 0.3, 2.0, 1.2, 0.1]) and not beb.in_bounds(names, [-1.0, 0.3, 2.0, 0.2, 0.1])
    assert not beb.in_bounds(["p_beta", "q_beta"], [0.0, 1.0]) and beb.in_bounds(["p_beta", "q_beta"], [3.0, 1.0])


def test_adaptive_grid_weights_are_the_normal_density():
    """The reference weights grid points with scipy's multivariate_normal pdf (beb.py:360-379); the mirror uses the closed
    form along the principal axes.  Same numbers, general offsets (n_grid_points = 4) and a singular Hessian included."""
    from scipy.stats import multivariate_normal
    from crabml_amd import beb
    rng = np.random.default_rng(3)
    A = rng.normal(size=(3, 3))
    H = (A @ A.T + 0.5 * np.eye(3)) * 1e3
    x = np.array([2.0, 0.4, 0.5])
    names = ["kappa", "omega0", "p0"]
    for n in (3, 4, 5):
        grid, w = beb.adaptive_grid(H, names, x, n)
        lam, vec = np.linalg.eigh(np.linalg.inv(H))
        cov = vec @ np.diag(np.maximum(lam, 1e-10)) @ vec.T
        pdf = np.array([multivariate_normal(mean=x, cov=cov).pdf(p) for p in grid])
        np.testing.assert_allclose(w, pdf / pdf.sum(), rtol=1e-9)
        assert len(grid) <= n ** 3 and abs(w.sum() - 1.0) < 1e-14
    Hs = np.outer(A[0], A[0]) * 1e3                       # rank one: inv() succeeds with huge entries or raises; either way finite
    grid, w = beb.adaptive_grid(Hs, names, x, 3)
    assert len(grid) >= 1 and np.isfinite(w).all() and abs(w.sum() - 1.0) < 1e-12
