"""Bayes empirical Bayes site posteriors on PAML's fixed grid (M2a), with the integration on the device.

Mirror of the reference's ``BEBCalculator.setup_paml_grid`` / ``compute_site_posteriors_paml``
(src/crabml/analysis/beb.py:389-661): the grid and the 21 rate matrices are built here exactly as the reference
builds them, everything after that (the [S, 21] site likelihood table, ln f(X | grid point) for the 10 000 grid
points, and the per-site class posteriors -- pure-Python loops over sites x grid points in the reference) is one
call through the C ABI, ``crab_eval_beb_grid``.
"""
from __future__ import annotations

import math

import numpy as np

from . import models

N1D = 10   # PAML's grid resolution (beb.py:453-454)


def get_index_ternary(itriangle: int, K: int):
    """PAML's GetIndexTernary (beb.py:389-426): triangle index -> (ix, iy, p0, p1), centroid coordinates."""
    ix = math.isqrt(itriangle)
    iy = itriangle - ix * ix
    x = (1 + (iy // 2) * 3 + (iy % 2)) / (3.0 * K)
    y = (1 + (K - 1 - ix) * 3 + (iy % 2)) / (3.0 * K)
    return ix, iy, x, y


def paml_omega_grid(n1d: int = N1D):
    """The 21 omegas of the M2a BEB grid: 10 bin midpoints in (0, 1), 1, 10 bin midpoints in (1, 11) (beb.py:562-565)."""
    w0 = [(i + 0.5) / n1d for i in range(n1d)]
    w2 = [1.0 + (i + 0.5) * 10.0 / n1d for i in range(n1d)]
    return w0 + [1.0] + w2


def paml_grid_m2a(n1d: int = N1D):
    """(cols [G, 3] int32, props [G, 3]) in the reference's order: triangles of the (p0, p1) simplex outermost, then
    omega0, then omega2 (beb.py:470-500); class 0 reads column i0, class 1 column n1d, class 2 column n1d + 1 + i2
    (:597-602)."""
    cols, props = [], []
    for tri in range(n1d * n1d):
        _, _, p0, p1 = get_index_ternary(tri, n1d)
        p2 = 1.0 - p0 - p1
        if not (0.0 <= p0 <= 1.0 and 0.0 <= p1 <= 1.0):   # BEBCalculator._check_bounds (:216-220); never fails by construction
            continue
        for i0 in range(n1d):
            for i2 in range(n1d):
                cols.append((i0, n1d, n1d + 1 + i2))
                props.append((p0, p1, p2))
    return np.asarray(cols, dtype=np.int32), np.asarray(props, dtype=np.float64)


def beb_rate_matrices(kappa, pi, mle_proportions, mle_omegas, n1d: int = N1D):
    """The 21 matrices of beb.py:545-573: unnormalised Q(kappa, omega) times Qfactor_NS = 1 / sum_k p_k mu_k at the MLE."""
    qfactor = 1.0 / sum(p * models.mean_rate(models.build_codon_Q_matrix(kappa, w, pi, 1.0), pi)
                        for p, w in zip(mle_proportions, mle_omegas))
    return np.stack([models.build_codon_Q_matrix(kappa, w, pi, 1.0) * qfactor for w in paml_omega_grid(n1d)])


def m2a_site_posteriors(engine, kappa, pi, mle_proportions, mle_omegas, branch_lengths, n1d: int = N1D):
    """[n_sites, 3] posterior class probabilities (omega0 < 1, omega = 1, omega2 > 1) and ln f(X)."""
    Q = beb_rate_matrices(kappa, pi, mle_proportions, mle_omegas, n1d)
    cols, props = paml_grid_m2a(n1d)
    post, _, ln_fx = engine.beb_grid(Q, pi, branch_lengths, cols, props)
    return post, ln_fx


def site_posteriors_over_grid(engine, class_parameters, weights, pi, branch_lengths, batch: int = 64):
    """The adaptive-grid path, ``BEBCalculator.compute_site_posteriors`` (beb.py:663-731): ``class_parameters`` is one
    (kappa, omegas[K], proportions[K]) triple per grid point -- what ``model_class(**params).get_site_classes()`` gives
    -- and ``weights`` the grid weights.  Grid points go to the device ``batch`` at a time; the accumulation over the
    grid happens there (kernel ``class_posteriors_kernel``), only [n_sites, K] comes back per batch."""
    total = None
    for lo in range(0, len(class_parameters), batch):
        chunk = class_parameters[lo:lo + batch]
        ka = np.array([c[0] for c in chunk], dtype=np.float64)
        om = np.array([c[1] for c in chunk], dtype=np.float64)
        pr = np.array([c[2] for c in chunk], dtype=np.float64)
        part = engine.codon_class_posteriors(ka, om, pr, np.asarray(weights[lo:lo + batch], dtype=np.float64), pi, branch_lengths)
        total = part if total is None else total + part
    return total


# ---- the adaptive grid: Hessian at the MLE, principal axes, weighted grid (beb.py:201-387, 733-796) ----------------
def in_bounds(names, x) -> bool:
    """BEBCalculator._check_bounds (beb.py:201-232) for the parameter vector x in the order of ``names``; parameters
    that are not integrated over are inside their bounds by assumption (they sit at the MLE)."""
    for name, v in zip(names, x):
        if name == "kappa" and v <= 0:
            return False
        if name.startswith("p") and not name.endswith("beta") and (v < 0 or v > 1):
            return False
        if name in ("p_beta", "q_beta") and v <= 0:
            return False
        if name.startswith("omega") and v < 0:
            return False
    return True


def hessian_points(x, h: float = 1e-5):
    """The distinct points of the reference's stencil (beb.py:275-294): x, x + h e_i, and (x + h e_i) + h e_j for
    i <= j -- 1 + n + n(n + 1)/2 of them where the reference makes 4 evaluations per (i, j), 2 n (n + 1) in all.
    Returns (points [m, n], single [n], pair [n, n]): row indices of x + h e_i and of x + h e_i + h e_j (j >= i)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    eye = np.eye(n) * h
    pts = [x.copy()]
    single = np.zeros(n, dtype=np.int64)
    pair = np.full((n, n), -1, dtype=np.int64)
    for i in range(n):
        single[i] = len(pts)
        pts.append(x + eye[i])
    for i in range(n):
        for j in range(i, n):
            pair[i, j] = len(pts)
            pts.append(x + eye[i] + eye[j])
    return np.asarray(pts), single, pair


def hessian_from_values(neg_lnl, single, pair, h: float = 1e-5):
    """beb.py:287-294 on the values at hessian_points: H_ij = (f_11 - f_10 - f_01 + f_00) / (h h), symmetric."""
    n = len(single)
    H = np.zeros((n, n))
    with np.errstate(invalid="ignore"):
        for i in range(n):
            for j in range(i, n):
                H[i, j] = H[j, i] = (neg_lnl[pair[i, j]] - neg_lnl[single[i]] - neg_lnl[single[j]] + neg_lnl[0]) / (h * h)
    return H


def _class_parameters(model_class, mle_params, names, x, pi):
    model = model_class(**dict(mle_params, **dict(zip(names, map(float, x)))), pi=pi)
    props, omegas = model.get_site_classes()
    return float(model.kappa), np.asarray(omegas, dtype=np.float64), np.asarray(props, dtype=np.float64)


def negative_log_likelihoods(engine, model_class, mle_params, names, points, pi, branch_lengths, no_clamp=False):
    """``neg_log_likelihood`` of beb.py:247-263 at many parameter vectors in ONE call through the C ABI
    (crab_eval_codon_site_classes with B = number of in-bounds points); out-of-bounds points are +inf as there."""
    out = np.full(len(points), np.inf)
    ok = [i for i, x in enumerate(points) if in_bounds(names, x)]
    if ok:
        cls = [_class_parameters(model_class, mle_params, names, points[i], pi) for i in ok]
        out[ok] = -engine.codon_site_classes(np.array([c[0] for c in cls]), np.array([c[1] for c in cls]),
                                             np.array([c[2] for c in cls]), pi,
                                             np.broadcast_to(np.asarray(branch_lengths, dtype=np.float64), (len(ok), len(branch_lengths))),
                                             no_clamp=no_clamp)
    return out


def hessian(engine, model_class, mle_params, names, pi, branch_lengths, h: float = 1e-5):
    """BEBCalculator.compute_hessian (beb.py:234-296): the whole stencil is one batched evaluation."""
    x = np.array([mle_params[k] for k in names], dtype=np.float64)
    pts, single, pair = hessian_points(x, h)
    f = negative_log_likelihoods(engine, model_class, mle_params, names, pts, pi, branch_lengths)
    return hessian_from_values(f, single, pair, h)


def adaptive_grid(H, names, x, n_grid_points: int = 5):
    """BEBCalculator.setup_integration_grid (beb.py:298-387): grid along the principal axes of the inverse Hessian,
    weighted by the normal density there.  The density is evaluated directly: a grid point is x + sum_i o_i s_i v_i with
    (s_i^2, v_i) the eigenpairs of the covariance, so its Mahalanobis distance is sum_i o_i^2 and the normalising
    constant drops out of the normalised weights (the reference asks scipy's multivariate_normal for the same number)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    try:
        cov = np.linalg.inv(H)
    except np.linalg.LinAlgError:
        lam, vec = np.linalg.eigh(H)
        cov = vec @ np.diag(1.0 / np.maximum(lam, 1e-6)) @ vec.T
    lam, vec = np.linalg.eigh(cov)
    lam = np.maximum(lam, 1e-10)
    sd = np.sqrt(lam)
    offsets = {5: [-2, -1, 0, 1, 2], 3: [-1, 0, 1]}.get(n_grid_points)
    if offsets is None:
        offsets = np.linspace(-2, 2, n_grid_points)
    offsets = np.asarray(offsets, dtype=np.float64)
    combos = np.stack(np.meshgrid(*([offsets] * n), indexing="ij"), axis=-1).reshape(-1, n)   # itertools.product order
    points, logw = [], []
    for o in combos:
        step = np.zeros(n)
        for i in range(n):                         # same accumulation order as beb.py:364-366
            step += o[i] * sd[i] * vec[:, i]
        pt = x + step
        if in_bounds(names, pt):
            points.append(pt)
            logw.append(-0.5 * float(o @ o))
    w = np.exp(np.asarray(logw))
    w = w / w.sum() if w.sum() > 0 else np.full(len(points), 1.0 / len(points))
    return points, w


def adaptive_site_posteriors(engine, model_class, mle_params, names, pi, branch_lengths, n_grid_points: int = 5, h: float = 1e-5,
                             batch: int = 64):
    """calculate_beb with grid_type='adaptive' (beb.py:746-796): Hessian, grid, accumulated class posteriors and the
    posterior mean / SE of omega per site.  Returns (posteriors [S, K], mean omega [S], SE [S], grid points, weights)."""
    x = np.array([mle_params[k] for k in names], dtype=np.float64)
    H = hessian(engine, model_class, mle_params, names, pi, branch_lengths, h)
    points, w = adaptive_grid(H, names, x, n_grid_points)
    cls = [_class_parameters(model_class, mle_params, names, p, pi) for p in points]
    post = site_posteriors_over_grid(engine, cls, w, pi, branch_lengths, batch)
    _, om, _ = _class_parameters(model_class, mle_params, names, x, pi)
    mean = post @ om
    se = np.sqrt(np.maximum(post @ om ** 2 - mean ** 2, 0.0))
    return post, mean, se, points, w


# ---- the reference's front door: BEBCalculator.calculate_beb (beb.py:133-799) ------------------------------------
from dataclasses import dataclass as _dataclass


@_dataclass
class BEBResult:
    """beb.py:21-76 without the report formatting."""
    site_numbers: np.ndarray
    posterior_probs: np.ndarray
    posterior_omega: np.ndarray
    posterior_omega_se: np.ndarray
    model_name: str
    site_classes: list
    class_omegas: np.ndarray
    class_proportions: np.ndarray

    def significant_sites(self, threshold: float = 0.95, class_idx: int = -1):
        return self.site_numbers[self.posterior_probs[:, class_idx] > threshold]


class BEBCalculator:
    """Same constructor and ``calculate_beb()`` as the reference's (beb.py:161-188, 733-799); ``model_class`` is one of
    crabml_amd.models' site-class models.  grid_type 'paml' (M2a only, as in the reference) integrates over PAML's 10 000
    point grid on the device; 'adaptive' evaluates the Hessian stencil and the grid in batched calls."""

    def __init__(self, mle_params, alignment, tree, model_class, param_names, fix_branch_lengths: bool = True, n_grid_points: int = 5,
                 grid_type: str = "adaptive", device: int = 0, engine=None):
        from .likelihood_rust import RustLikelihoodCalculator
        self.mle_params, self.alignment, self.tree, self.model_class = dict(mle_params), alignment, tree, model_class
        self.param_names, self.fix_branch_lengths = list(param_names), fix_branch_lengths
        self.n_grid_points, self.grid_type = n_grid_points, grid_type
        self.pi = models.compute_codon_frequencies_f3x4(alignment)
        self.calc = RustLikelihoodCalculator(alignment, tree, device=device, engine=engine)
        self.mle_model = model_class(**self.mle_params, pi=self.pi)

    def calculate_beb(self) -> BEBResult:
        bl = self.calc._get_current_branch_lengths()
        props, omegas = self.mle_model.get_site_classes()
        omegas = np.asarray(omegas, dtype=np.float64)
        if self.grid_type == "paml":
            post, _ = m2a_site_posteriors(self.calc.engine, float(self.mle_model.kappa), self.pi, props, omegas, bl)
        elif self.grid_type == "adaptive":
            post, *_ = adaptive_site_posteriors(self.calc.engine, self.model_class, self.mle_params, self.param_names, self.pi, bl,
                                                self.n_grid_points)
        else:
            raise ValueError(f"Unknown grid_type: {self.grid_type}. Must be 'adaptive' or 'paml'.")
        mean = post @ omegas
        se = np.sqrt(np.maximum(post @ omegas ** 2 - mean ** 2, 0))
        return BEBResult(site_numbers=np.arange(1, self.alignment.n_sites + 1), posterior_probs=post, posterior_omega=mean,
                         posterior_omega_se=se, model_name=self.model_class.__name__, site_classes=[f"ω={w:.2f}" for w in omegas],
                         class_omegas=omegas, class_proportions=np.asarray(props, dtype=np.float64))
