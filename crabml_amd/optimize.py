"""Maximum-likelihood fits with the whole finite-difference gradient in one call.

Mirror of the reference's optimisers for the site-class models (src/crabml/optimize/optimizer.py: M0Optimizer :24-249,
M1aOptimizer :252-408, M2aOptimizer :411-601, M3Optimizer :604-864, M7Optimizer :867-1029, M8Optimizer :1032-1218,
M8aOptimizer :1221-1412, M5/M9/M4/M6Optimizer :1415-2100;
src/crabml/optimize/branch_site.py: BranchSiteModelAOptimizer; src/crabml/optimize/branch.py: BranchModelOptimizer): same parameter transforms, bounds, starting values and return values,
same ``scipy.optimize.minimize(method='L-BFGS-B')`` driver.  The difference is where the gradient comes from.  The
reference passes no ``jac``, so scipy differentiates numerically: n + 1 sequential calls through the FFI per gradient
(optimizer.py:225-231).  Here the objective hands scipy ``(f, grad)`` itself: the point and its n forward-difference
neighbours -- the very points and steps scipy's ``approx_derivative(method='2-point', abs_step=1e-8, bounds=...)``
would visit, reproduced in :func:`forward_difference_points` and tested bit for bit against scipy -- are one batched
evaluation through the C ABI (``crab_eval_codon_site_classes`` with B = n + 1; the device builds each distinct
transition matrix once).  Given equal likelihood values the iterates are therefore the reference's.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import minimize

from . import models
from .likelihood_rust import RustLikelihoodCalculator

EPS = 1e-8   # scipy's L-BFGS-B default ``eps``, the absolute step of its numerical gradient


def _bound_arrays(bounds):
    lb = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
    ub = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
    return lb, ub


def forward_difference_points(x, bounds=None, eps: float = EPS):
    """(points [n, n], dx [n]): row i is x with x_i + h_i, where h follows scipy's rules for an absolute step with
    bounds -- +eps, flipped to -eps where x_i + eps leaves [lb_i, ub_i] and eps still fits on the other side, else the
    distance to the farther bound; dx_i = (x_i + h_i) - x_i, the exactly representable step scipy divides by."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    h = np.full(n, float(eps))
    lost = (x + h) - x == 0                                   # a step lost to rounding falls back to a relative one
    h[lost] = np.sqrt(np.finfo(np.float64).eps) * np.where(x[lost] >= 0, 1.0, -1.0) * np.maximum(1.0, np.abs(x[lost]))
    if bounds is not None:
        lb, ub = bounds if isinstance(bounds, tuple) and len(bounds) == 2 and isinstance(bounds[0], np.ndarray) else _bound_arrays(bounds)
        if not np.all((lb == -np.inf) & (ub == np.inf)):
            lower, upper = x - lb, ub - x
            moved = x + h
            violated = (moved < lb) | (moved > ub)
            fitting = np.abs(h) <= np.maximum(lower, upper)
            h[violated & fitting] *= -1
            fwd = (upper >= lower) & ~fitting
            h[fwd] = upper[fwd]
            back = (upper < lower) & ~fitting
            h[back] = -lower[back]
    pts = np.tile(x, (n, 1))
    pts[np.arange(n), np.arange(n)] += h
    return pts, pts[np.arange(n), np.arange(n)] - x


def value_and_gradient(batch_fun, x, bounds=None, eps: float = EPS):
    """f(x) and its forward-difference gradient from ONE call of ``batch_fun`` on the [n + 1, n] stack of points."""
    pts, dx = forward_difference_points(x, bounds, eps)
    f = np.asarray(batch_fun(np.vstack([np.asarray(x, dtype=np.float64)[None, :], pts])), dtype=np.float64)
    return float(f[0]), (f[1:] - f[0]) / dx


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


class _SiteClassOptimizer:
    """What the reference's optimisers share: frequencies, calculator, branch bookkeeping, the scipy driver."""

    n_model = 0                     # leading model parameters before the branch lengths / the branch scale
    model_bounds: list = []

    def __init__(self, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True, init_with_m0: bool = True,
                 device: int = 0, batched_gradient: bool = True, engine=None, rank: int = 0, world_size: int = 1, group=None,
                 no_clamp: bool = False):
        # no_clamp: evaluate without the reference's max(f, 1e-300) (CRAB_FLAG_NO_CLAMP).  The default keeps the
        # reference's arithmetic; on trees of several hundred taxa most site likelihoods lie below 1e-300, the clamped
        # objective is the constant n_sites * ln(1e-300) with a zero gradient and L-BFGS-B stops where it started --
        # fits at that size need no_clamp=True (the device's rescaled likelihood is exact there).
        self.no_clamp = no_clamp
        self.alignment, self.tree = alignment, tree
        self.init_with_m0, self.device, self.engine = init_with_m0, device, engine
        self.rank, self.world_size, self.group = rank, world_size, group
        self.use_f3x4, self.optimize_branch_lengths = use_f3x4, optimize_branch_lengths
        self.pi = models.compute_codon_frequencies_f3x4(alignment) if use_f3x4 else np.ones(61) / 61
        # engine: one to re-target, e.g. a scan worker's.  world_size > 1: this rank's engine holds its slice of the site
        # patterns and every likelihood below is the rank-ordered sum of the ranks' partial sums (one all-gather per
        # gradient), so all ranks see the same bits and walk the same path.
        self.calc = RustLikelihoodCalculator(alignment, tree, device=device, engine=engine,
                                             pattern_slice=self._pattern_slice(alignment, rank, world_size))
        self.branch_nodes = [v for v in tree.postorder() if tree.parent[v] >= 0]          # optimizer.py:72
        self.n_branches = len(self.branch_nodes)
        self._branch_rows = np.array([self.calc.node_id_map[v] for v in self.branch_nodes])
        self.batched_gradient = batched_gradient
        self.history = []
        self.n_calls = 0            # calls through the C ABI
        self.n_vectors = 0          # parameter vectors evaluated

    @staticmethod
    def _pattern_slice(alignment, rank, world_size):
        if world_size == 1:
            return None
        from .dist import pattern_slices
        from .engine import compress_patterns
        n_patterns = len(compress_patterns(np.ascontiguousarray(alignment.sequences, dtype=np.int32))[1])
        return pattern_slices(n_patterns, world_size)[rank]

    # -- model specifics, per subclass ---------------------------------------------------------
    def model_at(self, p):          # p: the leading n_model entries of a parameter vector
        raise NotImplementedError

    def record(self, model, lnl):
        raise NotImplementedError

    # -- shared ------------------------------------------------------------------------------
    def _branch_lengths(self, params):
        """[B, n_nodes] in the boundary's node order (the root's entry stays 0)."""
        B = params.shape[0]
        if self.optimize_branch_lengths:
            bl = np.zeros((B, len(self.calc.tree_structure)))
            bl[:, self._branch_rows] = np.exp(params[:, self.n_model:self.n_model + self.n_branches])
            return bl
        return self.calc._get_current_branch_lengths()[None, :] * np.exp(params[:, self.n_model])[:, None]

    def negative_log_likelihoods(self, params):
        """The reference's ``compute_log_likelihood`` (e.g. optimizer.py:78-136) for a stack of parameter vectors."""
        params = np.atleast_2d(np.asarray(params, dtype=np.float64))
        # the rows of a stencil mostly share their model parameters (all the branch-length rows do): one model each
        head = params[:, :self.n_model]
        same = np.all(head == head[0], axis=1)
        if same.all():
            lead, row_model = head[:1], np.zeros(len(params), dtype=np.intp)
        elif same.sum() + self.n_model >= len(params):      # a stencil: row 0's model, plus one row per model parameter
            others = np.flatnonzero(~same)
            lead = np.vstack([head[:1], head[others]])
            row_model = np.zeros(len(params), dtype=np.intp)
            row_model[others] = 1 + np.arange(len(others))
        else:
            lead, row_model = np.unique(head, axis=0, return_inverse=True)
            row_model = row_model.reshape(-1)
        distinct = [self.model_at(p) for p in lead]
        for m in distinct:
            m.c_quantiles = True        # beta quantiles from the C library (models._beta_quantiles_c)
        cls = [m.get_site_classes() for m in distinct]
        lnl = self.calc.engine.codon_site_classes(np.array([m.kappa for m in distinct], dtype=np.float64)[row_model],
                                                  np.array([c[1] for c in cls], dtype=np.float64)[row_model],
                                                  np.array([c[0] for c in cls], dtype=np.float64)[row_model], self.pi,
                                                  self._branch_lengths(params), no_clamp=self.no_clamp)
        if self.world_size > 1:
            from .dist import ordered_sum_vector
            lnl = ordered_sum_vector(lnl, self.group)
        self.n_calls += 1
        self.n_vectors += len(params)
        if self.optimize_branch_lengths:                                   # the reference leaves the last values in the tree
            self.tree.branch_length[np.array(self.branch_nodes)] = np.exp(params[0, self.n_model:self.n_model + self.n_branches])
        self.history.append(self.record(distinct[row_model[0]], float(lnl[0])))
        return -lnl

    def compute_log_likelihood(self, params):
        return float(self.negative_log_likelihoods(np.asarray(params, dtype=np.float64)[None, :])[0])

    def _initial_branch_parameters(self):
        if not self.optimize_branch_lengths:
            return [np.log(1.0)], [(np.log(0.01), np.log(100))]
        init = [float(self.tree.branch_length[v]) for v in self.branch_nodes]
        if max(init) < 0.001 and isinstance(self, M0Optimizer):          # optimizer.py:170-188: only M0 has the fallback
            from .distance_init import initialize_branch_lengths_ls
            init = [float(b) for b in initialize_branch_lengths_ls(self.alignment, self.tree, self.branch_nodes)]
            self.tree.branch_length[np.array(self.branch_nodes)] = init
        return [np.log(max(b, 0.001)) for b in init], [(np.log(0.0001), np.log(50))] * self.n_branches

    def _minimize(self, init_model, method, maxiter):
        if self.init_with_m0 and self.optimize_branch_lengths and not isinstance(self, M0Optimizer):   # optimizer.py:346-355
            m0 = M0Optimizer(self.alignment, self.tree, use_f3x4=self.use_f3x4, optimize_branch_lengths=True, device=self.device,
                             batched_gradient=self.batched_gradient, engine=self.engine, rank=self.rank, world_size=self.world_size,
                             group=self.group, no_clamp=self.no_clamp)
            m0.optimize()
            self.n_calls += m0.n_calls
            self.n_vectors += m0.n_vectors
        b0, bb = self._initial_branch_parameters()
        x0 = np.array(list(init_model) + b0, dtype=np.float64)
        bounds = list(self.model_bounds) + bb
        self.history = []
        if self.batched_gradient and method == "L-BFGS-B":
            bound_arrays = _bound_arrays(bounds)
            return minimize(lambda x: value_and_gradient(self.negative_log_likelihoods, x, bound_arrays), x0, jac=True,
                            method=method, bounds=bounds, options={"maxiter": maxiter})
        return minimize(self.compute_log_likelihood, x0, method=method, bounds=bounds, options={"maxiter": maxiter})


class M0Optimizer(_SiteClassOptimizer):
    """optimizer.py:24-249.  M0 is evaluated as a one-class mixture: same normalisation, same clamp, ln 1 = 0 added."""
    n_model = 2
    model_bounds = [(np.log(0.1), np.log(100)), (np.log(0.001), np.log(10))]

    class _OneClass(models._SiteClassModel):
        def __init__(self, kappa, omega, pi):
            self.kappa, self.omega, self.pi = kappa, omega, pi

        def get_site_classes(self):
            return [1.0], [self.omega]

    def model_at(self, p):
        return self._OneClass(float(np.exp(p[0])), float(np.exp(p[1])), self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "omega": m.omega, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_omega: float = 0.4, method: str = "L-BFGS-B", maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_omega)], method, maxiter)
        return float(np.exp(r.x[0])), float(np.exp(r.x[1])), float(-r.fun)


class M1aOptimizer(_SiteClassOptimizer):
    """optimizer.py:252-408."""
    n_model = 3
    model_bounds = [(np.log(0.1), np.log(100)), (np.log(0.001), np.log(0.999)), (-10, 10)]

    def model_at(self, p):
        return models.M1aCodonModel(kappa=float(np.exp(p[0])), omega0=float(np.exp(p[1])), p0=float(_sigmoid(p[2])), pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "omega0": m.omega0, "p0": m.p0, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_omega0: float = 0.5, init_p0: float = 0.7, method: str = "L-BFGS-B",
                 maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_omega0), np.log(init_p0 / (1 - init_p0))], method, maxiter)
        return float(np.exp(r.x[0])), float(np.exp(r.x[1])), float(_sigmoid(r.x[2])), float(-r.fun)


class M2aOptimizer(_SiteClassOptimizer):
    """optimizer.py:411-601: p0 and the neutral fraction f of the rest are the free proportions (p1 = (1 - p0) f)."""
    n_model = 5
    model_bounds = [(np.log(0.1), np.log(100)), (np.log(0.001), np.log(0.999)), (np.log(1.001), np.log(20)), (-10, 10), (-10, 10)]

    def model_at(self, p):
        p0 = float(_sigmoid(p[3]))
        return models.M2aCodonModel(kappa=float(np.exp(p[0])), omega0=float(np.exp(p[1])), omega2=float(np.exp(p[2])),
                                    p0=p0, p1=(1.0 - p0) * float(_sigmoid(p[4])), pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "omega0": m.omega0, "omega2": m.omega2, "p0": m.p0, "p1": m.p1, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_omega0: float = 0.5, init_omega2: float = 2.0, init_p0: float = 0.5,
                 init_p1: float = 0.3, method: str = "L-BFGS-B", maxiter: int = 200):
        f = init_p1 / (1.0 - init_p0) if init_p0 < 1.0 else 0.5
        f = max(0.01, min(0.99, f))
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_omega0), np.log(init_omega2),
                                          np.log(init_p0 / (1 - init_p0)), np.log(f / (1 - f))], method, maxiter)
        p0 = float(_sigmoid(r.x[3]))
        return (float(np.exp(r.x[0])), float(np.exp(r.x[1])), float(np.exp(r.x[2])), p0, (1.0 - p0) * float(_sigmoid(r.x[4])),
                float(-r.fun))


_BETA_BOUND = (np.log(0.005), np.log(99))


class M7Optimizer(_SiteClassOptimizer):
    """optimizer.py:867-1029: kappa and the two beta shape parameters; ``ncatG`` equal-weight median quantiles."""
    n_model = 3
    model_bounds = [(np.log(0.1), np.log(100)), _BETA_BOUND, _BETA_BOUND]

    def __init__(self, alignment, tree, ncatG: int = 10, **kw):
        super().__init__(alignment, tree, **kw)
        self.ncatG = ncatG

    def model_at(self, p):
        return models.M7CodonModel(kappa=float(np.exp(p[0])), p_beta=float(np.exp(p[1])), q_beta=float(np.exp(p[2])),
                                   ncatG=self.ncatG, pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "p_beta": m.p_beta, "q_beta": m.q_beta, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_p_beta: float = 0.5, init_q_beta: float = 0.5, method: str = "L-BFGS-B",
                 maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_p_beta), np.log(init_q_beta)], method, maxiter)
        return float(np.exp(r.x[0])), float(np.exp(r.x[1])), float(np.exp(r.x[2])), float(-r.fun)


class M8Optimizer(M7Optimizer):
    """optimizer.py:1032-1218: M7's beta classes with weight p0 plus one class omega_s > 1."""
    n_model = 5
    model_bounds = [(np.log(0.1), np.log(100)), (-10, 10), _BETA_BOUND, _BETA_BOUND, (np.log(1.001), np.log(20))]

    def model_at(self, p):
        return models.M8CodonModel(kappa=float(np.exp(p[0])), p0=float(_sigmoid(p[1])), p_beta=float(np.exp(p[2])),
                                   q_beta=float(np.exp(p[3])), omega_s=float(np.exp(p[4])), ncatG=self.ncatG, pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "p0": m.p0, "p_beta": m.p_beta, "q_beta": m.q_beta, "omega_s": m.omega_s, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_p0: float = 0.9, init_p_beta: float = 0.7, init_q_beta: float = 1.5,
                 init_omega_s: float = 2.5, method: str = "L-BFGS-B", maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_p0 / (1 - init_p0)), np.log(init_p_beta),
                                          np.log(init_q_beta), np.log(init_omega_s)], method, maxiter)
        return (float(np.exp(r.x[0])), float(_sigmoid(r.x[1])), float(np.exp(r.x[2])), float(np.exp(r.x[3])), float(np.exp(r.x[4])),
                float(-r.fun))


class M8aOptimizer(M7Optimizer):
    """optimizer.py:1221-1412: M8 with omega_s fixed at 1, the null of the M8a-M8 test."""
    n_model = 4
    model_bounds = [(np.log(0.1), np.log(100)), (-10, 10), _BETA_BOUND, _BETA_BOUND]

    def model_at(self, p):
        return models.M8aCodonModel(kappa=float(np.exp(p[0])), p0=float(_sigmoid(p[1])), p_beta=float(np.exp(p[2])),
                                    q_beta=float(np.exp(p[3])), ncatG=self.ncatG, pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "p0": m.p0, "p_beta": m.p_beta, "q_beta": m.q_beta, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_p0: float = 0.9, init_p_beta: float = 0.7, init_q_beta: float = 1.5,
                 method: str = "L-BFGS-B", maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_p0 / (1 - init_p0)), np.log(init_p_beta),
                                          np.log(init_q_beta)], method, maxiter)
        return float(np.exp(r.x[0])), float(_sigmoid(r.x[1])), float(np.exp(r.x[2])), float(np.exp(r.x[3])), float(-r.fun)


class BranchSiteModelAOptimizer:
    """optimize/branch_site.py:17-398 -- Model A (``fix_omega=False``) and its null (omega2 = 1).

    One behaviour of the reference is kept on purpose, because results depend on it: its objective hands the FFI
    ``calc.branch_lengths_template``, the lengths the tree had when the calculator was built (branch_site.py:180,
    likelihood_rust.py:124-132), not the lengths it has just written into the tree from the parameter vector (:143-150).
    The branch-length parameters therefore never reach the likelihood: their gradient components are zero, L-BFGS-B
    leaves them where they started, and the fit is over (kappa, omega0[, omega2], p0, p1) on the input tree's lengths --
    yet every numerical gradient still pays one evaluation per branch.  Here the rows of a stencil that evaluate to the
    same thing are evaluated once (rows are compared on what reaches the likelihood), so a gradient is 5 or 6 distinct
    vectors in one call whatever the size of the tree.  ``live_branch_lengths=True`` instead lets the branch-length
    parameters through (what the docstring of the reference promises); every row is then distinct."""

    def __init__(self, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True, fix_omega: bool = False,
                 device: int = 0, batched_gradient: bool = True, engine=None, live_branch_lengths: bool = False,
                 rank: int = 0, world_size: int = 1, group=None):
        self.alignment, self.tree = alignment, tree
        self.rank, self.world_size, self.group = rank, world_size, group
        self.use_f3x4, self.optimize_branch_lengths, self.fix_omega = use_f3x4, optimize_branch_lengths, fix_omega
        self.batched_gradient, self.live_branch_lengths = batched_gradient, live_branch_lengths
        self.pi = models.compute_codon_frequencies_f3x4(alignment) if use_f3x4 else np.ones(61) / 61
        self.calc = RustLikelihoodCalculator(alignment, tree, device=device, engine=engine,      # world_size > 1: as the site-class fits
                                             pattern_slice=_SiteClassOptimizer._pattern_slice(alignment, rank, world_size))
        self.branch_nodes = [v for v in tree.postorder() if tree.parent[v] >= 0]
        self.n_branches = len(self.branch_nodes)
        self._branch_rows = np.array([self.calc.node_id_map[v] for v in self.branch_nodes])
        self.branch_labels = tree.branch_labels()
        self.branch_labels_u8 = self.calc.renumber_branch_labels()                       # branch_site.py:79-90
        self.model = models.BranchSiteModelA(self.pi, self.branch_labels, fix_omega=fix_omega)   # validates the labels
        self.n_model = 4 if fix_omega else 5
        self._template = np.asarray(self.calc.branch_lengths_template, dtype=np.float64)
        self.history = []
        self.n_calls = self.n_vectors = self.n_rows = 0

    def _unpack(self, p):
        """(kappa, omega0, omega2, p0, p1) of one parameter vector, branch_site.py:116-139."""
        kappa, omega0 = float(np.exp(p[0])), float(np.exp(p[1]))
        omega2, at = (1.0, 2) if self.fix_omega else (float(np.exp(p[2])), 3)
        p0, p1 = float(_sigmoid(p[at])), float(_sigmoid(p[at + 1]))
        total = p0 + p1
        if total >= 0.999:
            p0, p1 = p0 / total * 0.999, p1 / total * 0.999
        return kappa, omega0, omega2, p0, p1

    def _branch_lengths(self, params):
        B = params.shape[0]
        if not self.live_branch_lengths:
            return np.broadcast_to(self._template, (B, len(self._template)))
        if self.optimize_branch_lengths:
            bl = np.zeros((B, len(self._template)))
            bl[:, self._branch_rows] = np.exp(params[:, self.n_model:self.n_model + self.n_branches])
            return bl
        return self._template[None, :] * np.exp(params[:, self.n_model])[:, None]

    def negative_log_likelihoods(self, params):
        params = np.atleast_2d(np.asarray(params, dtype=np.float64))
        reaching = params if self.live_branch_lengths else params[:, :self.n_model]
        distinct, first, row_of = np.unique(reaching, axis=0, return_index=True, return_inverse=True)
        row_of = row_of.reshape(-1)
        rows = params[first]
        q = np.array([self._unpack(p) for p in rows])
        lnl = self.calc.engine.codon_branch_site(q[:, 0], q[:, 1], q[:, 2], q[:, 3], q[:, 4], self.pi,
                                                 np.ascontiguousarray(self._branch_lengths(rows)), self.branch_labels_u8)
        if self.world_size > 1:
            from .dist import ordered_sum_vector
            lnl = ordered_sum_vector(lnl, self.group)
        self.n_calls += 1
        self.n_vectors += len(rows)
        self.n_rows += len(params)
        if self.optimize_branch_lengths:                                   # branch_site.py:143-145: written, not read back
            self.tree.branch_length[np.array(self.branch_nodes)] = np.exp(params[0, self.n_model:self.n_model + self.n_branches])
        lnl = np.asarray(lnl, dtype=np.float64)[row_of]
        k, w0, w2, p0, p1 = self._unpack(params[0])
        self.history.append({"kappa": k, "omega0": w0, "omega2": w2, "p0": p0, "p1": p1, "log_likelihood": float(lnl[0])})
        return -lnl

    def compute_log_likelihood(self, params):
        return float(self.negative_log_likelihoods(np.asarray(params, dtype=np.float64)[None, :])[0])

    def optimize(self, init_kappa: float = 2.0, init_omega0: float = 0.1, init_omega2: float = 2.0, init_p0: float = 0.3,
                 init_p1: float = 0.3, method: str = "L-BFGS-B", maxiter: int = 300):
        lead = [np.log(init_kappa), np.log(init_omega0)] + ([] if self.fix_omega else [np.log(init_omega2)]) + \
               [np.log(init_p0 / (1 - init_p0)), np.log(init_p1 / (1 - init_p1))]
        bounds = [(np.log(0.1), np.log(100)), (np.log(0.001), np.log(0.999))] + \
                 ([] if self.fix_omega else [(np.log(1.001), np.log(20))]) + [(-10, 10), (-10, 10)]
        if self.optimize_branch_lengths:
            lead += [np.log(max(float(self.tree.branch_length[v]), 0.001)) for v in self.branch_nodes]
            bounds += [(np.log(0.0001), np.log(50))] * self.n_branches
        else:
            lead += [np.log(1.0)]
            bounds += [(np.log(0.01), np.log(100))]
        x0 = np.array(lead, dtype=np.float64)
        self.history = []
        if self.batched_gradient and method == "L-BFGS-B":
            r = minimize(lambda x: value_and_gradient(self.negative_log_likelihoods, x, bounds), x0, jac=True, method=method,
                         bounds=bounds, options={"maxiter": maxiter, "disp": False})
        else:
            r = minimize(self.compute_log_likelihood, x0, method=method, bounds=bounds, options={"maxiter": maxiter, "disp": False})
        self.result = r
        kappa, omega0, omega2, p0, p1 = self._unpack(r.x)
        return kappa, omega0, omega2, p0, p1, float(-r.fun)


class BranchModelOptimizer:
    """optimize/branch.py -- two-ratio / multi-ratio (one omega per branch label) and free-ratio (one per branch) models.

    Parameters: log kappa, log omega_j, log branch lengths.  As in the reference, the i-th branch-length parameter is
    written to entry i of the boundary's branch-length vector (``branch_lengths[:-1] = optimized_lengths``, branch.py
    ``compute_log_likelihood``) while its starting value is the length of the i-th non-root node of ``tree.postorder()``;
    the two orders differ for most trees, which permutes the starting lengths but not the optimum.  The fitted lengths
    are written back to the tree in the boundary's order, i.e. onto the nodes they were evaluated on."""

    def __init__(self, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True, free_ratio: bool = False,
                 device: int = 0, batched_gradient: bool = True, engine=None, rank: int = 0, world_size: int = 1, group=None,
                 no_clamp: bool = False):
        if not optimize_branch_lengths:
            raise NotImplementedError("Global branch scaling not yet implemented")     # as the reference
        self.no_clamp = no_clamp                                                       # see _SiteClassOptimizer: needed on large trees
        self.rank, self.world_size, self.group = rank, world_size, group
        self.alignment, self.tree, self.use_f3x4, self.free_ratio = alignment, tree, use_f3x4, free_ratio
        self.optimize_branch_lengths, self.batched_gradient = optimize_branch_lengths, batched_gradient
        self.pi = models.compute_codon_frequencies_f3x4(alignment) if use_f3x4 else np.ones(61) / 61
        self.calc = RustLikelihoodCalculator(alignment, tree, device=device, engine=engine,
                                             pattern_slice=_SiteClassOptimizer._pattern_slice(alignment, rank, world_size))
        order = self.calc.postorder_nodes                                              # leaves first, then internals; root last
        labels = np.array([int(tree.label[v]) if tree.parent[v] >= 0 else 0 for v in order])
        self.branch_nodes = [v for v in tree.postorder() if tree.parent[v] >= 0]
        self.n_branches = len(self.branch_nodes)
        self.model = models.CodonBranchModel(self.pi, labels, free_ratio=free_ratio, n_nodes=len(order))
        self._template = np.asarray(self.calc.branch_lengths_template, dtype=np.float64)
        self.history = []
        self.n_calls = self.n_vectors = 0

    def negative_log_likelihoods(self, params):
        params = np.atleast_2d(np.asarray(params, dtype=np.float64))
        n_w = self.model.n_omega
        if params.shape[1] != 1 + n_w + self.n_branches:
            raise ValueError(f"Expected {self.n_branches} branch lengths, got {params.shape[1] - 1 - n_w}")
        kappa = np.exp(params[:, 0])
        omegas = np.exp(params[:, 1:1 + n_w])
        bl = np.tile(self._template, (len(params), 1))
        bl[:, :-1] = np.exp(params[:, 1 + n_w:])
        lnl = np.asarray(self.calc.engine.codon_branch(kappa, np.ascontiguousarray(omegas[:, self.model.omega_labels]), self.pi, bl,
                                                       no_clamp=self.no_clamp))
        if self.world_size > 1:
            from .dist import ordered_sum_vector
            lnl = ordered_sum_vector(lnl, self.group)
        self.n_calls += 1
        self.n_vectors += len(params)
        self.history.append({"kappa": float(kappa[0]), "omega": {f"omega{i}": float(omegas[0, i]) for i in range(n_w)},
                             "log_likelihood": float(lnl[0])})
        return -lnl

    def compute_log_likelihood(self, params):
        return float(self.negative_log_likelihoods(np.asarray(params, dtype=np.float64)[None, :])[0])

    def optimize(self, init_kappa: float = 3.0, init_omega: float = 0.4, method: str = "L-BFGS-B", maxiter: int = 500):
        n_w = self.model.n_omega
        x0 = np.array([np.log(init_kappa)] + [np.log(init_omega)] * n_w +
                      [np.log(max(float(self.tree.branch_length[v]), 0.001)) for v in self.branch_nodes], dtype=np.float64)
        bounds = [(np.log(0.1), np.log(999))] + [(np.log(1e-6), np.log(999))] * n_w + [(np.log(0.0001), np.log(50))] * self.n_branches
        self.history = []
        if self.batched_gradient and method == "L-BFGS-B":
            r = minimize(lambda x: value_and_gradient(self.negative_log_likelihoods, x, bounds), x0, jac=True, method=method,
                         bounds=bounds, options={"maxiter": maxiter, "disp": False})
        else:
            r = minimize(self.compute_log_likelihood, x0, method=method, bounds=bounds, options={"maxiter": maxiter, "disp": False})
        self.result = r
        omega = np.exp(r.x[1:1 + n_w])
        self.tree.branch_length[np.array(self.calc.postorder_nodes[:-1])] = np.exp(r.x[1 + n_w:])
        return float(np.exp(r.x[0])), {f"omega{i}": float(omega[i]) for i in range(n_w)}, float(-r.fun)


class _NoM0Start(_SiteClassOptimizer):
    """M4, M5, M6 and M9 of the reference take no ``init_with_m0`` (optimizer.py:1415-2100): they start from the tree as given."""

    def __init__(self, alignment, tree, ncatG: int = 10, use_f3x4: bool = True, optimize_branch_lengths: bool = True, **kw):
        kw.pop("init_with_m0", None)
        super().__init__(alignment, tree, use_f3x4=use_f3x4, optimize_branch_lengths=optimize_branch_lengths, init_with_m0=False, **kw)
        self.ncatG = ncatG


class M5Optimizer(_NoM0Start):
    """optimizer.py:1415-1557: omega ~ Gamma(alpha, beta), ``ncatG`` median quantiles."""
    n_model = 3
    model_bounds = [(np.log(0.1), np.log(100)), _BETA_BOUND, _BETA_BOUND]

    def model_at(self, p):
        return models.M5CodonModel(kappa=float(np.exp(p[0])), alpha=float(np.exp(p[1])), beta=float(np.exp(p[2])), ncatG=self.ncatG, pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "alpha": m.alpha, "beta": m.beta, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_alpha: float = 1.0, init_beta: float = 1.0, method: str = "L-BFGS-B", maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_alpha), np.log(init_beta)], method, maxiter)
        return float(np.exp(r.x[0])), float(np.exp(r.x[1])), float(np.exp(r.x[2])), float(-r.fun)


class M9Optimizer(_NoM0Start):
    """optimizer.py:1560-1732: omega ~ p0 Beta(p, q) + (1 - p0) Gamma(alpha, beta)."""
    n_model = 6
    model_bounds = [(np.log(0.1), np.log(100)), (-10, 10), _BETA_BOUND, _BETA_BOUND, _BETA_BOUND, _BETA_BOUND]

    def model_at(self, p):
        return models.M9CodonModel(kappa=float(np.exp(p[0])), p0=float(_sigmoid(p[1])), p_beta=float(np.exp(p[2])), q_beta=float(np.exp(p[3])),
                                   alpha=float(np.exp(p[4])), beta_gamma=float(np.exp(p[5])), ncatG=self.ncatG, pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "p0": m.p0, "p_beta": m.p_beta, "q_beta": m.q_beta, "alpha": m.alpha, "beta_gamma": m.beta_gamma,
                "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_p0: float = 0.5, init_p_beta: float = 0.5, init_q_beta: float = 0.5,
                 init_alpha: float = 1.0, init_beta_gamma: float = 1.0, method: str = "L-BFGS-B", maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_p0 / (1 - init_p0)), np.log(init_p_beta), np.log(init_q_beta),
                                          np.log(init_alpha), np.log(init_beta_gamma)], method, maxiter)
        return (float(np.exp(r.x[0])), float(_sigmoid(r.x[1])), float(np.exp(r.x[2])), float(np.exp(r.x[3])), float(np.exp(r.x[4])),
                float(np.exp(r.x[5])), float(-r.fun))


class M6Optimizer(_NoM0Start):
    """optimizer.py:1899-2100: omega ~ p0 Gamma(alpha1, beta1) + (1 - p0) Gamma(alpha2, alpha2)."""
    n_model = 5
    model_bounds = [(np.log(0.1), np.log(100)), (-10, 10), _BETA_BOUND, _BETA_BOUND, _BETA_BOUND]

    def model_at(self, p):
        return models.M6CodonModel(kappa=float(np.exp(p[0])), p0=float(_sigmoid(p[1])), alpha1=float(np.exp(p[2])), beta1=float(np.exp(p[3])),
                                   alpha2=float(np.exp(p[4])), ncatG=self.ncatG, pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "p0": m.p0, "alpha1": m.alpha1, "beta1": m.beta1, "alpha2": m.alpha2, "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_p0: float = 0.5, init_alpha1: float = 1.0, init_beta1: float = 1.0,
                 init_alpha2: float = 1.0, method: str = "L-BFGS-B", maxiter: int = 200):
        self.result = r = self._minimize([np.log(init_kappa), np.log(init_p0 / (1 - init_p0)), np.log(init_alpha1), np.log(init_beta1),
                                          np.log(init_alpha2)], method, maxiter)
        return (float(np.exp(r.x[0])), float(_sigmoid(r.x[1])), float(np.exp(r.x[2])), float(np.exp(r.x[3])), float(np.exp(r.x[4])),
                float(-r.fun))


class M4Optimizer(_NoM0Start):
    """optimizer.py:1735-1896: five fixed omegas (0, 1/3, 2/3, 1, 3), free proportions as a softmax over four logits and 1."""
    n_model = 5
    model_bounds = [(np.log(0.1), np.log(100))] + [(-10, 10)] * 4

    def __init__(self, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True, **kw):
        super().__init__(alignment, tree, use_f3x4=use_f3x4, optimize_branch_lengths=optimize_branch_lengths, **kw)

    @staticmethod
    def _proportions(logits):
        raw = np.concatenate([np.exp(logits - np.max(logits)), [1.0]])                 # optimizer.py:1782-1785
        return raw / raw.sum()

    def model_at(self, p):
        return models.M4CodonModel(kappa=float(np.exp(p[0])), proportions=self._proportions(p[1:5]).tolist(), pi=self.pi)

    def record(self, m, lnl):
        return {"kappa": m.kappa, "proportions": list(m.get_site_classes()[0]), "log_likelihood": lnl}

    def optimize(self, init_kappa: float = 2.0, init_proportions=None, method: str = "L-BFGS-B", maxiter: int = 200):
        if init_proportions is None:
            init_proportions = [0.2, 0.2, 0.2, 0.2, 0.2]
        else:
            if len(init_proportions) != 5:
                raise ValueError("M4 requires exactly 5 initial proportions")
            init_proportions = np.array(init_proportions, dtype=np.float64)
            init_proportions = init_proportions / init_proportions.sum()
        self.result = r = self._minimize([np.log(init_kappa)] + np.log(init_proportions[:4]).tolist(), method, maxiter)
        return float(np.exp(r.x[0])), self._proportions(r.x[1:5]).tolist(), float(-r.fun)


class M3Optimizer(_SiteClassOptimizer):
    """optimizer.py:604-864: K free omegas and proportions in PAML's layout -- kappa and the omegas on their natural scale,
    proportions as f_k = e^{x_k} / (1 + sum e^{x_j}) with the last class as the reference, starting values from PAML's
    GetInitialsCodon formulas with ``numpy.random.default_rng(seed=trial)`` in place of rndu(), tight stopping rule."""

    def __init__(self, alignment, tree, n_classes: int = 3, **kw):
        super().__init__(alignment, tree, **kw)
        self.n_classes = n_classes
        self.n_model = 1 + (n_classes - 1) + n_classes

    @staticmethod
    def _f_and_x(logits):
        e = np.exp(np.asarray(logits, dtype=float))
        denom = 1.0 + np.sum(e)
        return np.append(e / denom, 1.0 / denom)

    def model_at(self, p):
        K = self.n_classes
        return models.M3CodonModel(kappa=float(p[0]), omegas=[float(w) for w in p[K:2 * K]], proportions=self._f_and_x(p[1:K]).tolist(),
                                   pi=self.pi)

    def record(self, m, lnl):
        props, omegas = m.get_site_classes()
        return {"kappa": m.kappa, "omegas": list(omegas), "proportions": list(props), "log_likelihood": lnl}

    def _build_bounds(self, K):
        return [(1e-4, 999.0)] + [(-99.0, 99.0)] * (K - 1) + [(1e-6, 999.0)] * K + \
            ([(np.log(0.0001), np.log(50))] * self.n_branches if self.optimize_branch_lengths else [(np.log(0.01), np.log(100))])

    def _build_init_params(self, K, kappa, omega, rng=None):
        rndu = rng.random if rng is not None else (lambda: 0.5)
        kappa_init = 0.1 + kappa * (0.8 + 0.4 * rndu())
        logits = [rndu() for _ in range(K - 1)]
        omegas = [max(omega * (0.5 + i * 2.0 / K * (0.8 + 0.4 * rndu())), 1e-4) for i in range(K)]
        tail = ([np.log(max(float(self.tree.branch_length[v]), 0.001)) for v in self.branch_nodes] if self.optimize_branch_lengths
                else [np.log(1.0)])
        return np.array([kappa_init] + logits + omegas + tail), kappa_init, omegas, logits

    @staticmethod
    def _enforce_bounds(params, bounds):
        for i, (lo, hi) in enumerate(bounds):
            if params[i] < lo * 1.005:
                params[i] = lo * 1.05
            if params[i] > hi / 1.005:
                params[i] = hi / 1.05

    def _extract_result(self, x, K):
        props, omegas = self._f_and_x(x[1:K]).tolist(), x[K:2 * K].tolist()
        order = np.argsort(omegas)                                                     # PAML's sortwM3
        return float(x[0]), [omegas[i] for i in order], [props[i] for i in order]

    def optimize(self, init_kappa: float = 2.0, init_omegas=None, init_proportions=None, method: str = "L-BFGS-B", maxiter: int = 500,
                 n_restarts: int = 1):
        m0_omega = 2.1
        if self.init_with_m0 and self.optimize_branch_lengths:
            m0 = M0Optimizer(self.alignment, self.tree, use_f3x4=self.use_f3x4, optimize_branch_lengths=True, device=self.device,
                             batched_gradient=self.batched_gradient, engine=self.engine, rank=self.rank, world_size=self.world_size,
                             group=self.group, no_clamp=self.no_clamp)
            init_kappa, m0_omega, _ = m0.optimize()
            self.n_calls += m0.n_calls
            self.n_vectors += m0.n_vectors
        K = self.n_classes
        bounds = self._build_bounds(K)
        nodes = np.array(self.branch_nodes)
        start_lengths = self.tree.branch_length[nodes].copy()
        best = (-np.inf, None, None, None)
        options = {"maxiter": maxiter, "ftol": 1e-15, "gtol": 1e-10}
        for trial in range(n_restarts):
            self.tree.branch_length[nodes] = start_lengths
            x0, *_ = self._build_init_params(K, init_kappa, m0_omega, rng=np.random.default_rng(seed=trial))
            self._enforce_bounds(x0, bounds)
            self.history = []
            if self.batched_gradient and method == "L-BFGS-B":
                r = minimize(lambda x: value_and_gradient(self.negative_log_likelihoods, x, bounds), x0, jac=True, method=method,
                             bounds=bounds, options=options)
            else:
                r = minimize(self.compute_log_likelihood, x0, method=method, bounds=bounds, options=options)
            if -r.fun > best[0]:
                best = (float(-r.fun), self._extract_result(r.x, K), self.tree.branch_length[nodes].copy(), r)
        self.tree.branch_length[nodes] = best[2]
        self.result = best[3]
        kappa, omegas, props = best[1]
        return kappa, omegas, props, best[0]
