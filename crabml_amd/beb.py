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
