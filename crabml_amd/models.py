"""Codon-model input producers for the likelihood boundary.

Host-side mirror of what crabML's model classes hand to the native core: the
61x61 rate matrices Q (per site class / per branch), class proportions, Model A
class frequencies and Qfactors.  Formulas follow the reference
(src/crabml/models/codon.py:166-246 Q construction, :91-152 F3x4,
:337-376/:435-461/:514-536/:581-627/:681-729/:782-830 site-class builders,
src/crabml/models/codon_branch_site.py:108-231 Model A,
src/crabml/models/codon_branch.py:129-161 per-node Q stack); the code is a
vectorised restatement around a precomputed 61x61 neighbour mask.
"""
from __future__ import annotations

import functools

import numpy as np

from .io import CODONS, GENETIC_CODE, N_CODONS

# ---------------------------------------------------------------------------
# codon graph: 526 ordered single-nucleotide neighbour pairs among sense codons
# ---------------------------------------------------------------------------
_TS = {("A", "G"), ("G", "A"), ("C", "T"), ("T", "C")}


def _codon_graph():
    nb = np.zeros((N_CODONS, N_CODONS), dtype=bool)
    ts = np.zeros_like(nb)
    nonsyn = np.zeros_like(nb)
    for i, a in enumerate(CODONS):
        for j, b in enumerate(CODONS):
            diff = [k for k in range(3) if a[k] != b[k]]
            if len(diff) != 1:
                continue
            nb[i, j] = True
            ts[i, j] = (a[diff[0]], b[diff[0]]) in _TS
            nonsyn[i, j] = GENETIC_CODE[a] != GENETIC_CODE[b]
    return nb, ts, nonsyn


NEIGHBOUR, TRANSITION, NONSYN = _codon_graph()


def build_codon_Q_matrix(kappa: float, omega: float, pi: np.ndarray, normalization_factor=None) -> np.ndarray:
    """Q_ij = kappa^[ts] * omega^[nonsyn] * pi_j for single-nucleotide neighbours;
    Q_ii = -sum_j Q_ij; divided by ``normalization_factor`` if given, else by the
    mean rate -sum_i pi_i Q_ii  (reference: models/codon.py:166-246, core/matrix.py:153-170)."""
    pi = np.asarray(pi, dtype=np.float64)
    S = np.zeros((N_CODONS, N_CODONS))
    S[NEIGHBOUR] = 1.0
    S[TRANSITION] *= kappa
    S[NONSYN] *= omega
    Q = S * pi[np.newaxis, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    if normalization_factor is not None:
        return Q / normalization_factor
    return Q / (-np.dot(pi, Q.diagonal()))


def mean_rate(Q: np.ndarray, pi: np.ndarray) -> float:
    return float(-np.dot(pi, np.diag(Q)))


def compute_codon_frequencies_f3x4(alignment) -> np.ndarray:
    """F3x4: position-specific nucleotide frequencies over sense codons only
    (codes 0..60), pi_c proportional to f1[c1] f2[c2] f3[c3], renormalised over the 61
    (reference: models/codon.py:91-152)."""
    codes = np.asarray(alignment.sequences)
    valid = codes[(codes >= 0) & (codes < N_CODONS)].astype(np.int64)
    nuc = {"T": 0, "C": 1, "A": 2, "G": 3}
    pos_idx = np.array([[nuc[c[k]] for k in range(3)] for c in CODONS])  # [61, 3]
    codon_counts = np.bincount(valid, minlength=N_CODONS).astype(np.float64)
    counts = np.zeros((3, 4))
    for k in range(3):
        np.add.at(counts[k], pos_idx[:, k], codon_counts)
    f = counts / counts.sum(axis=1, keepdims=True)
    pi = f[0, pos_idx[:, 0]] * f[1, pos_idx[:, 1]] * f[2, pos_idx[:, 2]]
    return pi / pi.sum()


def _normalise_pi(pi):
    if pi is None:
        return np.ones(N_CODONS) / N_CODONS
    pi = np.asarray(pi, dtype=np.float64)
    if len(pi) != N_CODONS:
        raise ValueError(f"pi must have length 61, got {len(pi)}")
    return pi / pi.sum()


class M0CodonModel:
    def __init__(self, kappa=2.0, omega=0.4, pi=None):
        self.kappa, self.omega, self.pi = kappa, omega, _normalise_pi(pi)

    def get_Q_matrix(self) -> np.ndarray:
        return build_codon_Q_matrix(self.kappa, self.omega, self.pi)


class _SiteClassModel:
    """Shared-scale mixture: Q_k <- Q_k / sum_k p_k (-sum_i pi_i Q_k,ii)
    (reference: e.g. models/codon.py:709-729)."""

    kappa: float
    pi: np.ndarray

    def get_site_classes(self):
        raise NotImplementedError

    def get_Q_matrices(self):
        props, omegas = self.get_site_classes()
        raw = [build_codon_Q_matrix(self.kappa, w, self.pi, normalization_factor=1.0) for w in omegas]
        scale = sum(p * mean_rate(Q, self.pi) for p, Q in zip(props, raw))
        return [Q / scale for Q in raw]


def _median_quantiles(ppf, K):
    return [float(ppf((2.0 * j + 1.0) / (2.0 * K))) for j in range(K)]


class M1aCodonModel(_SiteClassModel):
    def __init__(self, kappa=2.0, omega0=0.5, p0=0.7, pi=None):
        self.kappa = kappa
        self.omega0 = min(omega0, 0.999)
        self.p0 = float(np.clip(p0, 1e-6, 1 - 1e-6))
        self.pi = _normalise_pi(pi)

    def get_site_classes(self):
        return [self.p0, 1.0 - self.p0], [self.omega0, 1.0]


class M2aCodonModel(_SiteClassModel):
    def __init__(self, kappa=2.0, omega0=0.5, omega2=2.0, p0=0.5, p1=0.3, pi=None):
        self.kappa = kappa
        self.omega0 = min(omega0, 0.999)
        self.omega2 = max(omega2, 1.001)
        tot = p0 + p1
        if tot >= 1.0:
            p0, p1 = p0 / (tot + 0.01) * 0.99, p1 / (tot + 0.01) * 0.99
        self.p0 = float(np.clip(p0, 0.001, 0.998))
        self.p1 = float(np.clip(p1, 0.001, 0.998 - self.p0))
        self.pi = _normalise_pi(pi)

    def get_site_classes(self):
        return [self.p0, self.p1, 1.0 - self.p0 - self.p1], [self.omega0, 1.0, self.omega2]


class M3CodonModel(_SiteClassModel):
    def __init__(self, kappa=2.0, omegas=None, proportions=None, pi=None):
        self.kappa = kappa
        self.omegas = list(omegas) if omegas is not None else [0.5, 1.0, 2.0]
        p = np.asarray(proportions if proportions is not None else [1.0 / len(self.omegas)] * len(self.omegas), dtype=float)
        if len(p) != len(self.omegas):
            raise ValueError("omegas and proportions must have same length")
        self.proportions = (p / p.sum()).tolist()
        self.pi = _normalise_pi(pi)

    def get_site_classes(self):
        return self.proportions, self.omegas


class M4CodonModel(M3CodonModel):
    def __init__(self, kappa=2.0, proportions=None, pi=None):
        if proportions is not None and len(proportions) != 5:
            raise ValueError(f"M4 requires exactly 5 proportions, got {len(proportions)}")
        super().__init__(kappa, [0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0, 3.0],
                         proportions if proportions is not None else [0.2] * 5, pi)


class M5CodonModel(_SiteClassModel):
    def __init__(self, kappa=2.0, alpha=1.0, beta=1.0, ncatG=10, pi=None):
        self.kappa, self.alpha, self.beta, self.ncatG = kappa, max(alpha, 0.005), max(beta, 0.005), ncatG
        self.pi = _normalise_pi(pi)

    def get_site_classes(self):
        from scipy.stats import gamma
        K = self.ncatG
        return [1.0 / K] * K, _median_quantiles(lambda p: gamma.ppf(p, self.alpha, scale=1.0 / self.beta), K)


@functools.lru_cache(maxsize=256)
def _beta_quantiles(p_beta, q_beta, K):
    """Median quantiles of Beta(p, q); remembered, because the rows of a finite-difference stencil mostly share them."""
    from scipy.stats import beta
    return tuple(_median_quantiles(lambda p: beta.ppf(p, p_beta, q_beta), K))


@functools.lru_cache(maxsize=256)
def _beta_quantiles_c(p_beta, q_beta, K):
    """The same quantiles from the C library (crab_discretise_omega: within 2e-11 relative of scipy over the optimisers'
    range, tests/test_host_logic.py; 5 us per quantile against 120 us for scipy.stats.beta.ppf, which at gene scale was
    most of the host time of a fit).  Used by the optimiser mirrors; fixtures and generators stay on scipy."""
    from .engine import discretise_omega
    return tuple(float(x) for x in discretise_omega("beta", p_beta, q_beta, K))


class M7CodonModel(_SiteClassModel):
    c_quantiles = False    # instances made by the optimiser mirrors set this

    def __init__(self, kappa=2.0, p_beta=0.5, q_beta=0.5, ncatG=10, pi=None):
        self.kappa, self.p_beta, self.q_beta, self.ncatG = kappa, max(p_beta, 0.005), max(q_beta, 0.005), ncatG
        self.pi = _normalise_pi(pi)

    def _beta_omegas(self):
        return list((_beta_quantiles_c if self.c_quantiles else _beta_quantiles)(self.p_beta, self.q_beta, self.ncatG))

    def get_site_classes(self):
        K = self.ncatG
        return [1.0 / K] * K, self._beta_omegas()


class M8CodonModel(M7CodonModel):
    def __init__(self, kappa=2.0, p0=0.9, p_beta=0.5, q_beta=0.5, omega_s=2.0, ncatG=10, pi=None):
        super().__init__(kappa, p_beta, q_beta, ncatG, pi)
        self.p0 = float(np.clip(p0, 1e-6, 1 - 1e-6))
        self.omega_s = max(omega_s, 1.001)

    def get_site_classes(self):
        K = self.ncatG
        return [self.p0 / K] * K + [1.0 - self.p0], self._beta_omegas() + [self.omega_s]


class M8aCodonModel(M8CodonModel):
    def __init__(self, kappa=2.0, p0=0.9, p_beta=0.5, q_beta=0.5, ncatG=10, pi=None):
        super().__init__(kappa, p0, p_beta, q_beta, 2.0, ncatG, pi)
        self.omega_s = 1.0


def _mixture_quantiles(cdf, K):
    from scipy.optimize import brentq
    out = []
    for j in range(K):
        p = (2.0 * j + 1.0) / (2.0 * K)
        try:
            out.append(float(brentq(lambda x: cdf(x) - p, 0.0001, 99.0)))
        except ValueError:
            out.append(p * 10)
    return out


@functools.lru_cache(maxsize=256)
def _mixture_quantiles_c(kind, params, K):
    """The same quantiles from the C library (crab_discretise_omega_mixture: Brent's method as scipy.optimize.brentq runs it;
    equal to the scipy route to ~1e-14 on the fixtures, tests/test_host_logic.py).  Used by the optimiser mirrors."""
    from .engine import discretise_omega_mixture
    return tuple(float(x) for x in discretise_omega_mixture(kind, list(params), K))


class M6CodonModel(_SiteClassModel):
    c_quantiles = False    # instances made by the optimiser mirrors set this

    def __init__(self, kappa=2.0, p0=0.5, alpha1=1.0, beta1=1.0, alpha2=1.0, ncatG=10, pi=None):
        self.kappa = kappa
        self.p0 = float(np.clip(p0, 1e-6, 1 - 1e-6))
        self.alpha1, self.beta1, self.alpha2 = max(alpha1, 0.005), max(beta1, 0.005), max(alpha2, 0.005)
        self.ncatG = ncatG
        self.pi = _normalise_pi(pi)

    def get_site_classes(self):
        K = self.ncatG
        if self.c_quantiles:
            return [1.0 / K] * K, list(_mixture_quantiles_c("M6", (self.p0, self.alpha1, self.beta1, self.alpha2), K))
        from scipy.stats import gamma
        cdf = lambda x: (self.p0 * gamma.cdf(x, self.alpha1, scale=1.0 / self.beta1)
                         + (1 - self.p0) * gamma.cdf(x, self.alpha2, scale=1.0 / self.alpha2))
        return [1.0 / K] * K, _mixture_quantiles(cdf, K)


class M9CodonModel(_SiteClassModel):
    c_quantiles = False

    def __init__(self, kappa=2.0, p0=0.5, p_beta=0.5, q_beta=0.5, alpha=1.0, beta_gamma=1.0, ncatG=10, pi=None):
        self.kappa = kappa
        self.p0 = float(np.clip(p0, 1e-6, 1 - 1e-6))
        self.p_beta, self.q_beta = max(p_beta, 0.005), max(q_beta, 0.005)
        self.alpha, self.beta_gamma = max(alpha, 0.005), max(beta_gamma, 0.005)
        self.ncatG = ncatG
        self.pi = _normalise_pi(pi)

    def get_site_classes(self):
        K = self.ncatG
        if self.c_quantiles:
            return [1.0 / K] * K, list(_mixture_quantiles_c("M9", (self.p0, self.p_beta, self.q_beta, self.alpha, self.beta_gamma), K))
        from scipy.stats import beta, gamma
        cdf = lambda x: (self.p0 * beta.cdf(x, self.p_beta, self.q_beta)
                         + (1 - self.p0) * gamma.cdf(x, self.alpha, scale=1.0 / self.beta_gamma))
        return [1.0 / K] * K, _mixture_quantiles(cdf, K)


# ---------------------------------------------------------------------------
# Branch-site Model A (Zhang, Nielsen & Yang 2005)
# ---------------------------------------------------------------------------
class BranchSiteModelA:
    """Class frequencies [p0, p1, p2 p0/(p0+p1), p2 p1/(p0+p1)] and the two
    Qfactors 1/mean_rate(Q(kappa, omega_bar)) (reference: codon_branch_site.py:108-231)."""

    n_site_classes = 4

    def __init__(self, codon_frequencies, branch_labels, fix_omega=False):
        self.pi = np.asarray(codon_frequencies, dtype=np.float64)
        self.branch_labels = np.asarray(branch_labels, dtype=int)
        self.fix_omega = fix_omega
        uniq = np.unique(self.branch_labels)
        if not np.array_equal(uniq, np.array([0, 1])):
            raise ValueError(f"Branch-site models require exactly 2 branch types (0 and 1). Found: {uniq}")

    def compute_site_class_frequencies(self, p0, p1):
        if p0 + p1 >= 1.0:
            raise ValueError(f"p0 + p1 must be < 1 (got {p0 + p1})")
        if p0 + p1 < 1e-10:
            raise ValueError(f"p0 + p1 too small ({p0 + p1})")
        p2, t = 1.0 - p0 - p1, p0 + p1
        return np.array([p0, p1, p2 * p0 / t, p2 * p1 / t])

    def compute_qfactors(self, kappa, p0, p1, omega0, omega2):
        p2, t = 1.0 - p0 - p1, p0 + p1
        if t < 1e-10:
            raise ValueError("p0 + p1 too small for Qfactor computation")
        w_back = (p0 / t) * omega0 + (p1 / t) * 1.0
        w_fore = p0 * omega0 + p1 * 1.0 + p2 * omega2
        qb = 1.0 / mean_rate(build_codon_Q_matrix(kappa, w_back, self.pi, 1.0), self.pi)
        qf = 1.0 / mean_rate(build_codon_Q_matrix(kappa, w_fore, self.pi, 1.0), self.pi)
        return qb, qf

    @staticmethod
    def get_omega_for_branch_class(branch_label, site_class, omega0, omega2):
        if branch_label not in (0, 1):
            raise ValueError(f"Invalid branch_label: {branch_label}")
        if site_class not in (0, 1, 2, 3):
            raise ValueError(f"Invalid site_class: {site_class}")
        if site_class == 0:
            return omega0
        if site_class == 1:
            return 1.0
        if branch_label == 1:
            return omega2
        return omega0 if site_class == 2 else 1.0

    def unnormalised_Qs(self, kappa, omega0, omega2):
        return tuple(build_codon_Q_matrix(kappa, w, self.pi, 1.0) for w in (omega0, 1.0, omega2))


# ---------------------------------------------------------------------------
# Branch (free-ratio / multi-ratio) model
# ---------------------------------------------------------------------------
class CodonBranchModel:
    """One individually normalised Q per node (reference: codon_branch.py:55-161).
    ``branch_labels`` has one entry per node (root included, last)."""

    def __init__(self, codon_frequencies, branch_labels, free_ratio=False, n_nodes=None):
        self.pi = np.asarray(codon_frequencies, dtype=np.float64)
        self.branch_labels = np.asarray(branch_labels, dtype=int)
        self.free_ratio = free_ratio
        self.n_nodes = len(self.branch_labels) if n_nodes is None else n_nodes
        if len(self.branch_labels) != self.n_nodes:
            raise ValueError(f"branch_labels length ({len(self.branch_labels)}) must match n_nodes ({self.n_nodes})")
        if free_ratio:
            self.n_omega = self.n_nodes - 1
            self.omega_labels = np.arange(self.n_nodes)
            self.omega_labels[-1] = 0
        else:
            uniq = np.unique(self.branch_labels)
            self.n_omega = len(uniq)
            self.omega_labels = np.searchsorted(uniq, self.branch_labels)

    def get_Q_matrices(self, kappa, omega_values) -> np.ndarray:
        omega_values = np.asarray(omega_values, dtype=np.float64)
        if len(omega_values) != self.n_omega:
            raise ValueError(f"Expected {self.n_omega} omega values, got {len(omega_values)}")
        cache = {}
        out = np.empty((self.n_nodes, N_CODONS, N_CODONS))
        for v in range(self.n_nodes):
            w = float(omega_values[self.omega_labels[v]])
            if w not in cache:
                cache[w] = build_codon_Q_matrix(kappa, w, self.pi)
            out[v] = cache[w]
        return out
