#!/usr/bin/env python3
"""Golden vectors for the BEB grid integration, from the reference's own code.

Runs ONLY in the build container (reference checkout at /root/reference).  It imports the reference's
``crabml.analysis.beb`` and calls ``BEBCalculator.setup_paml_grid`` and ``BEBCalculator.compute_site_posteriors_paml``
unchanged on the lysozyme M2a maximum-likelihood estimates of tests/golden/lysozyme_m2a.npz.  The one thing the
reference cannot do here is evaluate ``self.calc.compute_site_log_likelihoods`` (its calculator is the Rust
extension, which cannot be built in this image), so ``calc`` is an adapter over the CPU oracle
(oracle/crab_oracle.c, itself pinned against the reference's Python calculator): the [S, 21] table it returned is
stored in the fixture next to the posteriors the reference computed from it, so the integration is pinned on exactly
the reference's input.  Output: tests/golden/lysozyme_m2a_beb.npz (arrays only).

    python tests/golden/make_golden_beb.py        # about a minute: the reference loops over 130 sites x 10 000 points in Python
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _import_reference_beb():
    pkg = types.ModuleType("crabml")
    pkg.__path__ = [f"{REF}/src/crabml"]
    sys.modules["crabml"] = pkg
    ana = types.ModuleType("crabml.analysis")       # skip analysis/__init__.py (it pulls in plotting libraries)
    ana.__path__ = [f"{REF}/src/crabml/analysis"]
    sys.modules["crabml.analysis"] = ana
    import crabml.analysis.beb as beb
    import crabml.models.codon as codon
    return beb, codon


def main():
    beb, codon = _import_reference_beb()
    from oracle import oracle as orc

    z = np.load(os.path.join(HERE, "lysozyme_m2a.npz"), allow_pickle=False)
    pi, parent, bl, seq = z["pi"], z["parent"], z["branch_lengths"], z["sequences"]
    ts = [(i, None if p < 0 else int(p)) for i, p in enumerate(parent)]
    leaf = list(range(seq.shape[0]))
    mle = dict(kappa=5.03734, omega0=0.35983, omega2=4.52341, p0=0.77376, p1=0.10399)   # tests/test_rust/test_paml_rust_validation.py

    captured = {}

    class Calc:   # stands in for RustLikelihoodCalculator.compute_site_log_likelihoods (likelihood_rust.py:300-364)
        def compute_site_log_likelihoods(self, Q_matrices, pi_, proportions):
            out = orc.compute_site_log_likelihoods_by_class(list(Q_matrices), pi_, ts, bl, None, seq, leaf)
            captured["Q"] = np.stack(Q_matrices)
            captured["site_log_liks"] = out.copy()
            return out

    calc = beb.BEBCalculator.__new__(beb.BEBCalculator)
    calc.mle_params = mle
    calc.alignment = types.SimpleNamespace(n_sites=seq.shape[1], n_species=seq.shape[0])
    calc.tree = None
    calc.model_class = codon.M2aCodonModel
    calc.param_names = ["kappa", "omega0", "omega2", "p0", "p1"]
    calc.grid_type = "paml"
    calc.pi = pi
    calc.calc = Calc()
    calc.mle_model = codon.M2aCodonModel(**mle, pi=pi)

    with contextlib.redirect_stdout(io.StringIO()):
        grid_points, weights = calc.setup_paml_grid()
        post = calc.compute_site_posteriors_paml(grid_points, weights)
    grid = np.asarray(grid_points)                     # [G, 5] in param_names order
    props_mle, omegas_mle = calc.mle_model.get_site_classes()
    np.savez_compressed(os.path.join(HERE, "lysozyme_m2a_beb.npz"),
                        pi=pi, parent=parent, branch_lengths=bl, sequences=seq,
                        kappa=mle["kappa"], mle_proportions=np.asarray(props_mle), mle_omegas=np.asarray(omegas_mle),
                        Q=captured["Q"], site_log_liks=captured["site_log_liks"],
                        grid_p0=grid[:, 3], grid_p1=grid[:, 4], grid_omega0=grid[:, 1], grid_omega2=grid[:, 2],
                        posteriors_ref=post)
    # ---- the adaptive-grid accumulation, BEBCalculator.compute_site_posteriors (beb.py:663-731), on a small
    #      hand-made grid around the MLE (the reference derives its grid from a Hessian; the accumulation is what is
    #      pinned here), with M8 so that the beta quantile step is inside the reference's model class ----
    mle8 = dict(kappa=5.029752, p0=0.868486, p_beta=9.517504, q_beta=13.543157, omega_s=4.370617)
    tables = []

    class Calc8:
        def compute_site_log_likelihoods(self, Q_matrices, pi_, proportions):
            out = orc.compute_site_log_likelihoods_by_class(list(Q_matrices), pi_, ts, bl, None, seq, leaf)
            tables.append(out.copy())
            return out

    c8 = beb.BEBCalculator.__new__(beb.BEBCalculator)
    c8.mle_params = dict(mle8, ncatG=10)
    c8.alignment = calc.alignment
    c8.model_class = codon.M8CodonModel
    c8.param_names = ["kappa", "p0", "p_beta", "q_beta", "omega_s"]
    c8.pi = pi
    c8.calc = Calc8()
    c8.mle_model = codon.M8CodonModel(**c8.mle_params, pi=pi)
    rng = np.random.default_rng(8)
    base = np.array([mle8[k] for k in c8.param_names])
    pts = [base] + [base * np.exp(rng.normal(0.0, 0.15, size=5)) for _ in range(8)]
    for p_ in pts:
        p_[1] = min(p_[1], 0.99)
    wts = rng.dirichlet(np.ones(len(pts)))
    with contextlib.redirect_stdout(io.StringIO()):
        post8 = c8.compute_site_posteriors(pts, list(wts))
    cls = [codon.M8CodonModel(**c8._array_to_params(p_), pi=pi).get_site_classes() for p_ in pts]
    np.savez_compressed(os.path.join(HERE, "lysozyme_m8_beb_grid.npz"),
                        pi=pi, parent=parent, branch_lengths=bl, sequences=seq, grid_points=np.asarray(pts), weights=wts,
                        grid_kappa=np.asarray([p_[0] for p_ in pts]), grid_omegas=np.asarray([c[1] for c in cls]),
                        grid_proportions=np.asarray([c[0] for c in cls]), tables=np.asarray(tables), posteriors_ref=post8)
    # ---- the adaptive grid itself: BEBCalculator.compute_hessian and setup_integration_grid (beb.py:234-387) at the
    #      M2a estimates, three grid points per axis; every likelihood the reference asked for is kept with the
    #      parameters it was asked at, so that the stencil values, the Hessian and the grid are all pinned ----
    asked = []

    class CalcH:   # stands in for RustLikelihoodCalculator.compute_log_likelihood_site_classes (likelihood_rust.py:232-298)
        def compute_log_likelihood_site_classes(self, Q_matrices, pi_, proportions):
            v = orc.compute_site_class_log_likelihood(list(Q_matrices), proportions, pi_, ts, bl, None, seq, leaf)
            asked.append((np.stack(Q_matrices), np.asarray(proportions, dtype=np.float64), v))
            return v

    calc.calc = CalcH()
    calc.n_grid_points = 3
    calc.grid_type = "adaptive"
    with contextlib.redirect_stdout(io.StringIO()):
        hess = calc.compute_hessian()
        gpts, gw = calc.setup_integration_grid(hess)
        calc.calc = Calc()
        post_ad = calc.compute_site_posteriors(gpts, gw)
    np.savez_compressed(os.path.join(HERE, "lysozyme_m2a_adaptive_grid.npz"),
                        pi=pi, parent=parent, branch_lengths=bl, sequences=seq,
                        param_names=np.array(calc.param_names), mle=calc._params_to_array(mle),
                        asked_Q0=np.asarray([a[0][0] for a in asked]), asked_proportions=np.asarray([a[1] for a in asked]),
                        asked_lnl=np.asarray([a[2] for a in asked]), hessian_ref=hess, grid_points_ref=np.asarray(gpts),
                        weights_ref=np.asarray(gw), posteriors_ref=post_ad)
    print("adaptive grid fixture:", len(asked), "likelihoods asked,", len(gpts), "grid points; Hessian diagonal", np.round(np.diag(hess), 2))
    print("adaptive-path fixture:", post8.shape, "row sums", post8.sum(axis=1)[:3])
    print("posterior P(omega>1) of the sites PAML lists (15 17 37 41 50 62 114 126):",
          np.round(post[[14, 16, 36, 40, 49, 61, 113, 125], 2], 3))


if __name__ == "__main__":
    main()
