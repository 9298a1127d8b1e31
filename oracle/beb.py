"""CPU restatement of the reference's PAML-grid BEB integration (src/crabml/analysis/beb.py:575-661).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: imported by tests/ and never by crabml_amd/.

Input is what ``self.calc.compute_site_log_likelihoods`` hands the reference at beb.py:577: the [S, K] table of
ln max(f_{k,s}, 1e-300).  Two forms:
  * ``site_posteriors_loops``   the reference's control flow line by line (running maximum S1, the 1e-300 guards,
                                grid order) -- pure-Python loops, for small cases only;
  * ``site_posteriors``         the same sums vectorised over sites (chunked), for sizes the loops cannot reach.
Pinned by tests/golden/lysozyme_m2a_beb.npz, which tests/golden/make_golden_beb.py produced by running the
reference's own ``BEBCalculator.compute_site_posteriors_paml`` in the build container.
"""
from __future__ import annotations

import numpy as np


def scale_table(site_log_liks):
    """beb.py:581-585: fhK[h, :] = exp(lnf[h, :] - max_k lnf[h, k])."""
    x = np.asarray(site_log_liks, dtype=np.float64)
    return np.exp(x - x.max(axis=1, keepdims=True))


def site_posteriors_loops(site_log_liks, cols, props):
    """beb.py:587-661, loop for loop.  cols/props: [G, n_mix] (for M2a: columns (w0_idx, 10, 11 + w2_idx), proportions
    (p0, p1, p2)).  Returns (posteriors [S, n_mix], lnfXs [G], ln f(X))."""
    fhK = scale_table(site_log_liks)
    S = fhK.shape[0]
    G, n_mix = np.shape(cols)
    lnfXs = np.zeros(G)
    for g in range(G):                                   # :591-607
        for h in range(S):
            fh = 0.0
            for c in range(n_mix):
                fh = fh + props[g][c] * fhK[h, cols[g][c]] if c else props[g][c] * fhK[h, cols[g][c]]
            if fh > 1e-300:
                lnfXs[g] += np.log(fh)
    S2 = np.max(lnfXs)                                   # :609-611
    fX = np.log(np.sum(np.exp(lnfXs - S2))) + S2
    post = np.zeros((S, n_mix))
    for h in range(S):                                   # :618-659
        post_site = np.zeros(n_mix)
        S1 = -1e300
        for g in range(G):
            terms = [props[g][c] * fhK[h, cols[g][c]] for c in range(n_mix)]
            fh = terms[0]
            for c in range(1, n_mix):
                fh = fh + terms[c]
            if fh < 1e-300:
                continue
            for c in range(n_mix):
                fh1 = terms[c]
                if fh1 < 1e-300:
                    continue
                fh1 /= fh
                t = np.log(fh1) + lnfXs[g]
                if t > S1:
                    post_site *= np.exp(S1 - t)
                    S1 = t
                post_site[c] += np.exp(t - S1)
        post_site *= np.exp(S1 - fX)
        post[h, :] = post_site
    return post, lnfXs, fX


def site_posteriors(site_log_liks, cols, props, weights=None, chunk=256):
    """Same quantities with the site loops vectorised.  ``weights`` (pattern multiplicities) lets the table hold
    unique patterns instead of sites: lnfX_g = sum_p w_p ln f_p(g)."""
    fhK = scale_table(site_log_liks)
    cols = np.asarray(cols)
    props = np.asarray(props, dtype=np.float64)
    S = fhK.shape[0]
    G, n_mix = cols.shape
    w = np.ones(S) if weights is None else np.asarray(weights, dtype=np.float64)
    lnfXs = np.zeros(G)
    for lo in range(0, S, chunk):
        F = fhK[lo:lo + chunk]
        fh = props[None, :, 0] * F[:, cols[:, 0]]
        for c in range(1, n_mix):
            fh = fh + props[None, :, c] * F[:, cols[:, c]]
        with np.errstate(divide="ignore"):
            lnfXs += (w[lo:lo + chunk, None] * np.where(fh > 1e-300, np.log(np.where(fh > 1e-300, fh, 1.0)), 0.0)).sum(axis=0)
    S2 = lnfXs.max()
    fX = np.log(np.exp(lnfXs - S2).sum()) + S2
    gw = np.exp(lnfXs - fX)
    post = np.zeros((S, n_mix))
    for lo in range(0, S, chunk):
        F = fhK[lo:lo + chunk]
        terms = [props[None, :, c] * F[:, cols[:, c]] for c in range(n_mix)]
        fh = terms[0]
        for c in range(1, n_mix):
            fh = fh + terms[c]
        ok = ~(fh < 1e-300)
        safe = np.where(ok, fh, 1.0)
        for c in range(n_mix):
            keep = ok & ~(terms[c] < 1e-300)
            post[lo:lo + chunk, c] = (np.where(keep, terms[c] / safe, 0.0) * gw[None, :]).sum(axis=1)
    return post, lnfXs, fX


def site_posteriors_weighted(tables, proportions, weights):
    """beb.py:689-731 (the adaptive-grid accumulation): ``tables[b]`` is the [S, K] table of ln max(f, 1e-300) the
    reference gets from compute_site_log_likelihoods at grid point b (:712), ``proportions[b]`` its class proportions,
    ``weights[b]`` the grid weight.  Statement for statement."""
    post = None
    for lnf, props, w in zip(tables, proportions, weights):
        site_liks = np.exp(np.asarray(lnf, dtype=np.float64))            # :718
        site_liks = site_liks * np.asarray(props, dtype=np.float64)[None, :]   # :720-721
        row_sums = np.maximum(site_liks.sum(axis=1, keepdims=True), 1e-300)     # :724-725
        contrib = w * (site_liks / row_sums)                            # :726-729
        post = contrib if post is None else post + contrib
    return post
