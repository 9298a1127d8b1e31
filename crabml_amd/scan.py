"""Gene-by-gene tests for positive selection: the M7-M8 (and M8a-M8) likelihood-ratio scan of BASELINE config 5.

Genes are independent problems: they are dealt round-robin to the ranks (``dist.gene_shard``; no collective on the data
path, one gather of the result rows at the end) and, within a rank, to a few engines that share the GPU
(``dist.map_genes``: one gene fills a fraction of an MI355X).  Each fit is the optimiser mirror's, with the whole
finite-difference gradient in one call.  The statistics follow the reference's ``analysis/lrt.py``: ``calculate_lrt``
(:8-64, chi-square with df degrees of freedom) and ``calculate_lrt_mixture`` (:67-138, the 50:50 mixture of a point mass
at zero and chi-square for a null on the boundary, M8a against M8).
"""
from __future__ import annotations

import warnings

from scipy.stats import chi2

from .dist import gene_shard, map_genes
from .optimize import M7Optimizer, M8aOptimizer, M8Optimizer


def calculate_lrt(lnL_null, lnL_alt, df):
    stat = 2 * (lnL_alt - lnL_null)
    if stat < 0:
        warnings.warn(f"Negative LRT detected (LRT={stat:.6f}); setting LRT=0 and p-value=1.0. Check convergence.", UserWarning)
        return 0.0, 1.0
    return (stat, 1.0) if stat == 0 else (stat, float(chi2.sf(stat, df)))


def calculate_lrt_mixture(lnL_null, lnL_alt, df):
    stat = 2 * (lnL_alt - lnL_null)
    if stat < 0:
        warnings.warn(f"Negative LRT detected (LRT={stat:.6f}); setting LRT=0 and p-value=1.0. Check convergence.", UserWarning)
        return 0.0, 1.0
    return (stat, 1.0) if stat == 0 else (stat, 0.5 * float(chi2.sf(stat, df)))


def lrt_gene(engine, alignment, tree, null: str = "M7", ncatG: int = 10, maxiter: int = 200, **optimizer_kw):
    """Fit the null (M7 or M8a) and M8 to one gene on ``engine`` and compare them.  The tree is copied per fit, so both
    start from the caller's branch lengths (each then takes its own M0 start, as the reference's optimisers do)."""
    import copy
    fits = {}
    for name, cls in ((null, {"M7": M7Optimizer, "M8a": M8aOptimizer}[null]), ("M8", M8Optimizer)):
        opt = cls(alignment, copy.deepcopy(tree), ncatG=ncatG, engine=engine, **optimizer_kw)
        out = opt.optimize(maxiter=maxiter)
        fits[name] = {"lnL": out[-1], "parameters": out[:-1], "vectors": opt.n_vectors, "calls": opt.n_calls}
    stat, p = (calculate_lrt(fits["M7"]["lnL"], fits["M8"]["lnL"], 2) if null == "M7"
               else calculate_lrt_mixture(fits["M8a"]["lnL"], fits["M8"]["lnL"], 1))
    return {"null": null, "lnL_null": fits[null]["lnL"], "lnL_M8": fits["M8"]["lnL"], "LRT": stat, "pvalue": p, "fits": fits}


def scan(genes, null: str = "M7", rank: int = 0, world_size: int = 1, workers: int = 8, group=None, make_engine=None, **kw):
    """``genes``: a list of (alignment, tree), the same list on every rank.  Returns the per-gene results in order on
    every rank; a gene whose input is unusable yields ``{"null": ..., "error": message}`` instead of ending the scan, a
    device error (CRAB_ERR_DEVICE) ends it (``torch.distributed.all_gather_object`` of this rank's rows when world_size > 1).
    Optimiser keywords pass through ``kw`` (e.g. ``no_clamp=True`` for trees of several hundred taxa)."""
    mine = list(gene_shard(len(genes), rank, world_size))
    def one(eng, i):
        from ._lib import CRAB_ERR_DEVICE, CrabError
        try:
            return lrt_gene(eng, genes[i][0], genes[i][1], null=null, **kw)
        except CrabError as e:
            if e.status == CRAB_ERR_DEVICE:   # a HIP failure poisons the engine and every gene after it: stop the scan, loudly
                raise
            return {"null": null, "error": f"{type(e).__name__}: {e}"}
        except (ValueError, KeyError, IndexError, ArithmeticError) as e:   # one unusable gene (names that do not match, no
            return {"null": null, "error": f"{type(e).__name__}: {e}"}     # codons) must not end a 10 000-gene scan

    rows = map_genes(one, mine, workers=workers, make_engine=make_engine)
    out = [None] * len(genes)
    if world_size > 1:
        import torch.distributed as dist
        parts = [None] * world_size
        dist.all_gather_object(parts, list(zip(mine, rows)), group=group)
        for part in parts:
            for i, r in part:
                out[i] = r
    else:
        for i, r in zip(mine, rows):
            out[i] = r
    return out
