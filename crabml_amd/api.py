"""``optimize_model`` and the positive-selection tests by name: the reference's entry points over the optimiser mirrors.

Mirror of src/crabml/api.py (``optimize_model`` :1021-1211, loaders :908-1018, result parsers :615-837: the same
``params`` keys and parameter counts) and of analysis/positive_selection.py's three tests (``m1a_vs_m2a`` :15,
``m7_vs_m8`` :200, ``m8a_vs_m8`` :320) reduced to what the likelihood path feeds: the two fits and the test statistic.
Report formatting, JSON export and the CLI stay with the reference (out of scope).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Optional

from . import optimize as _opt
from .io import Alignment, Tree
from .scan import calculate_lrt, calculate_lrt_mixture


@dataclass
class SiteModelResult:
    model_name: str
    lnL: float
    kappa: float
    params: Dict[str, Any]
    tree: Tree
    alignment: Alignment
    n_params: int
    convergence_info: Optional[Dict[str, Any]] = None

    @property
    def omega(self):
        """The single omega of M0, None for the other models (api.py:190-203)."""
        return self.params.get("omega") if self.model_name == "M0" else None

    def to_dict(self):
        return {"model_name": self.model_name, "lnL": self.lnL, "kappa": self.kappa, "params": dict(self.params),
                "n_params": self.n_params, "tree": self.tree.to_newick()}


def _load_alignment(alignment):
    if isinstance(alignment, Alignment):
        return alignment
    path = Path(alignment)
    if not path.exists():
        raise FileNotFoundError(f"Alignment file not found: {path}")
    readers = [Alignment.from_fasta, Alignment.from_phylip]
    if path.suffix.lower() in (".phy", ".phylip", ".txt"):
        readers.reverse()
    error = None
    for read in readers:                                    # the suffix picks the first try, the other format is the fallback
        try:
            return read(str(path), seqtype="codon")
        except Exception as e:   # noqa: BLE001 -- any parse failure moves on to the other format, as the reference does
            error = error or e
    raise ValueError(f"Failed to load alignment from {path}: {error}")


def _load_tree(tree):
    if isinstance(tree, Tree):
        return tree
    text = str(tree)
    try:
        if len(text) < 4096 and Path(text).exists():
            text = Path(text).read_text().strip()
        return Tree.from_newick(text)
    except Exception as e:   # noqa: BLE001
        raise ValueError(f"Failed to parse tree: {e}")


# model -> (optimiser, leading free parameters, result tuple -> params)
_REGISTRY = {
    "M0": (_opt.M0Optimizer, 2, lambda r: {"omega": r[1]}),
    "M1A": (_opt.M1aOptimizer, 3, lambda r: {"p0": r[2], "p1": 1.0 - r[2], "omega0": r[1]}),
    "M2A": (_opt.M2aOptimizer, 5, lambda r: {"p0": r[3], "p1": r[4], "p2": 1.0 - r[3] - r[4], "omega0": r[1], "omega2": r[2]}),
    "M3": (_opt.M3Optimizer, None, lambda r: {"omegas": list(r[1]), "proportions": list(r[2])}),
    "M7": (_opt.M7Optimizer, 3, lambda r: {"p": r[1], "q": r[2]}),
    "M8": (_opt.M8Optimizer, 5, lambda r: {"p0": r[1], "p": r[2], "q": r[3], "omega_s": r[4]}),
    "M8A": (_opt.M8aOptimizer, 4, lambda r: {"p0": r[1], "p": r[2], "q": r[3], "omega_s": 1.0}),
}


def optimize_model(model: str, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True, **optimizer_kwargs):
    """api.py:1021-1211.  Extra keywords go to ``optimize()`` (``maxiter``, ``init_*``) except ``init_with_m0``, ``n_classes``
    and, here, ``device`` / ``engine`` / ``batched_gradient``, which go to the optimiser's constructor."""
    aln, tree_obj = _load_alignment(alignment), _load_tree(tree)
    name = model.upper()
    if name not in _REGISTRY:
        raise ValueError(f"Unknown model: '{model}'. Valid models are: {', '.join(sorted(_REGISTRY))}")
    cls, n_lead, to_params = _REGISTRY[name]
    init = {"use_f3x4": use_f3x4, "optimize_branch_lengths": optimize_branch_lengths}
    init_with_m0 = optimizer_kwargs.pop("init_with_m0", True)
    n_classes = optimizer_kwargs.pop("n_classes", None)
    if name != "M0":
        init["init_with_m0"] = init_with_m0
    if name == "M3" and n_classes is not None:
        init["n_classes"] = n_classes
    for key in ("device", "engine", "batched_gradient"):
        if key in optimizer_kwargs:
            init[key] = optimizer_kwargs.pop(key)
    opt = cls(aln, tree_obj, **init)
    out = opt.optimize(**optimizer_kwargs)
    if name == "M3":
        K = len(out[1])
        n_lead = 1 + (K - 1) + K
    n_params = n_lead + (opt.n_branches if optimize_branch_lengths else 1)
    return SiteModelResult(model_name=name, lnL=out[-1], kappa=out[0], params=to_params(out), tree=tree_obj, alignment=aln,
                           n_params=n_params, convergence_info={"iterations": int(opt.result.nit), "calls": opt.n_calls,
                                                                "parameter_vectors": opt.n_vectors, "message": str(opt.result.message)})


def _compare(null_name, alt_name, alignment, tree, df, mixture, **kw):
    import copy
    aln, tree_obj = _load_alignment(alignment), _load_tree(tree)
    null = optimize_model(null_name, aln, copy.deepcopy(tree_obj), **kw)
    alt = optimize_model(alt_name, aln, copy.deepcopy(tree_obj), **kw)
    stat, p = (calculate_lrt_mixture if mixture else calculate_lrt)(null.lnL, alt.lnL, df)
    return {"test": f"{null_name} vs {alt_name}", "null": null, "alternative": alt, "LRT": stat, "df": df, "pvalue": p,
            "significant": p < 0.05}


def m1a_vs_m2a(alignment, tree, **kw):
    """analysis/positive_selection.py:15-197: two extra parameters, chi-square with 2 degrees of freedom."""
    return _compare("M1a", "M2a", alignment, tree, 2, False, **kw)


def m7_vs_m8(alignment, tree, **kw):
    """analysis/positive_selection.py:200-317: chi-square with 2 degrees of freedom."""
    return _compare("M7", "M8", alignment, tree, 2, False, **kw)


def m8a_vs_m8(alignment, tree, **kw):
    """analysis/positive_selection.py:320-460: omega_s = 1 is on the boundary, 50:50 mixture of 0 and chi-square(1)."""
    return _compare("M8a", "M8", alignment, tree, 1, True, **kw)


BranchModelResult = BranchSiteModelResult = SiteModelResult   # same fields (api.py:354-612 differ in report formatting only)


def _info(opt):
    return {"iterations": int(opt.result.nit), "calls": opt.n_calls, "parameter_vectors": opt.n_vectors, "message": str(opt.result.message)}


def optimize_branch_model(model: str, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True, **optimizer_kwargs):
    """api.py:1214-1323: 'free-ratio' / 'model1' or 'multi-ratio' / 'model2'; an M0 fit supplies the starting kappa and omega."""
    aln, tree_obj = _load_alignment(alignment), _load_tree(tree)
    names = {"free-ratio": (True, "Free-ratio"), "model1": (True, "Free-ratio"), "multi-ratio": (False, "Multi-ratio"), "model2": (False, "Multi-ratio")}
    if model.lower() not in names:
        raise ValueError(f"Unknown branch model: '{model}'. Valid models are: 'free-ratio', 'multi-ratio'")
    free_ratio, model_name = names[model.lower()]
    init = {k: optimizer_kwargs.pop(k) for k in ("device", "engine", "batched_gradient") if k in optimizer_kwargs}
    kappa0, omega0, _ = _opt.M0Optimizer(aln, tree_obj, use_f3x4=use_f3x4, **init).optimize()
    opt = _opt.BranchModelOptimizer(aln, tree_obj, use_f3x4=use_f3x4, free_ratio=free_ratio, optimize_branch_lengths=optimize_branch_lengths, **init)
    optimizer_kwargs.setdefault("init_kappa", kappa0)
    optimizer_kwargs.setdefault("init_omega", omega0)
    kappa, omega_dict, lnl = opt.optimize(**optimizer_kwargs)
    n_params = 1 + len(omega_dict) + (opt.n_branches if optimize_branch_lengths else 1)
    return BranchModelResult(model_name=model_name, lnL=lnl, kappa=kappa, params={"omega_dict": omega_dict}, tree=tree_obj, alignment=aln,
                             n_params=n_params, convergence_info=_info(opt))


def optimize_branch_site_model(model: str, alignment, tree, use_f3x4: bool = True, optimize_branch_lengths: bool = True,
                               fix_omega: bool = False, **optimizer_kwargs):
    """api.py:1326-1434: 'model-a' / 'branch-site-a' / 'modela'; ``fix_omega`` fits the null (omega2 = 1)."""
    aln, tree_obj = _load_alignment(alignment), _load_tree(tree)
    if model.lower() not in ("model-a", "branch-site-a", "modela"):
        raise ValueError(f"Unknown branch-site model: '{model}'. Valid models are: 'model-a', 'branch-site-a'")
    init = {k: optimizer_kwargs.pop(k) for k in ("device", "engine", "batched_gradient", "live_branch_lengths") if k in optimizer_kwargs}
    opt = _opt.BranchSiteModelAOptimizer(aln, tree_obj, use_f3x4=use_f3x4, optimize_branch_lengths=optimize_branch_lengths,
                                         fix_omega=fix_omega, **init)
    kappa, omega0, omega2, p0, p1, lnl = opt.optimize(**optimizer_kwargs)
    name = "Branch-Site Model A" + (" (null, ω₂=1)" if fix_omega else "")
    return BranchSiteModelResult(model_name=name, lnL=lnl, kappa=kappa, params={"omega0": omega0, "omega2": omega2, "p0": p0, "p1": p1},
                                 tree=tree_obj, alignment=aln, n_params=5 + (opt.n_branches if optimize_branch_lengths else 1),
                                 convergence_info=_info(opt))


def positive_selection(alignment, tree, test: str = "both", verbose: bool = False, **kw):
    """analysis/positive_selection.py:463-end: one of the three tests by name, or M1a-M2a and M7-M8 together."""
    key = test.lower().replace("_", "").replace("-", "")
    if key == "both":
        return {"M1a_vs_M2a": m1a_vs_m2a(alignment, tree, **kw), "M7_vs_M8": m7_vs_m8(alignment, tree, **kw)}
    tests = {"m1avsm2a": m1a_vs_m2a, "m1avm2a": m1a_vs_m2a, "m7vsm8": m7_vs_m8, "m7vm8": m7_vs_m8, "m8avsm8": m8a_vs_m8, "m8avm8": m8a_vs_m8}
    if key not in tests:
        raise ValueError(f"Unknown test: '{test}'. Must be 'M1a_vs_M2a', 'M7_vs_M8', 'M8a_vs_M8', or 'both'.")
    return tests[key](alignment, tree, **kw)
