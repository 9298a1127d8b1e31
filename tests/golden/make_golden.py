#!/usr/bin/env python3
"""Generate golden input/output vectors for the likelihood hot path.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It imports the reference's *Python* modules (the pure-Python
LikelihoodCalculator is the reference's own oracle for M0 and site-class
mixtures; the Rust crate cannot be built here) and writes small ``.npz`` files
holding inputs exactly as they cross the crabml_rust boundary plus the expected
outputs.  The ``.npz`` files are data (arrays only, loadable with
``allow_pickle=False``); nothing from the reference's sources is stored.

    python tests/golden/make_golden.py            # all cases (~10 min, 8 cores)
    python tests/golden/make_golden.py m0 m8      # a subset

Expected-output provenance, per case:
  * ``lnL_pyref``            reference Python calculator, whole alignment
  * ``site_class_lnf_pyref`` ln f_{k,s} from the reference Python calculator run on
                             single-column alignments (one per unique pattern)
  * ``lnL_paml``             the PAML value the reference's own tests assert
  * lnf tables               PAML per-pattern ``lnf`` files the reference ships
"""
import json
import os
import sys
import types
from concurrent.futures import ProcessPoolExecutor

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")


def _import_reference():
    pkg = types.ModuleType("crabml")
    pkg.__path__ = [f"{REF}/src/crabml"]
    sys.modules["crabml"] = pkg
    import crabml.io.sequences as seqs
    import crabml.io.trees as trees
    import crabml.models.codon as codon
    import crabml.models.codon_branch_site as cbs
    import crabml.models.codon_branch as cb
    import crabml.core.likelihood as lik
    import crabml.core.likelihood_rust as lrust
    return seqs, trees, codon, cbs, cb, lik, lrust


seqs, trees, codon, cbs, cb, lik, lrust = _import_reference()

# --- KAT parameter sets: the values the reference's tests feed in
#     (tests/test_rust/test_paml_rust_validation.py) ---------------------------
LYS = {
    "m0": dict(tree="((Hsa_Human: 0.025561, Hla_gibbon: 0.038887): 0.067982, ((Cgu/Can_colobus: 0.043792, Pne_langur: 0.052539): 0.076369, Mmu_rhesus: 0.021684): 0.043448, (Ssc_squirrelM: 0.040804, Cja_marmoset: 0.023918): 0.122664);",
               model=("M0CodonModel", dict(kappa=4.54008, omega=0.80663)), paml=-906.017440),
    "m1a": dict(tree="((Hsa_Human: 0.025558, Hla_gibbon: 0.039375): 0.069610, ((Cgu/Can_colobus: 0.044140, Pne_langur: 0.052277): 0.077775, Mmu_rhesus: 0.021343): 0.044215, (Ssc_squirrelM: 0.041574, Cja_marmoset: 0.023673): 0.125738);",
                model=("M1aCodonModel", dict(kappa=4.29790, omega0=0.000001, p0=0.41271)), paml=-902.503872),
    "m2a": dict(tree="((Hsa_Human: 0.025240, Hla_gibbon: 0.041269): 0.077423, ((Cgu/Can_colobus: 0.044409, Pne_langur: 0.053758): 0.085025, Mmu_rhesus: 0.019175): 0.046022, (Ssc_squirrelM: 0.042962, Cja_marmoset: 0.024797): 0.136928);",
                model=("M2aCodonModel", dict(kappa=5.03734, omega0=0.35983, omega2=4.52341, p0=0.77376, p1=0.10399)), paml=-899.998568),
    "m3": dict(tree="((Hsa_Human: 0.025496, Hla_gibbon: 0.041612): 0.078581, ((Cgu/Can_colobus: 0.044623, Pne_langur: 0.054035): 0.085125, Mmu_rhesus: 0.019785): 0.045122, (Ssc_squirrelM: 0.043139, Cja_marmoset: 0.024888): 0.138453);",
               model=("M3CodonModel", dict(kappa=5.07876, omegas=[0.37564, 3.35232, 8.39917], proportions=[0.83729, 0.14475, 0.01796])), paml=-899.985262),
    "m4": dict(tree="((Hsa_Human: 0.025116, Hla_gibbon: 0.040656): 0.074120, ((Cgu/Can_colobus: 0.044275, Pne_langur: 0.052863): 0.082198, Mmu_rhesus: 0.019989): 0.045550, (Ssc_squirrelM: 0.042315, Cja_marmoset: 0.024402): 0.132002);",
               model=("M4CodonModel", dict(kappa=4.82959, proportions=[0.04251, 0.75935, 0.00000, 0.00000, 0.19814])), paml=-900.139530),
    "m5": dict(tree="((Hsa_Human: 0.025431, Hla_gibbon: 0.040846): 0.075035, ((Cgu/Can_colobus: 0.044631, Pne_langur: 0.053006): 0.082704, Mmu_rhesus: 0.020012): 0.045472, (Ssc_squirrelM: 0.042442, Cja_marmoset: 0.024538): 0.133407);",
               model=("M5CodonModel", dict(kappa=4.91082, alpha=0.43434, beta=0.44697, ncatG=10)), paml=-900.222360),
    "m6": dict(tree="((Hsa_Human: 0.025319, Hla_gibbon: 0.040604): 0.074894, ((Cgu/Can_colobus: 0.044277, Pne_langur: 0.052821): 0.082456, Mmu_rhesus: 0.019800): 0.045217, (Ssc_squirrelM: 0.042294, Cja_marmoset: 0.024371): 0.133164);",
               model=("M6CodonModel", dict(kappa=4.88043, p0=0.20472, alpha1=8.31960, beta1=10.86728, alpha2=0.31083, ncatG=10)), paml=-900.167515),
    "m7": dict(tree="((Hsa_Human: 0.025610, Hla_gibbon: 0.039441): 0.069693, ((Cgu/Can_colobus: 0.044220, Pne_langur: 0.052394): 0.077888, Mmu_rhesus: 0.021399): 0.044279, (Ssc_squirrelM: 0.041624, Cja_marmoset: 0.023739): 0.125873);",
               model=("M7CodonModel", dict(kappa=4.314921, p_beta=0.007506, q_beta=0.005000, ncatG=10)), paml=-902.510018),
    "m8": dict(tree="((Hsa_Human: 0.025174, Hla_gibbon: 0.041255): 0.077239, ((Cgu/Can_colobus: 0.044374, Pne_langur: 0.053701): 0.084936, Mmu_rhesus: 0.019172): 0.046065, (Ssc_squirrelM: 0.042928, Cja_marmoset: 0.024782): 0.136634);",
               model=("M8CodonModel", dict(kappa=5.029752, p0=0.868486, p_beta=9.517504, q_beta=13.543157, omega_s=4.370617, ncatG=10)), paml=-899.999237),
    "m8a": dict(tree="((Hsa_Human: 0.025558, Hla_gibbon: 0.039375): 0.069610, ((Cgu/Can_colobus: 0.044140, Pne_langur: 0.052277): 0.077775, Mmu_rhesus: 0.021343): 0.044215, (Ssc_squirrelM: 0.041574, Cja_marmoset: 0.023673): 0.125738);",
                model=("M8aCodonModel", dict(kappa=4.297905, p0=0.41271, p_beta=0.005, q_beta=58.524404, ncatG=10)), paml=-902.503869),
    "m9": dict(tree="((Hsa_Human: 0.025366, Hla_gibbon: 0.041324): 0.077781, ((Cgu/Can_colobus: 0.044497, Pne_langur: 0.053862): 0.085140, Mmu_rhesus: 0.019255): 0.045871, (Ssc_squirrelM: 0.043034, Cja_marmoset: 0.024828): 0.137479);",
               model=("M9CodonModel", dict(kappa=5.05056, p0=0.84895, p_beta=16.31508, q_beta=28.11235, alpha=9.04750, beta_gamma=2.05077, ncatG=10)), paml=-899.997270),
}


def boundary_inputs(aln, tree):
    """Run the reference adapter's tree renumbering / sequence packing
    (core/likelihood_rust.py:98-158) without the native extension."""
    calc = object.__new__(lrust.RustLikelihoodCalculator)
    calc.alignment, calc.tree = aln, tree
    calc.n_states, calc.n_sites = 61, aln.n_sites
    calc.leaf_to_seq_idx = {n: i for i, n in enumerate(aln.names)}
    calc._prepare_tree_structure()
    calc._prepare_sequences()
    parent = np.array([-1 if p is None else p for _, p in calc.tree_structure], dtype=np.int32)
    assert [i for i, _ in calc.tree_structure] == list(range(len(parent)))
    return calc, parent


def sub_alignment(aln, cols, missing_to_minus1=True):
    s = aln.sequences[:, cols].copy()
    if missing_to_minus1:  # the Python oracle indexes with any code >= 0; gap code 64 would overflow
        s = np.where((s >= 0) & (s < 61), s, -1).astype(s.dtype)
    return seqs.Alignment(names=aln.names, sequences=s, n_species=aln.n_species, n_sites=len(cols), seqtype="codon")


def unique_patterns(sequences):
    cols, first, inverse = np.unique(sequences.T, axis=0, return_index=True, return_inverse=True)
    return first, inverse.reshape(-1)


def _one_pattern(args):
    aln_names, column, newick, Qs, pi = args
    aln = seqs.Alignment(names=aln_names, sequences=column.reshape(-1, 1), n_species=len(aln_names), n_sites=1, seqtype="codon")
    tree = trees.Tree.from_newick(newick)
    c = lik.LikelihoodCalculator(aln, tree)
    return [float(c.compute_log_likelihood(Q, pi)) for Q in Qs]


def site_class_lnf(aln, newick, Qs, pi, pool):
    """ln(f_{k,s} + 1e-100) per site and class via the reference M0 path on single columns."""
    clean = sub_alignment(aln, np.arange(aln.n_sites))
    first, inverse = unique_patterns(clean.sequences)
    jobs = [(aln.names, clean.sequences[:, c].copy(), newick, Qs, pi) for c in first]
    res = np.array(list(pool.map(_one_pattern, jobs, chunksize=4)))  # [n_pat, K]
    return res[inverse]  # [S, K]


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}", flush=True)


def make_site_model_case(name, aln, newick, Qs, props, pi, pool, paml=None, whole=True, extra=None):
    tree = trees.Tree.from_newick(newick)
    calc, parent = boundary_inputs(aln, tree)
    Qs = [np.asarray(Q) for Q in Qs]
    out = dict(pi=pi, Q=np.stack(Qs), proportions=np.asarray(props, dtype=np.float64), parent=parent,
               branch_lengths=np.asarray(calc.branch_lengths_template, dtype=np.float64),
               sequences=calc.sequences.astype(np.int32), leaf_names=np.array(calc.leaf_names_ordered))
    pyaln = sub_alignment(aln, np.arange(aln.n_sites))
    if whole:
        pc = lik.LikelihoodCalculator(pyaln, tree)
        if len(Qs) == 1 and props is None:
            out["lnL_pyref"] = float(pc.compute_log_likelihood(Qs[0], pi))
        else:
            out["lnL_pyref"] = float(pc.compute_log_likelihood_site_classes(Qs, pi, list(props)))
    # per-site, per-class values need the leaf order of the *boundary* sequences: columns are the same sites
    out["site_class_lnf_pyref"] = site_class_lnf(aln, newick, Qs, pi, pool)
    if paml is not None:
        out["lnL_paml"] = float(paml)
    if extra:
        out.update(extra)
    save(name, **out)


def lysozyme_cases(which, pool):
    aln = seqs.Alignment.from_phylip(f"{DATA}/lysozymeSmall.txt", seqtype="codon")
    pi = codon.compute_codon_frequencies_f3x4(aln)
    for key, spec in LYS.items():
        if which and key not in which:
            continue
        cls, kw = spec["model"]
        model = getattr(codon, cls)(pi=pi, **kw)
        if key == "m0":
            Qs, props = [model.get_Q_matrix()], None
            omegas = [kw["omega"]]
        else:
            Qs = model.get_Q_matrices()
            props, omegas = model.get_site_classes()
        extra = dict(omegas=np.asarray(omegas, dtype=np.float64), kappa=float(kw["kappa"]),
                     params_json=np.array(json.dumps(kw)), newick=np.array(spec["tree"]))
        make_site_model_case(f"lysozyme_{key}", aln, spec["tree"], Qs,
                             props if props is not None else [1.0], pi, pool, paml=spec["paml"], extra=extra)


HIV_TREE = ("(U68496: 0.024500, U68497: 0.086487, ((U68498: 0.052048, (U68499: 0.029598, U68500: 0.107658): 0.092830): 0.024846, "
            "(U68501: 0.144215, (U68502: 0.218116, (U68503: 0.221305, ((((U68504: 0.019878, U68506: 0.073917): 0.058486, "
            "U68505: 0.032344): 0.040849, U68507: 0.063213): 0.022662, U68508: 0.232978): 0.060823): 0.066725): 0.026058): 0.052033): 0.170282);")


def hiv_case(pool):
    """HIV env M8a at PAML's MLEs (tests/data/paml_reference/hiv_nssites/mlc_m8a:206-230)."""
    aln = seqs.Alignment.from_fasta(f"{DATA}/HIVenvSweden.fasta", seqtype="codon")
    pi = codon.compute_codon_frequencies_f3x4(aln)
    kw = dict(kappa=2.589539, p0=0.502730, p_beta=1.106068, q_beta=10.214396, ncatG=10)
    model = codon.M8aCodonModel(pi=pi, **kw)
    Qs = model.get_Q_matrices()
    props, omegas = model.get_site_classes()
    make_site_model_case("hiv_m8a", aln, HIV_TREE, Qs, props, pi, pool, paml=-1114.579213,
                         extra=dict(omegas=np.asarray(omegas), kappa=kw["kappa"], params_json=np.array(json.dumps(kw)),
                                    newick=np.array(HIV_TREE)))


def lysin_case(pool):
    """25 taxa x 135 codons with gap codons (code 64); arbitrary but fixed parameters."""
    aln = seqs.Alignment.from_fasta(f"{DATA}/lysin.fasta", seqtype="codon")
    pi = codon.compute_codon_frequencies_f3x4(aln)
    txt = open(f"{DATA}/lysin.trees").read()
    tree = trees.Tree.from_newick(txt[txt.index("("):])  # drop the bare "1" header line
    rng = np.random.default_rng(7)
    for node in tree.postorder():
        if node.parent is not None:
            node.branch_length = float(np.round(rng.uniform(0.01, 0.4), 6))
    newick = tree.to_newick()
    model = codon.M2aCodonModel(kappa=1.8, omega0=0.1, omega2=3.5, p0=0.6, p1=0.25, pi=pi)
    Qs = model.get_Q_matrices()
    props, omegas = model.get_site_classes()
    make_site_model_case("lysin_m2a_gaps", aln, newick, Qs, props, pi, pool,
                         extra=dict(omegas=np.asarray(omegas), kappa=1.8, newick=np.array(newick)))


def synthetic_edge_cases(pool):
    """Tiny trees exercising: missing data (-1 / 64), a multifurcating internal node, a
    four-way root, a zero-length branch, duplicated columns, and a zero class proportion."""
    rng = np.random.default_rng(11)
    newick = "((A:0.1,B:0.2,C:0.05):0.3,(D:0.0,E:0.4):0.15,F:0.25,(G:0.2,(H:0.1,I:0.3):0.0):0.1);"
    names = list("ABCDEFGHI")
    S = 24
    codes = rng.integers(0, 61, size=(9, S)).astype(np.int8)
    # make sites correlated so likelihoods are not vanishingly small
    base = rng.integers(0, 61, size=S)
    mask = rng.random((9, S)) < 0.6
    codes[mask] = np.broadcast_to(base, (9, S))[mask]
    codes[0, 3] = -1
    codes[4, 3] = 64
    codes[:, 5] = 64            # an all-missing column: f = 1
    codes[2, 7] = -1
    codes[:, 20] = codes[:, 1]  # duplicated columns
    codes[:, 21] = codes[:, 1]
    aln = seqs.Alignment(names=names, sequences=codes, n_species=9, n_sites=S, seqtype="codon")
    pi = rng.dirichlet(5 * np.ones(61))
    model = codon.M3CodonModel(kappa=2.3, omegas=[0.05, 0.9, 2.5, 6.0], proportions=[0.5, 0.3, 0.0, 0.2], pi=pi)
    Qs = model.get_Q_matrices()
    props, omegas = model.get_site_classes()
    make_site_model_case("synthetic_edge", aln, newick, Qs, props, pi, pool,
                         extra=dict(omegas=np.asarray(omegas), kappa=2.3, newick=np.array(newick)))


def parse_lnf(path):
    rows = []
    for ln in open(path):
        p = ln.split()
        if len(p) >= 5 and p[0].isdigit() and p[1].isdigit():
            try:
                rows.append((int(p[0]), int(p[1]), float(p[2])))
            except ValueError:
                pass
    return np.array(rows)


def model_a_case():
    """Branch-site Model A at PAML's MLEs (tests/test_branch_site_likelihood.py:30-128).
    No Python oracle exists for this path; expected values are PAML's lnL and the
    per-pattern lnf file (tests/data/paml_examples/lysozyme/lnf)."""
    aln = seqs.Alignment.from_phylip(f"{DATA}/lysozymeLarge_clean.nuc", seqtype="codon")
    tree = trees.Tree.from_newick(open(f"{DATA}/lysozymeLarge.trees").read())
    paml_bl = [0.042333, 0.075622, 0.026519, 0.008018, 0.007767, 0.000004, 0.023339,
               0.007530, 0.037779, 0.023589, 0.000004, 0.000004, 0.007956, 0.000004,
               0.007935, 0.016006, 0.008104, 0.016154, 0.007906, 0.016186, 0.008059,
               0.069643, 0.006258, 0.008004, 0.015965, 0.000004, 0.000004, 0.034475,
               0.135219, 0.027063, 0.000004, 0.032959, 0.040867]
    branches = tree.get_branches()
    for (par, child), t in zip(branches, paml_bl):
        child.branch_length = t
    kappa, p0, p1, omega0, omega2 = 4.154927, 0.326513, 0.269308, 0.000001, 4.809765
    pi = codon.compute_codon_frequencies_f3x4(aln)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = cbs.BranchSiteModelA(pi, tree.get_branch_labels())
    freqs = model.compute_site_class_frequencies(p0, p1)
    qb, qf = model.compute_qfactors(kappa, p0, p1, omega0, omega2)
    Q0 = codon.build_codon_Q_matrix(kappa=kappa, omega=omega0, pi=pi, normalization_factor=1.0)
    Q1 = codon.build_codon_Q_matrix(kappa=kappa, omega=1.0, pi=pi, normalization_factor=1.0)
    Q2 = codon.build_codon_Q_matrix(kappa=kappa, omega=omega2, pi=pi, normalization_factor=1.0)
    calc, parent = boundary_inputs(aln, tree)
    labels = [0] * (len(parent) - 1)
    orig = tree.get_branch_labels()
    for i, (par, child) in enumerate(branches):
        rid = calc.node_id_map[child.id]
        if rid < len(labels):
            labels[rid] = orig[i]
    lnf = parse_lnf(f"{DATA}/lysozymeLarge_modelA.lnf")
    save("lysozymeLarge_modelA", pi=pi, Q=np.stack([Q0, Q1, Q2]), qfactors=np.array([qb, qf]),
         omega0=omega0, omega2=omega2, kappa=kappa, p0=p0, p1=p1, site_class_freqs=freqs, parent=parent,
         branch_lengths=np.asarray(calc.branch_lengths_template), branch_labels=np.array(labels, dtype=np.uint8),
         sequences=calc.sequences.astype(np.int32), leaf_names=np.array(calc.leaf_names_ordered),
         lnL_paml=-1035.533916, paml_lnf=lnf, newick=np.array(tree.to_newick()),
         branch_lengths_preorder=np.array(paml_bl))


def main():
    which = set(sys.argv[1:])
    with ProcessPoolExecutor(max_workers=8) as pool:
        if not which or which & set(LYS):
            lysozyme_cases(which & set(LYS), pool)
        if not which or "hiv" in which:
            hiv_case(pool)
        if not which or "lysin" in which:
            lysin_case(pool)
        if not which or "edge" in which:
            synthetic_edge_cases(pool)
        if not which or "modela" in which:
            model_a_case()


if __name__ == "__main__":
    main()
