"""CPU-side checks of everything in the product that is not a kernel: the C-ABI library loads and exports
what include/crabml_amd.h declares, the host eigensolver, the tree -> device-program compiler, pattern
compression, and the Python input producers (io, models, adapter) against the reference's fixtures.
No compute entry point is called here (there is no GPU in this tier)."""
import os
import re

import numpy as np
import pytest

from util import DATA, LYSOZYME_MODELS, emulate_program, leaf_ids, load_case, tree_structure

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------
# the boundary
# ---------------------------------------------------------------------------------------------
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "crabml_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crab_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from crabml_amd import _lib
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/crabml_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes prototypes and header disagree"


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device every compute entry point must fail with CRAB_ERR_DEVICE; with one, create succeeds."""
    from crabml_amd import _lib
    from crabml_amd.engine import Engine, device_available
    if device_available():
        Engine(0).close()
        return
    with pytest.raises(_lib.CrabError) as ei:
        Engine(0)
    assert ei.value.status == _lib.CRAB_ERR_DEVICE and "no CPU fallback" in str(ei.value)
    from crabml_amd import crabml_rust as shim
    c = load_case("lysozyme_m0")
    with pytest.raises(ValueError, match="no CPU fallback"):
        shim.compute_log_likelihood(c["Q"][0], c["pi"], tree_structure(c["parent"]), list(c["branch_lengths"]), None,
                                    c["sequences"], leaf_ids(c))
    with pytest.raises(ValueError, match="no CPU fallback"):
        shim.matrix_exponential(c["Q"][0], c["pi"], 0.1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "crabml_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("independent of oracle/", ""), f"{f} mentions the oracle"


# ---------------------------------------------------------------------------------------------
# host eigensolver (product) against the oracle's
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lysozyme_m0", "lysozyme_m7", "lysozymeLarge_modelA"])
def test_spectral_decomposition_reconstructs_q_and_p(oracle, name):
    from crabml_amd.engine import spectral_decompose
    c = load_case(name)
    pi = c["pi"]
    for Q in c["Q"]:
        U, lam, Ui = spectral_decompose(Q, pi)
        scale = np.abs(Q).max()
        np.testing.assert_allclose((U * lam) @ Ui, Q, atol=3e-14 * scale)
        np.testing.assert_allclose(U @ Ui, np.eye(61), atol=1e-13)
        assert abs(lam.max()) < 1e-13 * scale and (lam <= 1e-13 * scale).all()
        for t in (0.0, 0.03, 1.5):
            np.testing.assert_allclose((U * np.exp(lam * t)) @ Ui, oracle.matrix_exponential(Q, pi, t), atol=2e-14)


def test_spectral_decomposition_rejects_bad_pi():
    from crabml_amd import _lib
    from crabml_amd.engine import spectral_decompose
    c = load_case("lysozyme_m0")
    pi = c["pi"].copy()
    pi[3] = 0.0
    with pytest.raises(_lib.CrabError) as ei:
        spectral_decompose(c["Q"][0], pi)
    assert ei.value.status == _lib.CRAB_ERR_EIGEN


# ---------------------------------------------------------------------------------------------
# tree -> program
# ---------------------------------------------------------------------------------------------
def _p_of_node(oracle, Q, pi, bl):
    # the oracle's own P(t), entry for entry: a different summation order inside U diag(e^{lam t}) U^-1 alone
    # moves ill-conditioned sites by 1e-10 relative, which would mask what this test is about (the program)
    return [oracle.matrix_exponential(Q, pi, t) for t in bl]


@pytest.mark.parametrize("name", ["lysozyme_m0", "hiv_m8a", "lysin_m2a_gaps", "synthetic_edge", "lysozymeLarge_modelA"])
def test_compiled_program_computes_the_pruning_recursion(oracle, name):
    """Interpreting the op list in numpy must give the oracle's per-site likelihoods: validates op order,
    PUSH/POP slots, leaf rows and FIN placement for bi- and multifurcating trees with missing data."""
    from crabml_amd.engine import compile_tree
    c = load_case(name)
    parent, seq = c["parent"], c["sequences"]
    ops, depth = compile_tree(parent, leaf_ids(c))
    Q = c["Q"][0]
    P = _p_of_node(oracle, Q, c["pi"], c["branch_lengths"])
    f = emulate_program(ops, P, seq, c["pi"])
    _, site = oracle.compute_log_likelihood(Q, c["pi"], tree_structure(parent), c["branch_lengths"], None, seq, leaf_ids(c),
                                            return_sites=True)
    np.testing.assert_allclose(np.log(f), site, rtol=1e-12, atol=1e-13)
    n_leaves = seq.shape[0]
    kinds = ops[:, 0]
    assert (kinds == 1).sum() == n_leaves                       # every leaf edge is a gather
    assert (kinds == 0).sum() == len(parent) - 1 - n_leaves     # every internal edge is one MFMA product
    assert ops[-1, 2] & 8                                       # the root finishes the program
    assert depth <= max(1, int(np.ceil(np.log2(n_leaves))))     # Sethi-Ullman bound


def test_program_of_random_trees_and_stack_depth(oracle):
    from crabml_amd import synth
    from crabml_amd.engine import compile_tree
    for seed, n_taxa in [(1, 5), (2, 17), (3, 64), (4, 200)]:
        case = synth.make_case("M0", n_taxa, 40, seed=seed, mean_branch=0.05)
        ops, depth = compile_tree(case.parent, case.leaf_node_ids)
        assert depth <= int(np.ceil(np.log2(n_taxa)))
        P = _p_of_node(oracle, case.Q[0], case.pi, case.branch_lengths)
        f = emulate_program(ops, P, case.codes, case.pi)
        _, site = oracle.compute_log_likelihood(case.Q[0], case.pi, case.tree_structure, case.branch_lengths, None, case.codes,
                                                case.leaf_node_ids, return_sites=True)
        np.testing.assert_allclose(np.log(f), site, rtol=1e-11)
        # prefetch chain: every GEMM names the index of the next GEMM op
        gemm_idx = np.flatnonzero(ops[:, 0] == 0)
        assert list(ops[gemm_idx[:-1], 3]) == list(gemm_idx[1:]) and ops[gemm_idx[-1], 3] == -1


def test_caterpillar_and_star_trees():
    from crabml_amd.engine import compile_tree
    n = 50
    # caterpillar: ((((0,1),2),3)...) -- stack depth 0, no pushes
    parent = np.full(2 * n - 1, -1, dtype=np.int32)
    nxt = n
    cur = 0
    for leaf in range(1, n):
        parent[cur] = parent[leaf] = nxt
        cur = nxt
        nxt += 1
    ops, depth = compile_tree(parent, np.arange(n))
    assert depth == 0 and not (ops[:, 2] & 4).any()
    # star: one root, n leaves -- rescale flags appear every 4th term
    parent = np.full(n + 1, n, dtype=np.int32)
    parent[n] = -1
    ops, depth = compile_tree(parent, np.arange(n))
    assert depth == 0 and len(ops) == n and ((ops[:, 2] & 16) != 0).sum() == n // 4 + 1  # every 4th term, and the root


def test_tree_errors_match_reference_behaviour():
    from crabml_amd import _lib
    from crabml_amd.engine import compile_tree
    with pytest.raises(_lib.CrabError, match="No root node found") as ei:   # likelihood.rs:57-59
        compile_tree([1, 2, 0], [0])
    assert ei.value.status == _lib.CRAB_ERR_NO_ROOT
    with pytest.raises(_lib.CrabError):
        compile_tree([5, -1], [0])           # parent id out of range
    with pytest.raises(_lib.CrabError):
        compile_tree([-1, 0, 0], [0, 1, 9])  # leaf id out of range


def test_children_keep_order_of_appearance_with_explicit_node_ids(oracle):
    """tree_structure may list nodes in any order (likelihood.rs:49-54 keeps children in list order)."""
    from crabml_amd.engine import compile_tree
    c = load_case("synthetic_edge")
    parent = c["parent"]
    perm = np.random.default_rng(0).permutation(len(parent))
    ops_a, _ = compile_tree(parent, leaf_ids(c))
    ops_b, _ = compile_tree(parent[perm], leaf_ids(c), node_ids=perm)
    P = _p_of_node(oracle, c["Q"][0], c["pi"], c["branch_lengths"])
    fa = emulate_program(ops_a, P, c["sequences"], c["pi"])
    fb = emulate_program(ops_b, P, c["sequences"], c["pi"])
    np.testing.assert_allclose(fa, fb, rtol=1e-13)


# ---------------------------------------------------------------------------------------------
# pattern compression
# ---------------------------------------------------------------------------------------------
def test_pattern_compression():
    from crabml_amd.engine import compress_patterns
    c = load_case("lysozymeLarge_modelA")
    s2p, w, first = compress_patterns(c["sequences"])
    assert len(w) == 86 and w.sum() == 130          # PAML's own pattern count for this alignment (lnf header)
    np.testing.assert_array_equal(c["sequences"][:, first][:, s2p], c["sequences"])
    # patterns are numbered in shard order: by (consensus codon of the column, leaves that differ from it, first appearance)
    seq = c["sequences"]
    keys = []
    for p in range(len(w)):
        col = seq[:, first[p]]
        cnt = np.bincount(col[(col >= 0) & (col < 61)], minlength=61)
        cons = int(cnt.argmax()) if cnt.max() > 0 else 61
        keys.append((cons, int(len(col) - (cnt[cons] if cons < 61 else np.sum((col < 0) | (col >= 61)))), int(first[p])))
    assert keys == sorted(keys)
    assert all(first[p] == np.flatnonzero(s2p == p)[0] for p in range(len(w)))   # and represented by their first column
    # missing-data codes are one symbol
    e = load_case("synthetic_edge")
    seq = e["sequences"].copy()
    s2p, w, first = compress_patterns(seq)
    assert s2p[20] == s2p[21] == s2p[1]
    seq2 = seq.copy()
    seq2[seq2 == 64] = -1
    s2p2, w2, _ = compress_patterns(seq2)
    np.testing.assert_array_equal(s2p, s2p2)
    # empty alignment
    s2p, w, first = compress_patterns(np.zeros((4, 0), dtype=np.int32))
    assert len(w) == 0


# ---------------------------------------------------------------------------------------------
# input producers against the reference's fixtures
# ---------------------------------------------------------------------------------------------
def _lysozyme():
    from crabml_amd.io import Alignment
    return Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"))


@pytest.mark.parametrize("model", LYSOZYME_MODELS)
def test_models_reproduce_reference_q_matrices(model):
    import json
    from crabml_amd import models
    c = load_case(f"lysozyme_{model}")
    aln = _lysozyme()
    pi = models.compute_codon_frequencies_f3x4(aln)
    np.testing.assert_allclose(pi, c["pi"], rtol=1e-14)
    kw = json.loads(str(c["params_json"]))
    cls = {"m0": "M0CodonModel", "m1a": "M1aCodonModel", "m2a": "M2aCodonModel", "m3": "M3CodonModel", "m4": "M4CodonModel",
           "m5": "M5CodonModel", "m6": "M6CodonModel", "m7": "M7CodonModel", "m8": "M8CodonModel", "m8a": "M8aCodonModel",
           "m9": "M9CodonModel"}[model]
    m = getattr(models, cls)(pi=pi, **kw)
    if model == "m0":
        Q, props = [m.get_Q_matrix()], [1.0]
    else:
        Q = m.get_Q_matrices()
        props, omegas = m.get_site_classes()
        np.testing.assert_allclose(omegas, c["omegas"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(props, c["proportions"], rtol=1e-14)
    np.testing.assert_allclose(np.stack(Q), c["Q"], rtol=1e-12, atol=1e-18)


def test_adapter_reproduces_reference_boundary_inputs():
    """Tree renumbering and sequence packing (core/likelihood_rust.py:98-158) without touching the device."""
    from crabml_amd.io import Alignment, Tree
    from crabml_amd.likelihood_rust import RustLikelihoodCalculator
    cases = [("lysozyme_m8", _lysozyme()),
             ("hiv_m8a", Alignment.from_fasta(os.path.join(DATA, "HIVenvSweden.fasta"))),
             ("lysin_m2a_gaps", Alignment.from_fasta(os.path.join(DATA, "lysin.fasta"))),
             ("lysozymeLarge_modelA", Alignment.from_phylip(os.path.join(DATA, "lysozymeLarge_clean.nuc")))]
    for name, aln in cases:
        c = load_case(name)
        tree = Tree.from_newick(str(c["newick"]))
        calc = object.__new__(RustLikelihoodCalculator)
        calc.alignment, calc.tree = aln, tree
        calc.leaf_to_seq_idx = {n: i for i, n in enumerate(aln.names)}
        calc._prepare_tree_structure()
        calc._prepare_sequences()
        assert [-1 if p is None else p for _, p in calc.tree_structure] == list(c["parent"]), name
        np.testing.assert_allclose(calc.branch_lengths_template, c["branch_lengths"], rtol=0, atol=0)
        np.testing.assert_array_equal(calc.sequences, c["sequences"])
        assert calc.leaf_names_ordered == [str(x) for x in c["leaf_names"]]
        assert calc.sequences.dtype == np.int32 and calc.sequences.flags.c_contiguous
        if name == "lysozymeLarge_modelA":
            np.testing.assert_array_equal(calc.renumber_branch_labels(), c["branch_labels"])


def test_model_a_inputs_reproduce_reference():
    from crabml_amd import models
    c = load_case("lysozymeLarge_modelA")
    ma = models.BranchSiteModelA(c["pi"], c["branch_labels"].astype(int))
    np.testing.assert_allclose(ma.compute_site_class_frequencies(float(c["p0"]), float(c["p1"])), c["site_class_freqs"], rtol=1e-14)
    qb, qf = ma.compute_qfactors(float(c["kappa"]), float(c["p0"]), float(c["p1"]), float(c["omega0"]), float(c["omega2"]))
    np.testing.assert_allclose([qb, qf], c["qfactors"], rtol=1e-13)
    np.testing.assert_allclose(np.stack(ma.unnormalised_Qs(float(c["kappa"]), float(c["omega0"]), float(c["omega2"]))), c["Q"],
                               rtol=1e-13, atol=1e-18)
    with pytest.raises(ValueError):
        ma.compute_site_class_frequencies(0.6, 0.5)


def test_newick_reader():
    from crabml_amd.io import Tree
    t = Tree.from_newick(open(os.path.join(DATA, "lysozymeLarge.trees")).read())
    assert t.n_leaves == 19 and t.n_nodes == 34 and sum(t.branch_labels()) == 1
    assert max(len(ch) for ch in t.children) == 3  # internal multifurcation + trifurcating root
    t2 = Tree.from_newick(t.to_newick())
    np.testing.assert_array_equal(t.parent, t2.parent)
    np.testing.assert_allclose(t.branch_length, t2.branch_length)
    with pytest.raises(ValueError):
        Tree.from_newick("((A,B),C)")
    t3 = Tree.from_newick("((1,2) #1, ((3,4), 5), (6,7) );        / * table 1C&G */")
    assert t3.n_leaves == 7 and t3.branch_labels().count(1) == 1


def test_codon_encoding():
    from crabml_amd.io import CODONS, encode_codon_string
    assert len(CODONS) == 61 and CODONS[0] == "TTT" and CODONS[-1] == "GGG" and "TAA" not in CODONS
    np.testing.assert_array_equal(encode_codon_string("TTTTAA---GGGNNNATG"), [0, -1, 64, 60, -1, CODONS.index("ATG")])


def test_c_and_python_q_builders_agree():
    from crabml_amd import crabml_rust as shim
    from crabml_amd.models import build_codon_Q_matrix
    rng = np.random.default_rng(5)
    pi = rng.dirichlet(np.ones(61))
    for kappa, omega, nf in [(2.0, 0.5, None), (4.3, 1.7, 1.0), (1.0, 0.0, None), (3.0, 2.0, 0.37)]:
        np.testing.assert_allclose(shim.build_codon_q_matrix_py(kappa, omega, pi, nf), build_codon_Q_matrix(kappa, omega, pi, nf),
                                   rtol=1e-13, atol=1e-17)
    with pytest.raises(ValueError):
        shim.build_codon_q_matrix_py(2.0, 0.5, pi[:60])


# ---------------------------------------------------------------------------------------------
# register plan of the pruning kernels
# ---------------------------------------------------------------------------------------------
def test_compiler_stays_out_of_the_fixed_tile_registers():
    """The pruning kernels keep their conditional-likelihood tiles in fixed VGPRs above an amdgpu_num_vgpr limit
    that the allocator treats as a request, not a guarantee: check the generated ISA."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_reserved_vgprs", os.path.join(ROOT, "scripts", "check_reserved_vgprs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.violations() == []
    # the diagnostic builds re-roll the allocation: the round-2 -DCRAB_PROF build overran its budget into the A-operand ring
    # and faulted on the GPU (scripts/gpu_prof_phases.sh)
    assert mod.violations(("-DCRAB_PROF",)) == []
    assert mod.violations(("-DCRAB_CLOCK",)) == []


def test_spectral_decomposition_of_every_fixture_matrix():
    """Includes M7/M8a classes whose omega is ~1e-100 (columns of the symmetrised matrix 50 orders of magnitude apart)."""
    from crabml_amd.engine import spectral_decompose
    from util import SITE_MODEL_CASES
    for name in list(SITE_MODEL_CASES) + ["lysozymeLarge_modelA"]:
        c = load_case(name)
        for Q in c["Q"]:
            U, lam, Ui = spectral_decompose(Q, c["pi"])
            assert np.abs((U * lam) @ Ui - Q).max() <= 1e-13 * np.abs(Q).max(), name


# ---------------------------------------------------------------------------------------------
# Q construction behind the boundary: the host side of crab_eval_codon_*
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [f"lysozyme_{m}" for m in LYSOZYME_MODELS] + ["hiv_m8a", "lysin_m2a_gaps", "synthetic_edge"])
def test_site_class_rate_matrices_built_in_c_match_the_reference_fixtures(name):
    """The fixtures hold the Q stacks the reference's Python model classes produced (tests/golden/make_golden.py)
    next to the (kappa, omegas, proportions, pi) they were built from."""
    from crabml_amd.engine import build_site_class_q_matrices
    c = load_case(name)
    Q = build_site_class_q_matrices(float(c["kappa"]), c["omegas"], c["proportions"], c["pi"])
    assert Q.shape == c["Q"].shape
    scale = np.abs(c["Q"]).max()
    assert np.abs(Q - c["Q"]).max() <= 4e-15 * scale      # summation order only (the reference's 3.7e-16 story, codon.py:14-19)
    # rows sum to zero, the mixture's mean rate is one
    assert np.abs(Q.sum(axis=2)).max() < 1e-13
    rate = sum(p * -(c["pi"] * np.diagonal(q)).sum() for p, q in zip(c["proportions"], Q))
    assert abs(rate - 1.0) < 1e-13


def test_model_a_quantities_built_in_c_match_the_reference_fixture():
    from crabml_amd.engine import model_a_quantities
    c = load_case("lysozymeLarge_modelA")
    fr, qf = model_a_quantities(c["kappa"], c["omega0"], c["omega2"], c["p0"], c["p1"], c["pi"])
    np.testing.assert_allclose(fr, c["site_class_freqs"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(qf, c["qfactors"], rtol=1e-14, atol=0)
    with pytest.raises(ValueError, match="must be < 1"):          # codon_branch_site.py:139-142
        model_a_quantities(2.0, 0.1, 2.0, 0.7, 0.3, c["pi"])
    with pytest.raises(ValueError, match="too small"):            # :143-146
        model_a_quantities(2.0, 0.1, 2.0, 0.0, 0.0, c["pi"])
    with pytest.raises(ValueError, match="kappa"):
        model_a_quantities(-1.0, 0.1, 2.0, 0.5, 0.3, c["pi"])
    from crabml_amd.engine import build_site_class_q_matrices
    with pytest.raises(ValueError, match="omega"):
        build_site_class_q_matrices(2.0, [0.1, np.nan], [0.5, 0.5], c["pi"])


@pytest.mark.parametrize("name", ["lysozyme_m5", "lysozyme_m7", "lysozyme_m8", "lysozyme_m8a", "hiv_m8a"])
def test_quantile_discretisation_in_c_matches_the_reference_fixtures(name):
    """The fixtures' class omegas came out of the reference's scipy ppf calls (models/codon.py:597-604, 694-701,
    795-802, 892-899) for the parameters in params_json; lysozyme_m7 / m8a sit at the 0.005 floor of the shape
    parameters, where quantiles run from 1e-263 to 1 - 1e-16."""
    import json
    from crabml_amd.engine import discretise_omega
    c = load_case(name)
    par = json.loads(str(c["params_json"]))
    K = par["ncatG"]
    got = discretise_omega("gamma", par["alpha"], par["beta"], K) if name.endswith("m5") else discretise_omega("beta", par["p_beta"], par["q_beta"], K)
    ref = c["omegas"][:K]
    # omega enters Q as omega * pi_j next to entries of order pi_j: what has to agree is omega on the scale of one
    assert np.abs(got - ref).max() <= 1e-13
    big = ref > 1e-200
    assert np.abs(got[big] / ref[big] - 1.0).max() <= 2e-11   # and relatively wherever omega is not a denormal-scale zero


def test_quantile_functions_against_scipy_over_the_optimisers_range():
    from scipy.stats import beta, gamma
    from crabml_amd.engine import discretise_omega
    worst = 0.0
    for a in (0.005, 0.05, 0.5, 0.7, 1.0, 1.5, 2.0, 9.5, 30.0, 99.0):
        for b in (0.005, 0.2, 0.5, 1.0, 1.5, 5.0, 13.5, 30.0, 99.0):
            for K in (1, 4, 10):
                p = (2.0 * np.arange(K) + 1.0) / (2.0 * K)
                for kind, ref in (("beta", beta.ppf(p, a, b)), ("gamma", gamma.ppf(p, a, scale=1.0 / b))):
                    got = discretise_omega(kind, a, b, K)
                    assert np.all(np.diff(got) >= 0) and np.all(got >= 0)
                    ok = ref > 1e-290
                    worst = max(worst, np.abs(got[ok] / ref[ok] - 1.0).max() if ok.any() else 0.0)
                    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, ref.max())
    assert worst <= 5e-11
    # the reference's floors (models/codon.py:570-571, 864-865)
    assert np.array_equal(discretise_omega("beta", 1e-9, -3.0, 5), discretise_omega("beta", 0.005, 0.005, 5))
    with pytest.raises(ValueError):
        discretise_omega("beta", np.nan, 1.0, 5)


@pytest.mark.parametrize("name,kind,keys", [("lysozyme_m6", "M6", ("p0", "alpha1", "beta1", "alpha2")),
                                            ("lysozyme_m9", "M9", ("p0", "p_beta", "q_beta", "alpha", "beta_gamma"))])
def test_mixture_quantiles_in_c_match_the_reference_fixtures(name, kind, keys):
    """M6 / M9 discretise a mixture CDF with scipy.optimize.brentq (models/codon.py:982-1020, 1183-1222): the fixtures' class
    omegas are the reference's own output; the C restatement of Brent's iteration lands on the same digits (2e-11 relative is
    the bar the beta / gamma quantiles meet; brentq's own xtol is 2e-12 absolute)."""
    import json
    from crabml_amd.engine import discretise_omega_mixture
    c = load_case(name)
    par = json.loads(str(c["params_json"]))
    got = discretise_omega_mixture(kind, [par[k] for k in keys], par["ncatG"])
    ref = c["omegas"][: par["ncatG"]]
    assert np.abs(got / ref - 1.0).max() <= 2e-11
    assert np.all(np.diff(got) > 0)


def test_mixture_quantiles_against_the_scipy_route_over_the_optimisers_range():
    """Random parameters inside the optimisers' bounds: against crabml_amd.models' scipy route (the reference's expression:
    brentq on p0 F1 + (1 - p0) F2 - p over [1e-4, 99], fallback 10 p).  Both routes stop within brentq's xtol = 2e-12 of the
    root; where a branch of the iteration flips on the last bits of a CDF value they stop at different points inside it."""
    from crabml_amd import models
    from crabml_amd.engine import discretise_omega_mixture
    rng = np.random.default_rng(11)
    n_fallback = 0
    for _ in range(60):
        p0 = rng.uniform(0.001, 0.999)
        a = np.exp(rng.uniform(np.log(0.005), np.log(99.0), 4))
        for kind, model, params in (("M6", models.M6CodonModel(2.0, p0, a[0], a[1], a[2]), [p0, a[0], a[1], a[2]]),
                                    ("M9", models.M9CodonModel(2.0, p0, a[0], a[1], a[2], a[3]), [p0, a[0], a[1], a[2], a[3]])):
            ref = np.array(model.get_site_classes()[1])
            got = discretise_omega_mixture(kind, params, 10)
            p = (2.0 * np.arange(10) + 1.0) / 20.0
            n_fallback += int(np.sum(ref == p * 10))
            assert np.array_equal(got == p * 10, ref == p * 10)          # the reference's fallback, on the same classes
            assert np.abs(got - ref).max() <= 4e-12 + 1e-13 * ref.max()
    assert n_fallback > 0                                               # (steep mixtures put the first quantiles below 1e-4)
    # the constructors' clamps (models/codon.py:966-971, 1168-1172), and bad arguments
    assert np.array_equal(discretise_omega_mixture("M6", [2.0, 1e-9, 1.0, 1.0], 5), discretise_omega_mixture("M6", [1 - 1e-6, 0.005, 1.0, 1.0], 5))
    with pytest.raises(ValueError):
        discretise_omega_mixture("M9", [0.5, 1.0, 1.0, 1.0], 5)         # M9 takes five parameters
    with pytest.raises(ValueError):
        discretise_omega_mixture("M6", [0.5, np.nan, 1.0, 1.0], 5)


def test_host_worker_pool_serves_concurrent_callers():
    """map_genes drives several engines from several host threads: their parallel loops share one pool."""
    from crabml_amd import _lib
    lib = _lib.load()
    for callers, n, rounds in ((1, 11, 50), (4, 11, 200), (16, 63, 100), (8, 1998, 5), (3, 2, 300)):
        _lib.check(lib.crab_host_pool_selftest(callers, n, rounds))


def test_tree_rejects_a_repeated_node_id():
    """tree_structure lists every node exactly once; a repeated id would leave another node without an entry."""
    from crabml_amd.engine import compile_tree
    with pytest.raises(ValueError, match="more than once"):
        compile_tree([2, 2, -1], [0, 1], node_ids=[0, 0, 2])
    ops, _ = compile_tree([2, 2, -1], [0, 1], node_ids=[1, 0, 2])
    assert len(ops) == 2


def test_shim_checks_the_number_of_leaf_node_ids():
    """The C library reads one node id per alignment row: a shorter list must be refused before it gets there."""
    from crabml_amd import crabml_rust as shim
    c = load_case("lysozyme_m0")
    ts = tree_structure(c["parent"])
    with pytest.raises(ValueError, match="leaf_node_ids has 6 entries but sequences has 7 rows"):
        shim.compute_log_likelihood(c["Q"][0], c["pi"], ts, c["branch_lengths"].tolist(), [], c["sequences"], list(range(6)))


def test_path_programs_recompute_exactly_what_a_moved_branch_changes():
    """Partial updates: for every branch of random trees (polytomies, unary nodes, leaves on the root included), the
    path program run on the tiles a base walk saved must give what the whole-tree program gives when that branch's
    transition matrix is replaced -- the same products in the same order, hence the same bits."""
    from crabml_amd.engine import compile_path_program
    rng = np.random.default_rng(7)
    n = 7
    for trial in range(12):
        n_leaves = int(rng.integers(2, 14))
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1 or nxt == n_leaves:
            k = int(min(len(open_nodes), rng.choice([1, 2, 2, 2, 3, 4])))
            if len(open_nodes) == 1 and nxt > n_leaves and k == 1:
                break
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        n_nodes = nxt
        par = np.array([parent.get(v, -1) for v in range(n_nodes)], dtype=np.int32)
        leaf = np.arange(n_leaves, dtype=np.int32)
        S = 9
        codes = rng.integers(0, n, size=(n_leaves, S)).astype(np.int32)
        codes[rng.random(codes.shape) < 0.1] = n
        pi = rng.dirichlet(np.ones(n))
        P = {v: rng.dirichlet(np.ones(n), size=n) for v in range(n_nodes)}
        whole, n_slots = compile_path_program(par, leaf)
        saved = {}
        base = emulate_program(whole, P, codes, pi, n_states=n, saved=saved, save=True)
        assert np.array_equal(base, emulate_program(whole, P, codes, pi, n_states=n))
        n_gemm = int((whole[:, 0] == 0).sum())
        assert n_slots == 3 * n_gemm
        for v in range(n_nodes):
            if par[v] < 0:
                continue
            path, _ = compile_path_program(par, leaf, node=v)
            P2 = dict(P)
            P2[v] = rng.dirichlet(np.ones(n), size=n)
            want = emulate_program(whole, P2, codes, pi, n_states=n)
            got = emulate_program(path, P2, codes, pi, n_states=n, saved=saved)
            assert np.array_equal(got, want), f"trial {trial}, node {v}"
            assert len(path) <= len(whole) + 1
            assert not (path[:, 2] & 4).any()                       # no stack traffic on a path
    with pytest.raises(ValueError, match="no branch above"):
        compile_path_program([2, 2, -1], [0, 1], node=2)


def test_tree_parts_compute_the_whole_tree_walk():
    """A tree cut into subtree programs plus a root program (the finer work quantum of single evaluations): run through
    the numpy emulator, parts first, the pieces must reproduce the whole-tree program bit for bit, for every part size."""
    from crabml_amd.engine import compile_path_program, compile_tree_parts
    rng = np.random.default_rng(17)
    n = 5
    seen_cut = False
    for trial in range(10):
        n_leaves = int(rng.integers(4, 60))
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1:
            k = int(min(len(open_nodes), rng.choice([1, 2, 2, 2, 3])))
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        par = np.array([parent.get(v, -1) for v in range(nxt)], dtype=np.int32)
        leaf = np.arange(n_leaves, dtype=np.int32)
        codes = rng.integers(0, n + 1, size=(n_leaves, 6)).astype(np.int32)
        pi = rng.dirichlet(np.ones(n))
        P = {v: rng.dirichlet(np.ones(n), size=n) for v in range(nxt)}
        whole, _ = compile_path_program(par, leaf)
        want = emulate_program(whole, P, codes, pi, n_states=n)
        n_gemm = int((whole[:, 0] == 0).sum())
        for target in (2, 4, 9, 1000):
            root, parts = compile_tree_parts(par, leaf, target)
            saved = {}
            for part in parts:
                assert (part[-1, 0], part[-1, 2] & 1024) == (3, 1024)          # ends by saving its tile
                assert 2 <= int((part[:, 0] == 0).sum()) <= target
                emulate_part = emulate_program(part[:-1], P, codes, pi, n_states=n)   # a well-formed program of its own
                assert np.isfinite(emulate_part).all()
                emulate_program(part, P, codes, pi, n_states=n, saved=saved)
            assert sorted(saved) == list(range(len(parts)))
            got = emulate_program(root, P, codes, pi, n_states=n, saved=saved)
            assert np.array_equal(got, want), f"trial {trial}, target {target}"
            assert int((root[:, 0] == 0).sum()) + sum(int((p[:, 0] == 0).sum()) for p in parts) == n_gemm
            seen_cut |= len(parts) > 1
    assert seen_cut


def test_clade_tables_compute_the_whole_tree_walk():
    """Clade tables (subtree site repeats): clades walked once per distinct sub-pattern, the main program gathering their
    tables, reproduce the whole-tree program on every pattern (numpy emulator); the plan is self-consistent (every pattern's
    sub-pattern carries its codes, products add up, clades are disjoint and small)."""
    from crabml_amd.engine import compile_clade_tables, compile_path_program
    rng = np.random.default_rng(23)
    n = 5
    used = seen_nested = 0
    for trial in range(8):
        n_leaves = int(rng.integers(6, 40))
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1:
            k = int(min(len(open_nodes), rng.choice([1, 2, 2, 2, 3])))
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        par = np.array([parent.get(v, -1) for v in range(nxt)], dtype=np.int32)
        leaf = np.arange(n_leaves, dtype=np.int32)
        n_pat = int(rng.integers(300, 700))
        base = rng.integers(0, n + 1, size=n_pat)                      # related sequences: sub-patterns repeat (code n = missing)
        codes = np.stack([np.where(rng.random(n_pat) < 0.15, rng.integers(0, n + 1, size=n_pat), base) for _ in range(n_leaves)]).astype(np.int32)
        pi = rng.dirichlet(np.ones(n))
        P = {v: rng.dirichlet(np.ones(n), size=n) for v in range(nxt)}
        whole, _ = compile_path_program(par, leaf)
        want = emulate_program(whole, P, codes, pi, n_states=n)
        main, clades = compile_clade_tables(par, leaf, codes, n_states=n)
        if not clades:
            assert np.array_equal(main, whole) or len(main) == len(whole)
            continue
        used += 1
        tables, nested = {}, 0
        n_sub = [c["sub_codes"].shape[1] for c in clades]
        canon = np.where((codes >= 0) & (codes < n), codes, n)
        for j, c in enumerate(clades):
            ops = c["ops"]
            assert (ops[-1, 0], ops[-1, 2] & (1024 | 2048)) == (3, 1024 | 2048) and ops[-2, 0] == 0 and ops[-2, 1] == c["root"]
            assert c["parent"] == -1 or c["parent"] > j                  # inner clades come first
            sub, U = c["sub_codes"], n_sub[j]
            n_consumer = n_pat if c["parent"] < 0 else n_sub[c["parent"]]
            index = c["index"][:n_consumer].astype(np.int64)
            assert 1 <= U <= n_pat and index.max() == U - 1 and len(np.unique(index)) == U
            rows = [int(r) for r in np.unique(ops[(ops[:, 0] == 1) & (ops[:, 3] < n_leaves), 3])]
            consumer_codes = canon if c["parent"] < 0 else clades[c["parent"]]["sub_codes"]
            assert np.array_equal(sub[rows][:, index], consumer_codes[rows])   # a consumer's pattern carries the clade's codes
            inner = {n_leaves + d["row"]: d["index"][:U].astype(np.int64) for d in clades if d["parent"] == j}
            nested += bool(inner)
            tables[c["root"]] = emulate_program(ops, P, sub, pi, n_states=n, tables=tables, wide=inner)
            assert tables[c["root"]].shape == (U, n)
        outer = [c for c in clades if c["parent"] < 0]
        wide = {n_leaves + c["row"]: c["index"].astype(np.int64) for c in outer}
        table_leaves = main[(main[:, 0] == 1) & (main[:, 3] >= n_leaves)]
        assert sorted(table_leaves[:, 3]) == sorted(wide) and sorted(table_leaves[:, 1]) == sorted(c["root"] for c in outer)
        n_gemm = lambda ops: int((ops[:, 0] == 0).sum())
        assert n_gemm(main) + sum(n_gemm(c["ops"]) for c in clades) == n_gemm(whole)
        got = emulate_program(main, P, codes, pi, n_states=n, tables=tables, wide=wide)
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)
        seen_nested += nested
    assert used >= 4 and seen_nested >= 1
