"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Needs an MI355X: run with -m gpu.

Tolerances are those of BASELINE.json's north_star: per-pattern log-likelihood within 1e-10 relative,
total lnL within 1e-6 absolute.  Where a comparison is made in a different way the test says why.

Two kinds of comparison are made for every case:
  end-to-end   Q, pi, t in -> lnL / per-site lnL out, against the oracle fed the same inputs;
  staged       the transition matrices kernel 1 built are exported and (a) compared with the oracle's P(t)
               entry by entry (absolute: the eigen route is accurate to ~2e-15 absolute per entry whatever
               the arithmetic order, see tests/test_oracle.py), (b) fed to the oracle's pruning, which must
               then reproduce kernel 2 + 3's per-site result to 1e-10 relative on *every* site, including
               the ill-conditioned ones where two eigensolvers legitimately differ by more than that.

The end-to-end per-site gate is |gpu - oracle| <= 1e-10 |oracle| + b_s on EVERY site, where b_s is the rigorous
per-site bound of tests/util.py:eigen_route_bound on how far two evaluations can be apart whose P(t) entries agree to
the accuracy every eigen route has (1e-14 absolute; pinned against 60-digit arithmetic by tests/golden/parity_tail.npz).
b_s is ~5e-13 on ordinary sites, so there the gate is the plain 1e-10; it opens only on sites that lean on entries of
P(t) that are themselves 1e-8 ... 1e-12.  test_parity_tail_sites_against_60_digit_truth closes the loop on those sites:
the device is at least as close to the truth there as the reference's own LAPACK route.
"""
import os

import numpy as np
import pytest

from util import (DATA, GOLDEN, SITE_MODEL_CASES, SITE_RTOL, TOTAL_ATOL, assert_by_class_close, check_work, eigen_route_bound, leaf_ids,
                  load_case, tree_structure)

pytestmark = pytest.mark.gpu

WELL_CONDITIONED = [c for c in SITE_MODEL_CASES if c not in ("lysin_m2a_gaps", "synthetic_edge")]
LN_TINY = np.log(1e-300)


def _engine(Engine, parent, codes, leaf=None, **kw):
    eng = Engine(0)
    eng.set_tree(parent, np.arange(codes.shape[0]) if leaf is None else leaf)
    eng.set_alignment(codes, n_states=61, **kw)
    return eng


def _mixture(lnf_by_class, props, clamp=True):
    x = np.maximum(lnf_by_class, LN_TINY) if clamp else lnf_by_class
    with np.errstate(divide="ignore"):
        x = x + np.log(props)[None, :]
    mx = x.max(axis=1)
    return mx + np.log(np.exp(x - mx[:, None]).sum(axis=1))


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))


# ---------------------------------------------------------------------------------------------
# fixtures from the reference: site-class models
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SITE_MODEL_CASES)
def test_site_class_models_against_oracle_and_python_reference(gpu_engine_cls, oracle, name):
    c = load_case(name)
    ts, leaf = tree_structure(c["parent"]), leaf_ids(c)
    Q, props, pi, bl, seq = c["Q"], c["proportions"], c["pi"], c["branch_lengths"], c["sequences"]
    with _engine(gpu_engine_cls, c["parent"], seq) as eng:
        lnL, site = eng.site_classes(Q, props, pi, bl, return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(len(Q))])
        by_class = eng.site_lnl_by_class(Q, pi, bl)
    ref, ref_site = oracle.compute_site_class_log_likelihood(list(Q), props, pi, ts, bl, None, seq, leaf, return_sites=True)
    # end-to-end: the north_star tolerances on every fixture, the ill-conditioned ones included (module docstring)
    assert abs(lnL - ref) < TOTAL_ATOL
    assert abs(lnL - float(c["lnL_pyref"])) < TOTAL_ATOL          # the reference's own Python calculator
    if "lnL_paml" in c and name != "hiv_m8a":
        assert abs(lnL - float(c["lnL_paml"])) < 1e-5 * abs(lnL) + 1e-5   # the reference's KAT tolerance
    bound = eigen_route_bound(oracle, P_dev, pi, ts, seq, leaf, lambda x: _mixture(x, props))
    assert np.all(np.abs(site - ref_site) <= SITE_RTOL * np.maximum(np.abs(ref_site), 1e-3) + bound)
    if name in WELL_CONDITIONED:
        assert _rel(site, ref_site) <= SITE_RTOL and bound.max() < 1e-8
    assert abs(site.sum() - lnL) < 1e-8
    # staged: kernel 1
    for k in range(len(Q)):
        for v in range(len(bl)):
            if c["parent"][v] >= 0:
                np.testing.assert_allclose(P_dev[k, v], oracle.matrix_exponential(Q[k], pi, bl[v]), rtol=0, atol=2e-14)
    # staged: kernels 2 + 3 on the device's own matrices -- strict on every site
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, pi, ts, seq, leaf)
    assert _rel(site, _mixture(lnf, props)) <= SITE_RTOL
    assert_by_class_close(by_class, np.maximum(lnf, LN_TINY), name)
    assert by_class.shape == (seq.shape[1], len(Q))


def test_m0_known_answer_through_the_adapter(gpu_engine_cls):
    """tests/test_rust/test_paml_rust_validation.py:50-97 with our adapter in place of RustLikelihoodCalculator."""
    from crabml_amd import models
    from crabml_amd.io import Alignment, Tree
    from crabml_amd.likelihood_rust import RustLikelihoodCalculator
    aln = Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"), seqtype="codon")
    pi = models.compute_codon_frequencies_f3x4(aln)
    tree = Tree.from_newick("((Hsa_Human: 0.025561, Hla_gibbon: 0.038887): 0.067982, ((Cgu/Can_colobus: 0.043792, "
                            "Pne_langur: 0.052539): 0.076369, Mmu_rhesus: 0.021684): 0.043448, (Ssc_squirrelM: 0.040804, "
                            "Cja_marmoset: 0.023918): 0.122664);")
    calc = RustLikelihoodCalculator(aln, tree)
    Q = models.M0CodonModel(kappa=4.54008, omega=0.80663, pi=pi).get_Q_matrix()
    lnL = calc.compute_log_likelihood(Q, pi)
    np.testing.assert_allclose(lnL, -906.017440, rtol=1e-6, atol=1e-6)
    assert abs(lnL - (-906.0174402566055)) < 1e-9
    assert calc.compute_log_likelihood(Q, pi) == lnL   # bit-identical repeat (tests/test_integration/test_lysozyme.py:129-136)
    # M8 through the same calculator
    m8 = models.M8CodonModel(kappa=5.029752, p0=0.868486, p_beta=9.517504, q_beta=13.543157, omega_s=4.370617, ncatG=10, pi=pi)
    tree8 = Tree.from_newick("((Hsa_Human: 0.025174, Hla_gibbon: 0.041255): 0.077239, ((Cgu/Can_colobus: 0.044374, "
                             "Pne_langur: 0.053701): 0.084936, Mmu_rhesus: 0.019172): 0.046065, (Ssc_squirrelM: 0.042928, "
                             "Cja_marmoset: 0.024782): 0.136634);")
    calc8 = RustLikelihoodCalculator(aln, tree8)
    lnL8 = calc8.compute_log_likelihood_site_classes(m8.get_Q_matrices(), pi, m8.get_site_classes()[0])
    np.testing.assert_allclose(lnL8, -899.999237, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        calc8.compute_log_likelihood_site_classes(m8.get_Q_matrices(), pi, [0.5, 0.5])
    with pytest.raises(ValueError):
        calc.compute_log_likelihood(Q[:60, :60], pi)
    # scale_branch_lengths is honoured
    a = calc.compute_log_likelihood(Q, pi, scale_branch_lengths=1.3)
    for v in range(tree.n_nodes):
        tree.branch_length[v] *= 1.3
    assert abs(calc.compute_log_likelihood(Q, pi) - a) < 1e-9


# ---------------------------------------------------------------------------------------------
# branch-site Model A and the branch model (Rust-only paths of the reference)
# ---------------------------------------------------------------------------------------------
def test_branch_site_model_a(gpu_engine_cls, oracle):
    c = load_case("lysozymeLarge_modelA")
    ts, leaf = tree_structure(c["parent"]), leaf_ids(c)
    args = (float(c["qfactors"][0]), float(c["qfactors"][1]), float(c["omega0"]), float(c["omega2"]), c["site_class_freqs"])
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        lnL, site = eng.branch_site(c["Q"], *args, c["pi"], c["branch_lengths"], c["branch_labels"], return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(4)])
        # all-background identity
        zero = np.zeros_like(c["branch_labels"])
        lnL0, site0 = eng.branch_site(c["Q"], *args, c["pi"], c["branch_lengths"], zero, return_sites=True)
        f = c["site_class_freqs"]
        qb = args[0]
        two, two_site = eng.site_classes(np.stack([c["Q"][0] * qb, c["Q"][1] * qb]), [f[0] + f[2], f[1] + f[3]], c["pi"],
                                         c["branch_lengths"], return_sites=True)
        with pytest.raises(ValueError, match="Branch labels length"):
            eng.branch_site(c["Q"], *args, c["pi"], c["branch_lengths"], c["branch_labels"][:-1])
        with pytest.raises(ValueError):
            eng.branch_site(c["Q"], *args, c["pi"], c["branch_lengths"], np.full_like(c["branch_labels"], 2))
    ref, ref_site = oracle.compute_branch_site_log_likelihood(c["Q"][0], c["Q"][1], c["Q"][2], *args, c["pi"], ts,
                                                              c["branch_lengths"], c["branch_labels"], None, c["sequences"],
                                                              leaf, return_sites=True)
    assert abs(lnL - float(c["lnL_paml"])) < 1e-5           # tests/test_branch_site_likelihood.py:121-128
    assert abs(lnL - ref) < TOTAL_ATOL
    assert _rel(site, ref_site) <= SITE_RTOL
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, c["pi"], ts, c["sequences"], leaf, naive=True)
    staged = np.log((np.exp(lnf) * c["site_class_freqs"][None, :]).sum(axis=1))
    assert _rel(site, staged) <= SITE_RTOL
    assert abs(lnL0 - two) < 1e-8 and _rel(site0, two_site) <= SITE_RTOL


@pytest.mark.parametrize("n_taxa,n_sites,seed", [(9, 200, 21), (60, 1500, 22)])
def test_branch_model_free_ratio(gpu_engine_cls, oracle, n_taxa, n_sites, seed):
    from crabml_amd import synth
    case = synth.make_case("free-ratio", n_taxa, n_sites, seed=seed)
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        lnL, site = eng.branch(case.Q_per_node, case.pi, case.branch_lengths, return_sites=True)
        P_dev = eng.export_transition_matrices(0)
        with pytest.raises(ValueError, match="must equal number of nodes"):
            eng.branch(case.Q_per_node[:-1], case.pi, case.branch_lengths[:-1])
        with pytest.raises(ValueError, match="must match branch_lengths length"):
            eng.branch(case.Q_per_node, case.pi, case.branch_lengths[:-1])
    ref, ref_site = oracle.compute_log_likelihood_branch(case.codes, case.parent, case.leaf_node_ids, case.Q_per_node,
                                                         case.branch_lengths, case.pi, return_sites=True)
    assert abs(lnL - ref) < TOTAL_ATOL
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, case.pi, case.tree_structure, case.codes, case.leaf_node_ids)
    assert _rel(site, np.maximum(lnf[:, 0], LN_TINY)) <= SITE_RTOL
    bound = eigen_route_bound(oracle, P_dev[None], case.pi, case.tree_structure, case.codes, case.leaf_node_ids,
                              lambda x: np.maximum(x[:, 0], LN_TINY))
    assert np.all(np.abs(site - ref_site) <= SITE_RTOL * np.abs(ref_site) + bound)
    assert np.quantile(bound / np.abs(ref_site), 0.99) < SITE_RTOL      # on 99 % of the sites the gate is the plain one (within 2x)


def test_branch_model_with_shared_omega_is_m0(gpu_engine_cls):
    c = load_case("lysozyme_m0")
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        m0 = eng.log_likelihood(c["Q"][0], c["pi"], c["branch_lengths"])
        br = eng.branch(np.repeat(c["Q"][:1], len(c["parent"]), axis=0), c["pi"], c["branch_lengths"])
    assert m0 == br


# ---------------------------------------------------------------------------------------------
# the seven crabml_rust entry points
# ---------------------------------------------------------------------------------------------
def test_shim_functions_match_the_handle_api(gpu_engine_cls, oracle):
    from crabml_amd import crabml_rust as shim
    c = load_case("lysozyme_m8")
    ts, leaf = tree_structure(c["parent"]), leaf_ids(c)
    bl = c["branch_lengths"].tolist()
    names = [str(x) for x in c["leaf_names"]]
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        want = eng.site_classes(c["Q"], c["proportions"], c["pi"], c["branch_lengths"])
        want_bc = eng.site_lnl_by_class(c["Q"], c["pi"], c["branch_lengths"])
        want_m0 = eng.log_likelihood(c["Q"][3], c["pi"], c["branch_lengths"])
    got = shim.compute_site_class_log_likelihood(q_matrices=list(c["Q"]), proportions=list(c["proportions"]), pi=c["pi"],
                                                 tree_structure=ts, branch_lengths=bl, leaf_names=names,
                                                 sequences=c["sequences"], leaf_node_ids=leaf)
    assert isinstance(got, float) and got == want
    bc = shim.compute_site_log_likelihoods_by_class(q_matrices=list(c["Q"]), pi=c["pi"], tree_structure=ts, branch_lengths=bl,
                                                    leaf_names=names, sequences=c["sequences"], leaf_node_ids=leaf)
    assert bc.shape == (130, 11) and np.array_equal(bc, want_bc)
    m0 = shim.compute_log_likelihood(q=c["Q"][3], pi=c["pi"], tree_structure=ts, branch_lengths=bl, leaf_names=names,
                                     sequences=c["sequences"], leaf_node_ids=leaf)
    assert m0 == want_m0
    # int64 / non-contiguous sequences are accepted (converted), results unchanged
    assert shim.compute_log_likelihood(c["Q"][3], c["pi"], ts, bl, names, c["sequences"].astype(np.int64), leaf) == want_m0
    # branch model entry point
    par = [-1 if p is None else p for _, p in ts]
    Qn = np.repeat(c["Q"][3:4], len(ts), axis=0)
    assert shim.compute_log_likelihood_branch(sequences=c["sequences"], parent_indices=par, leaf_node_ids=leaf, Q_matrices=Qn,
                                              branch_lengths=c["branch_lengths"], pi=c["pi"]) == want_m0
    assert shim.compute_log_likelihood_branch(c["sequences"], par, leaf, list(Qn), bl, c["pi"]) == want_m0
    with pytest.raises(ValueError):
        shim.compute_log_likelihood_branch(c["sequences"], par, leaf, Qn[:-1], bl, c["pi"])
    # branch-site entry point, positional as the reference calls it (optimize/branch_site.py:169-185)
    a = load_case("lysozymeLarge_modelA")
    lnL = shim.compute_branch_site_log_likelihood(a["Q"][0], a["Q"][1], a["Q"][2], float(a["qfactors"][0]),
                                                  float(a["qfactors"][1]), float(a["omega0"]), float(a["omega2"]),
                                                  a["site_class_freqs"].tolist(), a["pi"], tree_structure(a["parent"]),
                                                  a["branch_lengths"].tolist(), [int(x) for x in a["branch_labels"]],
                                                  [str(x) for x in a["leaf_names"]], a["sequences"], leaf_ids(a))
    assert abs(lnL - (-1035.533916)) < 1e-5
    # errors surface as ValueError (lib.rs PyValueError)
    with pytest.raises(ValueError, match="square"):
        shim.compute_log_likelihood(c["Q"][3][:, :60], c["pi"], ts, bl, names, c["sequences"], leaf)
    with pytest.raises(ValueError, match="pi must have length"):
        shim.compute_log_likelihood(c["Q"][3], c["pi"][:60], ts, bl, names, c["sequences"], leaf)
    with pytest.raises(ValueError, match="No root"):
        shim.compute_log_likelihood(c["Q"][3], c["pi"], [(i, (i + 1) % len(ts)) for i in range(len(ts))], bl, names,
                                    c["sequences"], leaf)
    shim.clear_cache()


def test_one_shot_callers_in_parallel_and_cache_turnover(gpu_engine_cls):
    """The handle cache behind the seven stateless functions: threads working on different alignments run side by side
    (the list lock is not held across an evaluation), twelve alignments turn the eight-entry cache over while calls are
    in flight, alignments that differ in one codon or one parent index are different entries, and every value equals
    the one a fresh engine gives."""
    from concurrent.futures import ThreadPoolExecutor
    from crabml_amd import crabml_rust as shim
    c = load_case("lysozyme_m8")
    ts, leaf = tree_structure(c["parent"]), leaf_ids(c)
    bl, names = c["branch_lengths"].tolist(), [str(x) for x in c["leaf_names"]]
    rng = np.random.default_rng(8)
    aligns = []
    for i in range(12):
        seq = c["sequences"].copy()
        if i:
            seq[rng.integers(0, seq.shape[0]), rng.integers(0, seq.shape[1])] = int(rng.integers(0, 61))   # one codon differs
        aligns.append(np.ascontiguousarray(seq))
    want = []
    for seq in aligns:
        with _engine(gpu_engine_cls, c["parent"], seq) as eng:
            want.append(eng.site_classes(c["Q"], c["proportions"], c["pi"], c["branch_lengths"]))
    assert len(set(want)) > 6

    def call(i):
        return shim.compute_site_class_log_likelihood(list(c["Q"]), list(c["proportions"]), c["pi"], ts, bl, names, aligns[i % 12], leaf)

    shim.clear_cache()
    with ThreadPoolExecutor(max_workers=6) as ex:
        got = list(ex.map(call, range(72)))
    assert got == [want[i % 12] for i in range(72)]
    # the same codes on a tree with two leaves exchanged under their parents: another entry, another value
    par = c["parent"].copy()
    kids = [i for i in range(len(par)) if par[i] != par[0] and i < len(leaf)]
    par[0], par[kids[0]] = par[kids[0]], par[0]
    swapped = shim.compute_site_class_log_likelihood(list(c["Q"]), list(c["proportions"]), c["pi"], tree_structure(par), bl, names, aligns[0], leaf)
    with _engine(gpu_engine_cls, par, aligns[0]) as eng:
        assert swapped == eng.site_classes(c["Q"], c["proportions"], c["pi"], c["branch_lengths"])
    assert call(0) == want[0]
    shim.clear_cache()


def test_matrix_exponential_on_device(gpu_engine_cls, oracle):
    from scipy.linalg import expm
    from crabml_amd import crabml_rust as shim
    c = load_case("lysozyme_m2a")
    for Q in c["Q"]:
        for t in (0.0, 1e-4, 0.05, 0.7, 5.0):
            P = shim.matrix_exponential(Q, c["pi"], t)
            np.testing.assert_allclose(P, expm(Q * t), atol=2e-14, rtol=0)
            np.testing.assert_allclose(P, oracle.matrix_exponential(Q, c["pi"], t), atol=2e-14, rtol=0)
    Q = c["Q"][0]
    np.testing.assert_allclose(shim.matrix_exponential(Q, c["pi"], 0.0), np.eye(61), atol=1e-14)     # matrix.rs:121-136
    np.testing.assert_allclose(shim.matrix_exponential(Q, c["pi"], 0.3) @ shim.matrix_exponential(Q, c["pi"], 0.5),
                               shim.matrix_exponential(Q, c["pi"], 0.8), atol=1e-13)                  # semigroup
    P = shim.matrix_exponential(Q, c["pi"], 0.1)
    np.testing.assert_allclose(c["pi"][:, None] * P, (c["pi"][:, None] * P).T, atol=1e-15)            # matrix.rs:139-168
    # JC69, n = 4 (tests/test_reference/test_matrix.py:21-45)
    Qj = np.full((4, 4), 0.25)
    np.fill_diagonal(Qj, -0.75)
    P = shim.matrix_exponential(Qj, np.full(4, 0.25), 0.3)
    np.testing.assert_allclose(np.diag(P), 0.25 + 0.75 * np.exp(-0.3), atol=1e-15)
    with pytest.raises(ValueError, match="square"):
        shim.matrix_exponential(Q[:, :5], c["pi"], 0.1)


# ---------------------------------------------------------------------------------------------
# BASELINE.json configurations at sizes the oracle finishes in seconds, plus size-independent properties
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2_case():
    from crabml_amd import synth
    return synth.make_case("M8", 100, 10000, seed=2)


def test_config2_m8_100x10k_full_size(gpu_engine_cls, oracle, c2_case):
    """BASELINE.json configs[1] at full size: M8, 100 taxa x 10 000 codons, every site against the oracle."""
    case = c2_case
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        lnL, site = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        again = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
        P0 = eng.export_transition_matrices(10)
        work, n_cl = eng.last_work(), eng.last_clade_tables()[0]
        n_pat = eng.n_patterns
    assert again == lnL                                                # run-to-run deterministic
    check_work(work, n_cl, 97 * n_pat * 11, 100 * n_pat * 11)          # N-3 internal edges, N leaf edges
    ref, ref_site = oracle.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, case.tree_structure,
                                                             case.branch_lengths, None, case.codes, case.leaf_node_ids,
                                                             return_sites=True)
    assert abs(lnL - ref) < TOTAL_ATOL
    rel = np.abs(site - ref_site) / np.abs(ref_site)
    # every one of the 10 000 sites inside 1e-10 (with the eigenvectors re-orthogonalised the three eigen routes agree to
    # 3e-12 on this alignment, tests/golden/parity_tail.npz)
    assert rel.max() <= SITE_RTOL
    np.testing.assert_allclose(P0[5], oracle.matrix_exponential(case.Q[10], case.pi, case.branch_lengths[5]), atol=2e-14, rtol=0)


def test_config2_staged_strict_on_a_subsample(gpu_engine_cls, oracle, c2_case):
    case = c2_case
    cols = np.arange(0, 10000, 7)
    codes = np.ascontiguousarray(case.codes[:, cols])
    with _engine(gpu_engine_cls, case.parent, codes) as eng:
        lnL, site = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(11)])
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, case.pi, case.tree_structure, codes, case.leaf_node_ids)
    assert _rel(site, _mixture(lnf, case.proportions)) <= SITE_RTOL


def test_properties_compression_slicing_linearity_padding(gpu_engine_cls, c2_case):
    case = c2_case
    Q, pr, pi, bl = case.Q, case.proportions, case.pi, case.branch_lengths
    codes = case.codes[:, :3001]  # deliberately not a multiple of the tile width
    with _engine(gpu_engine_cls, case.parent, codes) as eng:
        whole, site = eng.site_classes(Q, pr, pi, bl, return_sites=True)
        n_pat = eng.n_patterns_total
    with _engine(gpu_engine_cls, case.parent, codes, compress=False) as eng:
        raw, site_raw = eng.site_classes(Q, pr, pi, bl, return_sites=True)
        assert eng.n_patterns == 3001
    assert abs(whole - raw) < 1e-7 and np.array_equal(site, site_raw)
    # duplicating every column doubles lnL and leaves the pattern count unchanged
    with _engine(gpu_engine_cls, case.parent, np.concatenate([codes, codes], axis=1)) as eng:
        assert eng.n_patterns_total == n_pat
        assert abs(eng.site_classes(Q, pr, pi, bl) - 2 * whole) < 1e-7
    # pattern slices (the multi-GPU shards) add up, and per-site outputs tile the alignment
    for parts in (2, 3):
        total, filled = 0.0, np.full(codes.shape[1], np.nan)
        edges = np.linspace(0, n_pat, parts + 1).astype(int)
        for a, b in zip(edges[:-1], edges[1:]):
            with _engine(gpu_engine_cls, case.parent, codes, pattern_begin=int(a), pattern_end=int(b)) as eng:
                assert eng.n_patterns == b - a
                part, s = eng.site_classes(Q, pr, pi, bl, return_sites=True)
            total += part
            assert not (np.isfinite(filled) & np.isfinite(s)).any()
            filled = np.where(np.isfinite(s), s, filled)
        assert abs(total - whole) < 1e-7 and np.array_equal(filled, site)
    # a permutation of the columns permutes the per-site output
    perm = np.random.default_rng(0).permutation(codes.shape[1])
    with _engine(gpu_engine_cls, case.parent, np.ascontiguousarray(codes[:, perm])) as eng:
        lnLp, sp = eng.site_classes(Q, pr, pi, bl, return_sites=True)
    assert abs(lnLp - whole) < 1e-7 and np.array_equal(sp, site[perm])


def test_long_trees_clamp_semantics_and_rescaling(gpu_engine_cls, oracle):
    """500 taxa with long branches: the reference's unscaled products underflow on most sites and classes
     and it returns ln(1e-300) for them.  With the clamp the engine reproduces exactly those numbers
    (clamped and unclamped classes mixed within one site included); without it, it returns the true
    log-likelihood, which the oracle's rescaled pruning confirms."""
    from crabml_amd import synth
    case = synth.make_case("M8", 500, 256, seed=9, mean_branch=0.6)
    ts = case.tree_structure
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        clamped, site_c = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        exact, site_x = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, no_clamp=True, return_sites=True)
    ref_c, ref_site_c = oracle.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, ts, case.branch_lengths,
                                                                 None, case.codes, case.leaf_node_ids, return_sites=True)
    clamped_sites = ref_site_c < -690.7
    assert clamped_sites.sum() > 50 and (~clamped_sites).sum() > 20   # both regimes are exercised
    np.testing.assert_allclose(site_c, ref_site_c, rtol=SITE_RTOL)
    assert abs(clamped - ref_c) < TOTAL_ATOL
    P = []
    for Qk in case.Q:
        U, lam, Ui = oracle.cached_q(Qk, case.pi)
        P.append([(U * np.exp(lam * t)) @ Ui for t in case.branch_lengths])
    P = np.array(P)
    lnf = oracle.site_lnf_from_transition_matrices(P, case.pi, ts, case.codes, case.leaf_node_ids)
    want = _mixture(lnf, case.proportions, clamp=False)
    assert (want[clamped_sites] < -690.77).all() and want.min() < -1500 and np.isfinite(site_x).all()
    assert _rel(site_x, want) <= 1e-9          # P(t) here comes from numpy's summation order, hence 1e-9
    assert abs(exact - want.sum()) < 1e-5


def test_edge_cases(gpu_engine_cls, oracle):
    c = load_case("synthetic_edge")
    Q, pr, pi, bl = c["Q"], c["proportions"], c["pi"], c["branch_lengths"]
    # empty alignment
    with _engine(gpu_engine_cls, c["parent"], np.zeros((9, 0), dtype=np.int32)) as eng:
        assert eng.site_classes(Q, pr, pi, bl) == 0.0 and eng.site_lnl_by_class(Q, pi, bl).shape == (0, 4)
    # a single column; an all-missing column has likelihood 1
    with _engine(gpu_engine_cls, c["parent"], c["sequences"][:, 5:6]) as eng:
        assert abs(eng.site_classes(Q, pr, pi, bl)) < 1e-13
    # a zero-probability class (omega = 0 forbids the observed change): clamp value, not NaN
    m4 = load_case("lysozyme_m4")
    with _engine(gpu_engine_cls, m4["parent"], m4["sequences"]) as eng:
        bc = eng.site_lnl_by_class(m4["Q"], m4["pi"], m4["branch_lengths"])
        assert np.isfinite(bc).all() and bc.min() >= LN_TINY - 1e-9
        lnL = eng.site_classes(m4["Q"], m4["proportions"], m4["pi"], m4["branch_lengths"])
    assert abs(lnL - (-900.139530)) < 9e-3
    # explicit node ids in shuffled order; leaves not numbered first
    perm = np.random.default_rng(3).permutation(len(c["parent"]))
    inv = np.argsort(perm)
    new_parent = np.array([-1 if c["parent"][perm[i]] < 0 else inv[c["parent"][perm[i]]] for i in range(len(perm))], dtype=np.int32)
    leaf_nodes = inv[np.arange(9)]
    with gpu_engine_cls(0) as eng:
        eng.set_tree(new_parent, leaf_nodes)
        eng.set_alignment(c["sequences"])
        got = eng.site_classes(Q, pr, pi, bl[perm])
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        want = eng.site_classes(Q, pr, pi, bl)
        # state errors
        with pytest.raises(ValueError, match="pi must have length"):
            eng.site_classes(Q, pr, pi[:60], bl)
        with pytest.raises(ValueError):
            eng.site_classes(Q, pr[:3], pi, bl)
    assert abs(got - want) < 1e-9
    with gpu_engine_cls(0) as eng:
        from crabml_amd import _lib
        with pytest.raises(_lib.CrabError) as ei:
            eng.n_states, eng.n_nodes, eng.n_sites = 61, len(bl), 1
            eng.site_classes(Q, pr, pi, bl)
        assert ei.value.status == _lib.CRAB_ERR_STATE


def test_deep_stack_register_and_global_slots(gpu_engine_cls, oracle):
    """A perfectly balanced 64-leaf tree needs 5 saved sibling products at once (Sethi-Ullman depth 5): the first
    slots live in registers, the rest spill to the global stack.  Both must give the oracle's per-site values."""
    from crabml_amd import synth
    from crabml_amd.engine import compile_tree
    base = synth.make_case("M8", 8, 257, seed=31, mean_branch=0.07)
    n = 64
    parent = np.full(2 * n - 1, -1, dtype=np.int32)
    level, nxt = list(range(n)), n
    while len(level) > 1:
        up = []
        for i in range(0, len(level), 2):
            parent[level[i]] = parent[level[i + 1]] = nxt
            up.append(nxt)
            nxt += 1
        level = up
    _, depth = compile_tree(parent, np.arange(n, dtype=np.int32))
    assert depth == 5
    rng = np.random.default_rng(5)
    bl = rng.exponential(0.05, size=len(parent))
    bl[-1] = 0.0
    codes = rng.integers(0, 61, size=(n, 257)).astype(np.int32)
    codes[rng.random(codes.shape) < 0.02] = 64
    ts = [(v, None if parent[v] < 0 else int(parent[v])) for v in range(len(parent))]
    with _engine(gpu_engine_cls, parent, codes) as eng:
        lnL, site = eng.site_classes(base.Q, base.proportions, base.pi, bl, return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(len(base.Q))])
    # random codes put the weight on near-impossible paths, where P(t)'s tiny entries differ between eigensolvers by
    # far more than 1e-10 relative: staged comparison (the device's own P(t) through the oracle's pruning), strict
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, base.pi, ts, codes, np.arange(n, dtype=np.int32))
    want = _mixture(lnf, base.proportions)
    assert _rel(site, want) <= SITE_RTOL and abs(lnL - want.sum()) <= TOTAL_ATOL
    ref, ref_site = oracle.compute_site_class_log_likelihood(list(base.Q), base.proportions, base.pi, ts, bl, None, codes,
                                                             np.arange(n, dtype=np.int32), return_sites=True)
    # end to end: within the per-site eigen-route bound (large here, for the reason above)
    bound = eigen_route_bound(oracle, P_dev, base.pi, ts, codes, np.arange(n, dtype=np.int32), lambda x: _mixture(x, base.proportions))
    assert np.all(np.abs(site - ref_site) <= SITE_RTOL * np.abs(ref_site) + bound)


def test_batched_parameter_vectors_equal_single_evaluations(gpu_engine_cls):
    """crab_eval_site_classes_batch: a finite-difference stencil (branch lengths and kappa/omega perturbed one at a
    time) in one launch sequence gives, element for element, the bits of the single evaluations."""
    from crabml_amd import models, synth
    case = synth.make_case("M8", 20, 700, seed=41)
    B = 7
    Q = np.repeat(case.Q[None], B, axis=0).copy()
    pr = np.repeat(np.asarray(case.proportions)[None], B, axis=0).copy()
    bl = np.repeat(case.branch_lengths[None], B, axis=0).copy()
    for b in range(1, 5):
        bl[b, 3 * b] *= 1.0 + 1e-6            # branch-length perturbations: no new rate matrix
    m = models.M8CodonModel(kappa=case.kappa * (1 + 1e-6), p0=0.9, p_beta=0.7, q_beta=1.5, omega_s=2.5, ncatG=10, pi=case.pi)
    Q[5] = np.stack(m.get_Q_matrices())
    pr[6] = pr[6][::-1] / pr[6].sum()         # a different mixture on the same matrices
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        got = eng.site_classes_batch(Q, pr, case.pi, bl)
        want = np.array([eng.site_classes(Q[b], pr[b], case.pi, bl[b]) for b in range(B)])
        again = eng.site_classes_batch(Q, pr, case.pi, bl)
    assert np.array_equal(got, want) and np.array_equal(got, again)
    assert len(set(got.tolist())) >= 6        # the perturbations are visible


# ---------------------------------------------------------------------------------------------
# evaluations from model parameters: rate matrices built behind the boundary, batched (8f.1)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SITE_MODEL_CASES)
def test_site_class_models_from_parameters(gpu_engine_cls, name):
    """(kappa, omegas, proportions) in, lnL out: against the same engine fed the reference's own Q stack, and
    against the reference's Python lnL.  The C-built matrices differ from the reference's in the last bit
    (summation order), so this is a tolerance comparison, not a bit comparison."""
    c = load_case(name)
    ka, om, pr, pi, bl = float(c["kappa"]), c["omegas"], c["proportions"], c["pi"], c["branch_lengths"]
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        from_q = eng.site_classes(c["Q"], pr, pi, bl)
        from_params = eng.codon_site_classes(ka, om, pr, pi, bl)
        # a batch: the vector itself, a kappa perturbation, a branch-length perturbation
        B = 3
        kas = np.array([ka, ka * (1 + 1e-6), ka])
        bls = np.repeat(bl[None], B, axis=0)
        bls[2, 0] *= 1.0 + 1e-6
        batch = eng.codon_site_classes(kas, np.repeat(om[None], B, 0), np.repeat(pr[None], B, 0), pi, bls)
        singles = [eng.codon_site_classes(kas[b], om, pr, pi, bls[b]) for b in range(B)]
    assert abs(from_params - from_q) < (1e-9 * max(1.0, abs(from_q)) if name in WELL_CONDITIONED else TOTAL_ATOL)
    assert abs(from_params - float(c["lnL_pyref"])) < TOTAL_ATOL
    assert batch[0] == from_params
    assert list(batch) == singles                              # bit-identical to one call per vector
    assert batch[1] != batch[0] and batch[2] != batch[0]       # the perturbations were seen


def test_branch_site_model_a_from_parameters(gpu_engine_cls):
    c = load_case("lysozymeLarge_modelA")
    pars = [float(c[k]) for k in ("kappa", "omega0", "omega2", "p0", "p1")]
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        from_q = eng.branch_site(c["Q"], float(c["qfactors"][0]), float(c["qfactors"][1]), float(c["omega0"]), float(c["omega2"]),
                                 c["site_class_freqs"], c["pi"], c["branch_lengths"], c["branch_labels"])
        from_params = eng.codon_branch_site(*pars, c["pi"], c["branch_lengths"], c["branch_labels"])
        # batch of 4: base, omega2 -> 1 (the null model: classes 2a/2b on the foreground use Q(1)), p0 and t perturbed
        B = 4
        P = np.repeat(np.array(pars)[None], B, axis=0)
        P[1, 2] = 1.0
        P[2, 3] *= 1.0 + 1e-6
        bls = np.repeat(c["branch_lengths"][None], B, axis=0)
        bls[3, 1] *= 1.0 + 1e-6
        batch = eng.codon_branch_site(P[:, 0], P[:, 1], P[:, 2], P[:, 3], P[:, 4], c["pi"], bls, c["branch_labels"])
        singles = [eng.codon_branch_site(*P[b], c["pi"], bls[b], c["branch_labels"]) for b in range(B)]
        with pytest.raises(ValueError, match="must be < 1"):
            eng.codon_branch_site(2.0, 0.1, 2.0, 0.7, 0.4, c["pi"], c["branch_lengths"], c["branch_labels"])
        with pytest.raises(ValueError, match="Branch labels length"):
            eng.codon_branch_site(*pars, c["pi"], c["branch_lengths"], c["branch_labels"][:-1])
    assert abs(from_params - from_q) < 1e-9
    assert abs(from_params - float(c["lnL_paml"])) < 1e-5      # tests/test_branch_site_likelihood.py:121-128
    assert batch[0] == from_params
    assert list(batch) == singles
    assert len(set(batch.tolist())) == B
    assert batch[1] < batch[0]                                 # the null model cannot beat the alternative at its optimum


def test_branch_model_from_parameters(gpu_engine_cls):
    from crabml_amd import synth
    case = synth.make_case("free-ratio", 40, 800, seed=23)
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        from_q = eng.branch(case.Q_per_node, case.pi, case.branch_lengths)
        from_params = eng.codon_branch(case.kappa, case.omega_per_node, case.pi, case.branch_lengths)
        B = 3
        om = np.repeat(case.omega_per_node[None], B, axis=0)
        om[1, 5] *= 1.0 + 1e-6
        om[2, :] = 0.3                                         # one shared omega: the M0 model
        batch = eng.codon_branch(np.full(B, case.kappa), om, case.pi, np.repeat(case.branch_lengths[None], B, 0))
        singles = [eng.codon_branch(case.kappa, om[b], case.pi, case.branch_lengths) for b in range(B)]
        m0 = eng.codon_site_classes(case.kappa, [0.3], [1.0], case.pi, case.branch_lengths)
    assert abs(from_params - from_q) < 1e-9 * abs(from_q)
    assert batch[0] == from_params
    assert list(batch) == singles
    assert batch[2] == m0                                      # tests/test_branch_models.py:131-174, exactly


def test_eigensystem_pool_growth_and_eviction(gpu_engine_cls):
    """The device-resident pool of decompositions grows (contents must survive the move) and, past 4096 entries,
    recycles its least recently used slots: results must not depend on what the pool has seen before."""
    c = load_case("lysozyme_m3")
    pi, bl = c["pi"], c["branch_lengths"]
    rng = np.random.default_rng(5)

    def fresh(ka, om, pr):
        with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as e2:
            return e2.codon_site_classes(ka, om, pr, pi, bl)

    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        first = eng.codon_site_classes(2.0, [0.1, 0.9, 2.0], [0.5, 0.3, 0.2], pi, bl)           # 3 entries, pool of 32
        K = 60
        om = np.sort(rng.uniform(0.01, 3.0, K))
        pr = np.full(K, 1.0 / K)
        grown = eng.codon_site_classes(2.0, np.concatenate([[0.1, 0.9, 2.0], om]), np.concatenate([[0.2, 0.2, 0.2], 0.4 * pr]), pi, bl)
        assert eng.codon_site_classes(2.0, [0.1, 0.9, 2.0], [0.5, 0.3, 0.2], pi, bl) == first    # hits, after the pool moved
        # content-keyed entries share the pool
        q_lnL = eng.site_classes(c["Q"], c["proportions"], pi, bl)
        # push more than 4096 distinct matrices through, 64 per call
        for i in range(66):
            eng.codon_site_classes(2.0 + 0.01 * (i + 1), np.linspace(0.05, 2.5, 64), np.full(64, 1.0 / 64), pi, bl)
        again = eng.codon_site_classes(2.0, [0.1, 0.9, 2.0], [0.5, 0.3, 0.2], pi, bl)            # evicted by now: decomposed anew
        q_again = eng.site_classes(c["Q"], c["proportions"], pi, bl)
        last = eng.codon_site_classes(2.66, np.linspace(0.05, 2.5, 64), np.full(64, 1.0 / 64), pi, bl)   # still resident
    assert first == fresh(2.0, [0.1, 0.9, 2.0], [0.5, 0.3, 0.2])
    assert grown == fresh(2.0, np.concatenate([[0.1, 0.9, 2.0], om]), np.concatenate([[0.2, 0.2, 0.2], 0.4 * pr]))
    assert again == first and q_again == q_lnL
    assert last == fresh(2.66, np.linspace(0.05, 2.5, 64), np.full(64, 1.0 / 64))
    assert abs(q_lnL - float(c["lnL_pyref"])) < TOTAL_ATOL


# ---------------------------------------------------------------------------------------------
# BEB site posteriors integrated over PAML's grid on the device (analysis/beb.py:513-661)
# ---------------------------------------------------------------------------------------------
def test_beb_grid_integration_against_the_reference_and_paml(gpu_engine_cls):
    from crabml_amd import beb
    from oracle import beb as obeb
    c = load_case("lysozyme_m2a_beb")
    cols, props = beb.paml_grid_m2a()
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        post, lnfx, ln_fx = eng.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
        table = eng.site_lnl_by_class(c["Q"], c["pi"], c["branch_lengths"])
        post2, ln_fx2 = beb.m2a_site_posteriors(eng, float(c["kappa"]), c["pi"], c["mle_proportions"], c["mle_omegas"], c["branch_lengths"])
        again = eng.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
        with pytest.raises(ValueError, match="class index out of range"):
            eng.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols + 19, props)
        with pytest.raises(ValueError, match="n_mix"):
            eng.beb_grid(c["Q"], c["pi"], c["branch_lengths"], np.zeros((4, 17), dtype=np.int32), np.ones((4, 17)))
    with _engine(gpu_engine_cls, c["parent"], c["sequences"], pattern_begin=0, pattern_end=64) as part:
        with pytest.raises(ValueError, match="whole alignment"):
            part.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
    ref = c["posteriors_ref"]                                  # the reference's compute_site_posteriors_paml
    big = ref > 1e-8
    assert np.abs(post - ref).max() <= 1e-11
    assert np.abs(post[big] / ref[big] - 1.0).max() <= 1e-9    # its input table came from the CPU oracle, ours from the device
    # against the oracle fed the device's own [S, 21] table: the integration alone, strict
    o_post, o_lnfx, o_ln_fx = obeb.site_posteriors(table, cols, props)
    assert np.abs(post[big] / o_post[big] - 1.0).max() <= SITE_RTOL
    assert np.abs(post - o_post).max() <= 1e-14
    np.testing.assert_allclose(lnfx, o_lnfx, rtol=1e-12)
    assert abs(ln_fx - o_ln_fx) < TOTAL_ATOL
    # PAML's printed Pr(w>1) (tests/data/paml_reference/lysozyme_m2a_out.txt:256-263)
    got = post[np.array([15, 17, 37, 41, 50, 62, 114, 126]) - 1, 2]
    assert np.abs(got - [0.693, 0.695, 0.744, 0.920, 0.835, 0.578, 0.689, 0.577]).max() < 5e-4
    np.testing.assert_allclose(post.sum(axis=1), 1.0, atol=1e-12)
    assert np.abs(post2 - post).max() <= 1e-12 and abs(ln_fx2 - ln_fx) < 1e-9   # host-built matrices differ in the last bit
    assert np.array_equal(again[0], post) and again[2] == ln_fx                 # fixed-order reductions: bit-identical repeats


def test_beb_grid_integration_at_size(gpu_engine_cls):
    """2 000 codons x 30 taxa under M2a: the device against the vectorised oracle on the device's own class table
    (the reference's loops would need 2 x 2e7 Python iterations here)."""
    from crabml_amd import beb, models, synth
    from oracle import beb as obeb
    case = synth.make_case("M8", 30, 2000, seed=31)
    m = models.M2aCodonModel(kappa=2.5, omega0=0.2, omega2=2.5, p0=0.7, p1=0.2, pi=case.pi)
    pr, om = m.get_site_classes()
    Q = beb.beb_rate_matrices(2.5, case.pi, pr, om)
    cols, props = beb.paml_grid_m2a()
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        post, lnfx, ln_fx = eng.beb_grid(Q, case.pi, case.branch_lengths, cols, props)
        table = eng.site_lnl_by_class(Q, case.pi, case.branch_lengths)
    o_post, o_lnfx, o_ln_fx = obeb.site_posteriors(table, cols, props)
    big = o_post > 1e-8
    assert np.abs(post[big] / o_post[big] - 1.0).max() <= SITE_RTOL
    assert np.abs(post - o_post).max() <= 1e-12
    np.testing.assert_allclose(lnfx, o_lnfx, rtol=1e-12)
    assert abs(ln_fx - o_ln_fx) < TOTAL_ATOL * 10      # a sum of 2 000 terms of magnitude ~30: 1e-12 relative


def test_many_genes_on_concurrent_engines(gpu_engine_cls, oracle):
    """Config 5 shape: independent genes, several engines of one GPU at once (crabml_amd.dist.map_genes); every gene's
    result equals its stand-alone evaluation bit for bit and matches the oracle."""
    from crabml_amd import synth
    from crabml_amd.dist import map_genes
    genes = [synth.make_case("M8" if g % 2 else "M7", 8 + g, 60 + 7 * g, seed=300 + g) for g in range(7)]

    def fn(eng, case):
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        return eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)

    together = map_genes(fn, genes, workers=3, make_engine=lambda: gpu_engine_cls(0))
    with gpu_engine_cls(0) as eng:
        alone = [fn(eng, case) for case in genes]      # one engine re-targeted gene after gene
    assert together == alone
    for case, lnL in zip(genes, together):
        ref = oracle.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, case.tree_structure,
                                                       case.branch_lengths, None, case.codes, case.leaf_node_ids)
        assert abs(lnL - ref) < TOTAL_ATOL


# ---------------------------------------------------------------------------------------------
# BASELINE configs 3 and 4 tree shapes (500 and 1000 taxa) at a number of sites the oracle finishes in seconds
# ---------------------------------------------------------------------------------------------
def test_config3_branch_site_500_taxa(gpu_engine_cls, oracle):
    from crabml_amd import synth
    case = synth.make_case("ModelA", 500, 600, seed=3)
    args = (case.qfactors[0], case.qfactors[1], case.omega0, case.omega2, case.site_class_freqs)
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        lnL, site = eng.branch_site(case.Q3, *args, case.pi, case.branch_lengths, case.branch_labels, return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(4)])
        from_params = eng.codon_branch_site(case.kappa, case.omega0, case.omega2, case.p0, case.p1, case.pi, case.branch_lengths,
                                            case.branch_labels)
        work, n_cl = eng.last_work(), eng.last_clade_tables()[0]
        n_pat = eng.n_patterns
    check_work(work, n_cl, 497 * n_pat * 4, 500 * n_pat * 4)
    ref, ref_site = oracle.compute_branch_site_log_likelihood(case.Q3[0], case.Q3[1], case.Q3[2], *args, case.pi, case.tree_structure,
                                                              case.branch_lengths, case.branch_labels, None, case.codes,
                                                              case.leaf_node_ids, return_sites=True)
    # the unscaled reference restatement underflows on a 500-taxon tree: compare where it is finite, and
    # strictly against the staged pruning of the device's own matrices with rescaling everywhere
    ok = np.isfinite(ref_site) & (ref_site > -600.0)
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, case.pi, case.tree_structure, case.codes, case.leaf_node_ids)
    mx = lnf.max(axis=1)
    staged = mx + np.log((np.exp(lnf - mx[:, None]) * case.site_class_freqs[None, :]).sum(axis=1))

    def linear_mix(x):
        m = x.max(axis=1)
        return m + np.log((np.exp(x - m[:, None]) * case.site_class_freqs[None, :]).sum(axis=1))

    bound = eigen_route_bound(oracle, P_dev, case.pi, case.tree_structure, case.codes, case.leaf_node_ids, linear_mix)
    assert ok.any() and np.all(np.abs(site[ok] - ref_site[ok]) <= SITE_RTOL * np.abs(ref_site[ok]) + bound[ok])
    assert np.median(bound / np.abs(site)) < SITE_RTOL                  # on most sites the gate is the plain one (within 2x)
    assert _rel(site, staged) <= SITE_RTOL
    assert abs(lnL - staged.sum()) < 1e-6 * 10
    assert abs(from_params - lnL) < 1e-9 * abs(lnL)


def test_config4_free_ratio_1000_taxa(gpu_engine_cls, oracle):
    from crabml_amd import synth
    case = synth.make_case("free-ratio", 1000, 400, seed=4)
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        lnL, site = eng.branch(case.Q_per_node, case.pi, case.branch_lengths, no_clamp=True, return_sites=True)
        P_dev = eng.export_transition_matrices(0)
        from_params = eng.codon_branch(case.kappa, case.omega_per_node, case.pi, case.branch_lengths, no_clamp=True)
        n_pat = eng.n_patterns_total
        # the 8-GPU sharding of this configuration: slices add up
        parts = 0.0
    for a, b in zip(np.linspace(0, n_pat, 4).astype(int)[:-1], np.linspace(0, n_pat, 4).astype(int)[1:]):
        with _engine(gpu_engine_cls, case.parent, case.codes, pattern_begin=int(a), pattern_end=int(b)) as eng:
            parts += eng.branch(case.Q_per_node, case.pi, case.branch_lengths, no_clamp=True)
    lnf = oracle.site_lnf_from_transition_matrices(P_dev[None], case.pi, case.tree_structure, case.codes, case.leaf_node_ids)
    assert _rel(site, lnf[:, 0]) <= SITE_RTOL
    assert abs(lnL - lnf[:, 0].sum()) < 1e-6 * 10
    assert abs(parts - lnL) < 1e-7
    assert abs(from_params - lnL) < 1e-9 * abs(lnL)
    for v in (0, 500, 1500):   # kernel 1 on a few of the 1 998 per-node matrices
        np.testing.assert_allclose(P_dev[v], oracle.matrix_exponential(case.Q_per_node[v], case.pi, case.branch_lengths[v]), atol=2e-14, rtol=0)


def test_random_tree_shapes_sweep(gpu_engine_cls, oracle):
    """40 random trees with polytomies, unary internal nodes, leaves hanging off the root, zero-length branches, missing
    data and leaf rows in shuffled order: the device's staged result must equal the oracle's pruning on every site."""
    from crabml_amd import synth
    base = synth.make_case("M8", 6, 10, seed=77)
    K = 3
    Q, props, pi = base.Q[[0, 6, 10]], np.array([0.5, 0.3, 0.2]), base.pi
    rng = np.random.default_rng(2024 + int(os.environ.get("CRAB_SWEEP_SEED", "0")))
    for trial in range(40):
        n_leaves = int(rng.integers(2, 41))
        # grow a random rooted tree: repeatedly group 1..4 open subtrees under a new node (1 = unary node)
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1 or nxt == n_leaves:
            k = int(min(len(open_nodes), rng.choice([1, 2, 2, 2, 3, 4])))
            if len(open_nodes) == 1 and nxt > n_leaves and k == 1 and rng.random() < 0.7:
                break
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        n_nodes = nxt
        par = np.array([parent.get(v, -1) for v in range(n_nodes)], dtype=np.int32)
        bl = rng.exponential(0.08, size=n_nodes)
        bl[rng.random(n_nodes) < 0.1] = 0.0
        bl[par < 0] = 0.0
        S = int(rng.integers(1, 150))
        codes = rng.integers(0, 61, size=(n_leaves, S)).astype(np.int32)
        # make sites less random than uniform: copy a column pattern down to half the taxa
        codes[: n_leaves // 2] = codes[0]
        codes[rng.random(codes.shape) < 0.05] = 64
        codes[rng.random(codes.shape) < 0.01] = -1
        rows = rng.permutation(n_leaves).astype(np.int32)          # alignment row r belongs to leaf node rows[r]
        ts = [(v, None if par[v] < 0 else int(par[v])) for v in range(n_nodes)]
        with gpu_engine_cls(0) as eng:
            eng.set_tree(par, rows)
            eng.set_alignment(codes, n_states=61)
            lnL, site = eng.site_classes(Q, props, pi, bl, return_sites=True, no_clamp=True)
            P_dev = np.stack([eng.export_transition_matrices(k) for k in range(K)])
        lnf = oracle.site_lnf_from_transition_matrices(P_dev, pi, ts, codes, rows)
        want = _mixture(lnf, props, clamp=False)
        finite = np.isfinite(want)
        assert np.array_equal(finite, np.isfinite(site)), f"trial {trial}"
        assert _rel(site[finite], want[finite]) <= SITE_RTOL, f"trial {trial}: {n_leaves} leaves, {n_nodes} nodes, {S} sites"


def test_impossible_data_on_zero_length_branches_is_zero_not_nan(gpu_engine_cls):
    """Two different codons joined by zero-length branches have likelihood exactly 0 under any model.  exp(Q 0) from the
    eigen route (the reference's route too, matrix.rs:88-102) carries entries of +-1e-15 off the diagonal, so what comes
    out is rounding noise: zero or some 1e-20.  Evaluations clamp the negative entries (kernel 1) so that
    power-of-two rescaling never turns negative noise into infinities and NaN (found by the random sweep above): the
    site must come out as "essentially impossible", never NaN, and must not disturb its neighbours."""
    c = load_case("lysozyme_m2a")
    parent = np.array([4, 4, 5, 5, 6, 6, -1], dtype=np.int32)
    bl = np.array([0.0, 0.0, 0.1, 0.2, 0.05, 0.3, 0.0])
    codes = np.array([[3, 7], [9, 7], [3, 7], [20, 8]], dtype=np.int32)    # site 0: leaves 0 and 1 disagree at distance 0
    with _engine(gpu_engine_cls, parent, codes) as eng:
        lnL, site = eng.site_classes(c["Q"], c["proportions"], c["pi"], bl, return_sites=True, no_clamp=True)
        lnLc, sitec = eng.site_classes(c["Q"], c["proportions"], c["pi"], bl, return_sites=True)
        bc = eng.site_lnl_by_class(c["Q"], c["pi"], bl, no_clamp=True)
    assert not np.isnan(site).any() and not np.isnan(bc).any() and not np.isnan(lnL)
    assert site[0] < -30.0 and np.all(bc[0] < -30.0)            # noise level or exactly zero
    assert np.isfinite(site[1]) and np.isfinite(bc[1]).all() and -40.0 < site[1] < 0.0
    assert sitec[1] == site[1] and np.isfinite(lnLc) and LN_TINY - 1e-9 <= sitec[0] < -30.0


def test_random_state_spaces_models_and_slices_sweep(gpu_engine_cls, oracle):
    """Random reversible models on 4 / 20 / 61 / 63 states, 1-5 classes, random trees, every entry point of the handle
    API that the models share (mixture, per class, branch model, pattern slices), each against the oracle."""
    rng = np.random.default_rng(99 + int(os.environ.get("CRAB_SWEEP_SEED", "0")))
    for trial in range(16):
        n = int(rng.choice([4, 20, 61, 63]))
        K = int(rng.integers(1, 6))
        n_leaves = int(rng.integers(3, 25))
        # random binary-ish tree by random joins
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1:
            k = int(min(len(open_nodes), rng.choice([2, 2, 3])))
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        n_nodes = nxt
        par = np.array([parent.get(v, -1) for v in range(n_nodes)], dtype=np.int32)
        bl = rng.exponential(0.15, size=n_nodes) + 1e-4
        bl[par < 0] = 0.0
        pi = rng.dirichlet(3.0 * np.ones(n))
        Qs = []
        for _ in range(K):
            Sx = rng.gamma(1.0, 1.0, size=(n, n))
            Sx = (Sx + Sx.T) * (rng.random((n, n)) < 0.6)
            Sx = np.triu(Sx, 1) + np.triu(Sx, 1).T + 1e-3 * (1 - np.eye(n))
            Qm = Sx * pi[None, :]
            np.fill_diagonal(Qm, 0.0)
            np.fill_diagonal(Qm, -Qm.sum(axis=1))
            Qs.append(Qm / -(pi * np.diag(Qm)).sum() * rng.uniform(0.3, 2.0))
        Q = np.stack(Qs)
        props = rng.dirichlet(np.ones(K))
        S = int(rng.integers(1, 300))
        # evolve-ish data: a few ancestral states copied with noise, so that sites are not uniformly random
        anc = rng.integers(0, n, size=S)
        codes = np.where(rng.random((n_leaves, S)) < 0.6, anc[None, :], rng.integers(0, n, size=(n_leaves, S))).astype(np.int32)
        codes[rng.random(codes.shape) < 0.03] = n + 3          # missing
        ts = [(v, None if par[v] < 0 else int(par[v])) for v in range(n_nodes)]
        leaf = np.arange(n_leaves, dtype=np.int32)
        with gpu_engine_cls(0) as eng:
            eng.set_tree(par, leaf)
            eng.set_alignment(codes, n_states=n)
            lnL, site = eng.site_classes(Q, props, pi, bl, return_sites=True)
            P_dev = np.stack([eng.export_transition_matrices(k) for k in range(K)])
            bc = eng.site_lnl_by_class(Q, pi, bl)
            br = eng.branch(np.repeat(Q[:1], n_nodes, axis=0), pi, bl)
            one = eng.site_classes(Q[:1], [1.0], pi, bl)
            n_pat = eng.n_patterns_total
        ref, ref_site = oracle.compute_site_class_log_likelihood(list(Q), props, pi, ts, bl, None, codes, leaf, return_sites=True)
        ref_bc = oracle.compute_site_log_likelihoods_by_class(list(Q), pi, ts, bl, None, codes, leaf)
        tag = f"trial {trial}: n={n} K={K} leaves={n_leaves} S={S}"
        assert abs(lnL - ref) < TOTAL_ATOL, tag
        # end to end, every site, within the eigen-route bound (module docstring): these are random dense models on
        # partly random data, where many sites lean on small entries of P
        bound = eigen_route_bound(oracle, P_dev, pi, ts, codes, leaf, lambda x: _mixture(x, props))
        assert np.all(np.abs(site - ref_site) <= SITE_RTOL * np.maximum(np.abs(ref_site), 1e-3) + bound), tag
        assert_by_class_close(bc, ref_bc, tag, rtol=1e-7)
        assert br == one, tag
        cut = int(rng.integers(0, n_pat + 1))
        parts = 0.0
        for a, b in ((0, cut), (cut, n_pat)):
            with gpu_engine_cls(0) as eng:
                eng.set_tree(par, leaf)
                eng.set_alignment(codes, n_states=n, pattern_begin=a, pattern_end=b)
                parts += eng.site_classes(Q, props, pi, bl)
        assert abs(parts - lnL) < 1e-8, tag


def test_random_branch_site_and_extreme_branch_lengths_sweep(gpu_engine_cls, oracle):
    """Model A on random trees with random foreground labels, branch lengths from 1e-3 to 30 substitutions per codon
    (below that, with partly random codons, the eigen route's +-1e-16 noise in P decides the answer: any two
    decompositions of matrices that differ in the last bit then disagree, the reference's included),
    pattern counts around the 64-pattern tile boundaries: staged strict against the oracle's pruning of the device's
    own matrices (naive, unscaled where it stays finite; the rescaled device result must be finite everywhere)."""
    from crabml_amd import models, synth
    base = synth.make_case("M0", 5, 4, seed=5)
    pi = base.pi
    rng = np.random.default_rng(11 + int(os.environ.get("CRAB_SWEEP_SEED", "0")))
    for trial, S in enumerate([1, 63, 64, 65, 127, 128, 129, 200, 257, 31]):
        n_leaves = int(rng.integers(4, 30))
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1:
            k = int(min(len(open_nodes), rng.choice([2, 2, 2, 3])))
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        n_nodes = nxt
        par = np.array([parent.get(v, -1) for v in range(n_nodes)], dtype=np.int32)
        bl = 10.0 ** rng.uniform(-3, 1.5, size=n_nodes) if trial % 2 else rng.exponential(0.1, size=n_nodes)
        bl[par < 0] = 0.0
        labels = (rng.random(n_nodes - 1) < 0.3).astype(np.uint8)
        if not labels.any():
            labels[0] = 1
        if labels.all():
            labels[1] = 0
        kappa, w0, w2, p0, p1 = 2.0 + rng.random(), 0.05 + 0.5 * rng.random(), 1.0 + 4 * rng.random(), 0.5, 0.3
        if trial == 3:
            w2 = 1.0                                         # the null model: classes 2a/2b share Q(1) on the foreground
        ma = models.BranchSiteModelA(pi, labels.astype(int))
        fr = ma.compute_site_class_frequencies(p0, p1)
        qb, qf = ma.compute_qfactors(kappa, p0, p1, w0, w2)
        Q3 = np.stack(ma.unnormalised_Qs(kappa, w0, w2))
        anc = rng.integers(0, 61, size=S)
        codes = np.where(rng.random((n_leaves, S)) < 0.7, anc[None, :], rng.integers(0, 61, size=(n_leaves, S))).astype(np.int32)
        codes[rng.random(codes.shape) < 0.03] = 64
        ts = [(v, None if par[v] < 0 else int(par[v])) for v in range(n_nodes)]
        leaf = np.arange(n_leaves, dtype=np.int32)
        with gpu_engine_cls(0) as eng:
            eng.set_tree(par, leaf)
            eng.set_alignment(codes, n_states=61, compress=False)
            assert eng.n_patterns == S
            lnL, site = eng.branch_site(Q3, qb, qf, w0, w2, fr, pi, bl, labels, return_sites=True)
            P_dev = np.stack([eng.export_transition_matrices(k) for k in range(4)])
            pars = eng.codon_branch_site(kappa, w0, w2, p0, p1, pi, bl, labels)
            P_par = np.stack([eng.export_transition_matrices(k) for k in range(4)])
        tag = f"trial {trial}: leaves={n_leaves} S={S}"
        assert np.isfinite(site).all(), tag
        lnf = oracle.site_lnf_from_transition_matrices(P_dev, pi, ts, codes, leaf)      # rescaled pruning of the same matrices
        mx = lnf.max(axis=1)
        staged = mx + np.log((np.exp(lnf - mx[:, None]) * fr[None, :]).sum(axis=1))
        assert _rel(site, staged) <= SITE_RTOL, tag
        assert abs(lnL - staged.sum()) < TOTAL_ATOL * max(1.0, S / 50), tag
        # the parameter path: its matrices equal the Q path's to the eigen route's absolute accuracy, and its pruning is
        # strict on its own matrices.  (The two totals themselves are not compared: with partly random codons the answer
        # leans on P entries of 1e-12 that the eigen route knows to 1e-16 absolute, and any two decompositions of
        # matrices differing in the last bit disagree at 1e-8 ... 1e-4 relative depending on the draw.)
        assert np.abs(P_par - P_dev).max() <= 1e-13, tag
        lnf_p = oracle.site_lnf_from_transition_matrices(P_par, pi, ts, codes, leaf)
        mxp = lnf_p.max(axis=1)
        staged_p = (mxp + np.log((np.exp(lnf_p - mxp[:, None]) * fr[None, :]).sum(axis=1))).sum()
        assert abs(pars - staged_p) < TOTAL_ATOL * max(1.0, S / 50), tag


def test_beb_adaptive_grid_accumulation_against_the_reference(gpu_engine_cls):
    """BEBCalculator.compute_site_posteriors (analysis/beb.py:663-731): nine weighted M8 grid points accumulated on the
    device, from parameters (quantiles and rate matrices built behind the boundary), against the reference's output."""
    from crabml_amd import beb
    from crabml_amd.engine import build_site_class_q_matrices, discretise_omega
    from oracle import beb as obeb
    c = load_case("lysozyme_m8_beb_grid")
    params = [(pt[0], np.append(discretise_omega("beta", pt[2], pt[3], 10), max(pt[4], 1.001)), pr)
              for pt, pr in zip(c["grid_points"], c["grid_proportions"])]
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        post = beb.site_posteriors_over_grid(eng, params, c["weights"], c["pi"], c["branch_lengths"], batch=4)   # 4 + 4 + 1
        once = beb.site_posteriors_over_grid(eng, params, c["weights"], c["pi"], c["branch_lengths"], batch=64)
        tables = [eng.site_lnl_by_class(build_site_class_q_matrices(k, om, pr, c["pi"]), c["pi"], c["branch_lengths"])
                  for k, om, pr in params]
        with pytest.raises(ValueError, match="at most 32"):
            eng.codon_class_posteriors([2.0], np.linspace(0.1, 2, 33)[None], np.full((1, 33), 1 / 33), [1.0], c["pi"], c["branch_lengths"])
    ref = c["posteriors_ref"]
    big = ref > 1e-8
    assert np.abs(post - ref).max() <= 1e-11 and np.abs(post[big] / ref[big] - 1.0).max() <= 1e-9
    assert np.abs(once - post).max() <= 1e-15                       # batching only changes where the partial sums are added
    strict = obeb.site_posteriors_weighted(tables, c["grid_proportions"], c["weights"])   # the device's own tables through the oracle
    assert np.abs(post[big] / strict[big] - 1.0).max() <= 1e-9 and np.abs(post - strict).max() <= 1e-12
    np.testing.assert_allclose(post.sum(axis=1), c["weights"].sum(), atol=1e-12)


def test_beb_hessian_stencil_and_adaptive_grid_against_the_reference(gpu_engine_cls):
    """calculate_beb with grid_type='adaptive' (analysis/beb.py:746-771): the 21 distinct points of the reference's
    Hessian stencil in one batched call against the 60 likelihoods the reference asked for, the accumulated posteriors
    over the reference's 30-point grid against the reference's output, and the whole pipeline end to end (Hessian on
    the device's likelihoods: the stencil's 1/h^2 = 1e10 makes that grid the device's own, see tests/test_beb.py)."""
    from crabml_amd import beb, models
    from test_beb import _asked_order
    c = load_case("lysozyme_m2a_adaptive_grid")
    names = [str(s) for s in c["param_names"]]
    mle = dict(zip(names, map(float, c["mle"])))
    pts, single, pair = beb.hessian_points(c["mle"])
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        f = beb.negative_log_likelihoods(eng, models.M2aCodonModel, mle, names, pts, c["pi"], c["branch_lengths"])
        outside = np.vstack([pts[:3], [[2.0, 0.3, 2.0, 1.2, 0.1]]])
        f_out = beb.negative_log_likelihoods(eng, models.M2aCodonModel, mle, names, outside, c["pi"], c["branch_lengths"])
        H = beb.hessian(eng, models.M2aCodonModel, mle, names, c["pi"], c["branch_lengths"])
        cls = [beb._class_parameters(models.M2aCodonModel, mle, names, p, c["pi"]) for p in c["grid_points_ref"]]
        post = beb.site_posteriors_over_grid(eng, cls, c["weights_ref"], c["pi"], c["branch_lengths"])
        full, mean, se, grid, w = beb.adaptive_site_posteriors(eng, models.M2aCodonModel, mle, names, c["pi"], c["branch_lengths"], 3)
    for r, lnl in zip(_asked_order(single, pair), c["asked_lnl"]):
        assert abs(f[r] + lnl) <= SITE_RTOL * abs(lnl), r
    assert np.isinf(f_out[3]) and np.array_equal(f_out[:3], f[:3])
    np.testing.assert_array_equal(H, beb.hessian_from_values(f, single, pair))
    assert np.array_equal(H, H.T)
    ref = c["posteriors_ref"]
    big = ref > 1e-8
    assert np.abs(post - ref).max() <= 1e-11 and np.abs(post[big] / ref[big] - 1.0).max() <= 1e-9
    assert full.shape == ref.shape and 1 <= len(grid) <= 243 and abs(w.sum() - 1.0) < 1e-12
    np.testing.assert_allclose(full.sum(axis=1), 1.0, atol=1e-12)
    om = np.array([min(mle["omega0"], 0.999), 1.0, max(mle["omega2"], 1.001)])
    assert (mean >= om[0] - 1e-12).all() and (mean <= om[2] + 1e-12).all() and (se >= 0).all()


def test_beb_grid_over_pattern_slices_equals_the_whole(gpu_engine_cls):
    """The multi-GPU form of the BEB integration: three pattern slices, their lnfX vectors added, posteriors of each
    slice's own sites from the sum -- together they must tile the whole-alignment result."""
    from crabml_amd import beb
    from crabml_amd.dist import sharded_beb_grid
    c = load_case("lysozyme_m2a_beb")
    cols, props = beb.paml_grid_m2a()
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        whole, lnfx, ln_fx = eng.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
        n_pat = eng.n_patterns_total
        alone, ln_alone = sharded_beb_grid(eng, c["Q"], c["pi"], c["branch_lengths"], cols, props)   # world size 1: same protocol
    edges = [0, 37, 90, n_pat]
    engines = [_engine(gpu_engine_cls, c["parent"], c["sequences"], pattern_begin=a, pattern_end=b) for a, b in zip(edges[:-1], edges[1:])]
    try:
        parts = [e.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props, partial=True)[1] for e in engines]
        total = parts[0] + parts[1] + parts[2]
        tiled = np.full_like(whole, np.nan)
        for e in engines:
            post, _, lf = e.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props, lnfx=total)
            have = ~np.isnan(post[:, 0])
            assert not (have & ~np.isnan(tiled[:, 0])).any()
            tiled[have] = post[have]
            assert abs(lf - ln_fx) < 1e-9
        with pytest.raises(ValueError, match="whole alignment"):
            engines[0].beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
    finally:
        for e in engines:
            e.close()
    np.testing.assert_allclose(total, lnfx, rtol=1e-12)
    assert not np.isnan(tiled).any() and np.abs(tiled - whole).max() <= 1e-12
    assert np.abs(alone - whole).max() <= 1e-13 and abs(ln_alone - ln_fx) < 1e-10


def test_parameter_changes_decompose_only_what_they_must(gpu_engine_cls):
    """The eigensystem pool is keyed by (kappa, omega, pi) with every normalisation folded into the branch lengths:
    repeating a point, moving a branch length or moving a class proportion decomposes nothing; moving one omega
    decomposes one matrix; moving kappa decomposes all K."""
    c = load_case("lysozyme_m8")
    ka, om, pr, pi, bl = float(c["kappa"]), c["omegas"], c["proportions"], c["pi"], c["branch_lengths"]
    K = len(om)
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        def misses(fn):
            before = eng.eig_pool_stats()[1]
            fn()
            return eng.eig_pool_stats()[1] - before
        assert misses(lambda: eng.codon_site_classes(ka, om, pr, pi, bl)) == K
        assert misses(lambda: eng.codon_site_classes(ka, om, pr, pi, bl)) == 0
        bl2 = bl.copy(); bl2[3] *= 1.001
        assert misses(lambda: eng.codon_site_classes(ka, om, pr, pi, bl2)) == 0
        pr2 = pr.copy(); pr2[:-1] *= 0.95 / pr[:-1].sum(); pr2[-1] = 0.05
        a = eng.codon_site_classes(ka, om, pr, pi, bl)
        assert misses(lambda: eng.codon_site_classes(ka, om, pr2, pi, bl)) == 0       # the shared scale changed: t / scale, same eigensystems
        assert eng.codon_site_classes(ka, om, pr2, pi, bl) != a
        om2 = om.copy(); om2[-1] *= 1.01
        assert misses(lambda: eng.codon_site_classes(ka, om2, pr, pi, bl)) == 1
        assert misses(lambda: eng.codon_site_classes(ka * 1.01, om, pr, pi, bl)) == K
        # a whole gradient stencil in one call: 1 + n_branches + K + 1 vectors, K + K + 1 ... new matrices only for the omega and kappa rows
        B = 1 + 4 + 2
        kas = np.full(B, ka); kas[6] *= 1.02
        oms = np.repeat(om[None], B, 0); oms[5, 0] *= 1.02
        bls = np.repeat(bl[None], B, 0); bls[np.arange(1, 5), np.arange(4)] *= 1.001
        assert misses(lambda: eng.codon_site_classes(kas, oms, np.repeat(pr[None], B, 0), pi, bls)) == 1 + K
        hits, miss, resident = eng.eig_pool_stats()
        assert resident == miss and hits > 0
        # matrices passed by value share the pool, keyed by their contents
        assert misses(lambda: eng.site_classes(c["Q"], pr, pi, bl)) == K
        assert misses(lambda: eng.site_classes(c["Q"], pr, pi, bl2)) == 0


def test_random_batches_equal_single_evaluations_bit_for_bit(gpu_engine_cls):
    """Batches share identical transition matrices between their vectors: whatever the pattern of repeats (identical
    vectors, one branch moved, many branches moved, a class or kappa moved), every element must equal the evaluation of
    its vector alone, bit for bit, for rate matrices passed by value and for parameters."""
    from crabml_amd import synth
    rng = np.random.default_rng(314 + int(os.environ.get("CRAB_SWEEP_SEED", "0")))
    case = synth.make_case("M8", 14, 150, seed=8)
    K, nn = len(case.proportions), case.n_nodes
    om, pr = case.class_omegas, np.asarray(case.proportions)
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        for B in (2, 5, 17):
            bl = np.repeat(case.branch_lengths[None], B, 0)
            ka = np.full(B, case.kappa)
            oms, prs = np.repeat(om[None], B, 0), np.repeat(pr[None], B, 0)
            Q = np.repeat(case.Q[None], B, 0)
            for b in range(1, B):
                kind = rng.integers(0, 6)
                if kind == 0:
                    pass                                                    # identical to the base vector
                elif kind == 1:
                    bl[b, rng.integers(0, nn - 1)] *= 1.0 + 1e-6
                elif kind == 2:
                    bl[b, rng.random(nn) < 0.5] *= 1.0 + 1e-3
                elif kind == 3:
                    ka[b] *= 1.0 + 1e-4
                    Q[b] = Q[b] * (1.0 + 1e-4)
                elif kind == 4:
                    k = rng.integers(0, K)
                    oms[b, k] *= 1.0 + 1e-4
                    Q[b, k] = Q[b, k] * (1.0 + 1e-4)
                else:
                    bl[b] = bl[rng.integers(0, b)]                          # repeats an earlier vector's branch lengths
            got = eng.codon_site_classes(ka, oms, prs, case.pi, bl)
            want = [eng.codon_site_classes(ka[b], oms[b], prs[b], case.pi, bl[b]) for b in range(B)]
            assert list(got) == want, f"parameters, B={B}"
            got_q = eng.site_classes_batch(Q, prs, case.pi, bl)
            want_q = [eng.site_classes(Q[b], prs[b], case.pi, bl[b]) for b in range(B)]
            assert list(got_q) == want_q, f"matrices by value, B={B}"
            assert np.isfinite(got).all() and np.isfinite(got_q).all()


def test_batched_stencil_over_pattern_slices_adds_up(gpu_engine_cls):
    """What a pattern-sharded fit exchanges (crabml_amd/optimize.py with world_size > 1): every rank evaluates the whole
    stencil on its slice and the B partial sums are added in rank order.  Three slices, the last one empty (a rank left
    without patterns on a small alignment), against the unsharded batch."""
    from crabml_amd.dist import pattern_slices
    c = load_case("lysozyme_m2a")
    pi, bl = c["pi"], c["branch_lengths"]
    B = 9
    rng = np.random.default_rng(4)
    kappa = 2.0 + rng.random(B)
    omegas = np.column_stack([0.1 + 0.5 * rng.random(B), np.ones(B), 1.5 + rng.random(B)])
    props = rng.dirichlet([5, 3, 2], size=B)
    bls = np.tile(bl, (B, 1)) * (1 + 0.01 * rng.random((B, len(bl))))
    with _engine(gpu_engine_cls, c["parent"], c["sequences"]) as eng:
        whole = eng.codon_site_classes(kappa, omegas, props, pi, bls)
        n_pat = eng.n_patterns_total
    slices = pattern_slices(n_pat, 3, align=64)
    assert slices[-1][0] == slices[-1][1] == n_pat and slices[0] == (0, 64)
    total = np.zeros(B)
    for a, b in slices:
        with _engine(gpu_engine_cls, c["parent"], c["sequences"], pattern_begin=a, pattern_end=b) as eng:
            part = eng.codon_site_classes(kappa, omegas, props, pi, bls)
            assert eng.n_patterns == b - a and (a < b or np.all(part == 0.0))
        total = total + part
    assert np.isfinite(whole).all() and np.abs(total - whole).max() < 1e-9


def test_beb_calculator_front_door(gpu_engine_cls):
    """BEBCalculator(...).calculate_beb() as the reference's callers use it (analysis/positive_selection.py), on lysozyme
    at PAML's M2a estimates: Pr(omega > 1) of the sites PAML lists, to its three printed decimals
    (tests/data/paml_reference/lysozyme_m2a_out.txt:256-263); the adaptive grid runs end to end."""
    from crabml_amd import beb, models
    from crabml_amd.io import Alignment, Tree
    aln = Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"), seqtype="codon")
    c = load_case("lysozyme_m2a_beb")
    # PAML's M2a branch lengths for lysozyme (data of tests/test_rust/test_paml_rust_validation.py:173-178, the tree the
    # fixture was made with)
    tree = Tree.from_newick("((Hsa_Human: 0.025240, Hla_gibbon: 0.041269): 0.077423, ((Cgu/Can_colobus: 0.044409, "
                            "Pne_langur: 0.053758): 0.085025, Mmu_rhesus: 0.019175): 0.046022, (Ssc_squirrelM: 0.042962, "
                            "Cja_marmoset: 0.024797): 0.136928);")
    mle = dict(kappa=5.03734, omega0=0.35983, omega2=4.52341, p0=0.77376, p1=0.10399)
    names = ["kappa", "omega0", "omega2", "p0", "p1"]
    calc = beb.BEBCalculator(mle, aln, tree, models.M2aCodonModel, names, grid_type="paml")
    res = calc.calculate_beb()
    assert res.model_name == "M2aCodonModel" and res.posterior_probs.shape == (130, 3)
    got = res.posterior_probs[np.array([15, 17, 37, 41, 50, 62, 114, 126]) - 1, 2]
    assert np.abs(got - [0.693, 0.695, 0.744, 0.920, 0.835, 0.578, 0.689, 0.577]).max() < 5e-4
    assert np.abs(res.posterior_probs - c["posteriors_ref"]).max() < 1e-9
    assert list(res.significant_sites(0.9)) == [41] and np.all(res.posterior_omega_se >= 0)
    ad = beb.BEBCalculator(mle, aln, tree, models.M2aCodonModel, names, grid_type="adaptive", n_grid_points=3).calculate_beb()
    np.testing.assert_allclose(ad.posterior_probs.sum(axis=1), 1.0, atol=1e-12)
    with pytest.raises(ValueError, match="Unknown grid_type"):
        beb.BEBCalculator(mle, aln, tree, models.M2aCodonModel, names, grid_type="nope").calculate_beb()
    # M8 through the adaptive grid (the reference's default for it): 11 classes, beta shape parameters in the Hessian
    mle8 = dict(kappa=5.029752, p0=0.868486, p_beta=9.517504, q_beta=13.543157, omega_s=4.370617, ncatG=10)
    r8 = beb.BEBCalculator(mle8, aln, tree, models.M8CodonModel, ["kappa", "p0", "p_beta", "q_beta", "omega_s"], n_grid_points=3).calculate_beb()
    assert r8.posterior_probs.shape == (130, 11) and r8.class_omegas[-1] == 4.370617 and len(r8.site_classes) == 11
    np.testing.assert_allclose(r8.posterior_probs.sum(axis=1), 1.0, atol=1e-12)
    assert r8.posterior_probs[40, -1] > 0.8 and r8.posterior_probs[40, -1] == r8.posterior_probs[:, -1].max()      # site 41 again


# ---------------------------------------------------------------------------------------------
# full-size configurations: size-independent properties where the oracle cannot go (configs 3 and 4, and the
# north_star target workload M8 on 500 taxa x 50 000 codons)
# ---------------------------------------------------------------------------------------------
def _full_size_properties(Engine, case, evaluate, tag, n_slices=8):
    """evaluate(engine) -> (lnL, per-site lnL).  Checks, on the whole alignment: a bit-identical repeat; that a 5 %
    column subsample gives, site for site, the bits of the whole run, compressed and uncompressed; that eight
    tile-aligned pattern slices (the 8-GPU sharding) tile the per-site output exactly and add up to the total within
    the north_star's 1e-6; that a column permutation permutes the output.  Returns (lnL, site, engine facts)."""
    from crabml_amd.dist import pattern_slices
    with _engine(Engine, case.parent, case.codes) as eng:
        lnL, site = evaluate(eng)
        again, site2 = evaluate(eng)
        n_pat = eng.n_patterns_total
        dense, gather = eng.last_work()
        n_clades = eng.last_clade_tables()[0]
        prune_ms = eng.last_timing()[1]
    assert np.isfinite(lnL) and np.isfinite(site).all(), tag
    assert again == lnL and np.array_equal(site, site2), tag
    assert abs(site.sum() - lnL) < 1e-6 * max(1.0, abs(lnL) / 1e6), tag
    cols = np.arange(7, case.codes.shape[1], 20)
    sub = np.ascontiguousarray(case.codes[:, cols])
    with _engine(Engine, case.parent, sub) as eng:
        _, s_c = evaluate(eng)
    with _engine(Engine, case.parent, sub, compress=False) as eng:
        _, s_r = evaluate(eng)
        assert eng.n_patterns == len(cols)
    assert np.array_equal(s_c, s_r) and np.array_equal(s_c, site[cols]), tag
    total, filled = 0.0, np.full(case.codes.shape[1], np.nan)
    for a, b in pattern_slices(n_pat, n_slices, align=64):
        with _engine(Engine, case.parent, case.codes, pattern_begin=a, pattern_end=b) as eng:
            part, s = evaluate(eng)
        total += part
        have = np.isfinite(s)
        assert not (have & np.isfinite(filled)).any(), tag
        filled[have] = s[have]
    assert np.array_equal(filled, site), tag
    assert abs(total - lnL) < TOTAL_ATOL, f"{tag}: slices {total!r} vs whole {lnL!r}"
    perm = np.random.default_rng(1).permutation(len(cols))
    with _engine(Engine, case.parent, np.ascontiguousarray(sub[:, perm])) as eng:
        _, sp = evaluate(eng)
    assert np.array_equal(sp, s_c[perm]), tag
    return lnL, site, dict(n_pat=n_pat, dense=dense, gather=gather, n_clades=n_clades, prune_ms=prune_ms)


def test_config3_model_a_500x50k_full_size_properties(gpu_engine_cls):
    from crabml_amd import synth
    case = synth.make_case("ModelA", 500, 50000, seed=3)
    args = (case.qfactors[0], case.qfactors[1], case.omega0, case.omega2, case.site_class_freqs)

    def evaluate(eng):
        return eng.branch_site(case.Q3, *args, case.pi, case.branch_lengths, case.branch_labels, return_sites=True)

    lnL, site, facts = _full_size_properties(gpu_engine_cls, case, evaluate, "config 3")
    check_work((facts["dense"], facts["gather"]), facts["n_clades"], 497 * facts["n_pat"] * 4, 500 * facts["n_pat"] * 4)
    assert -400.0 * 50000 < lnL < -20.0 * 50000


def test_config4_free_ratio_1000x100k_full_size_properties(gpu_engine_cls):
    from crabml_amd import synth
    case = synth.make_case("free-ratio", 1000, 100000, seed=4)

    def evaluate(eng):
        return eng.branch(case.Q_per_node, case.pi, case.branch_lengths, no_clamp=True, return_sites=True)

    lnL, site, facts = _full_size_properties(gpu_engine_cls, case, evaluate, "config 4")
    check_work((facts["dense"], facts["gather"]), facts["n_clades"], 997 * facts["n_pat"], 1000 * facts["n_pat"])
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        from_params = eng.codon_branch(case.kappa, case.omega_per_node, case.pi, case.branch_lengths, no_clamp=True)
    assert abs(from_params - lnL) < 1e-9 * abs(lnL)


def test_target_workload_m8_500x50k(gpu_engine_cls, oracle):
    """BASELINE.json north_star's target: M8 on 500 taxa x 50 000 codons.  Full-size properties as for configs 3 and 4;
    parity on a 1 000-column subsample: staged (the device's own matrices through the oracle's pruning) strict at 1e-10
    on every site, end to end against the oracle's reference mode within the per-site eigen-route bound."""
    from util import eigen_route_bound
    from crabml_amd import synth
    case = synth.make_case("M8", 500, 50000, seed=6)
    Q, pr, pi, bl = case.Q, case.proportions, case.pi, case.branch_lengths

    def evaluate(eng):
        return eng.site_classes(Q, pr, pi, bl, return_sites=True)

    lnL, site, facts = _full_size_properties(gpu_engine_cls, case, evaluate, "target workload")
    check_work((facts["dense"], facts["gather"]), facts["n_clades"], 497 * facts["n_pat"] * 11, 500 * facts["n_pat"] * 11)
    cols = np.linspace(0, 49999, 1000).astype(int)
    sub = np.ascontiguousarray(case.codes[:, cols])
    with _engine(gpu_engine_cls, case.parent, sub) as eng:
        lnL_s, site_s = eng.site_classes(Q, pr, pi, bl, return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(11)])
    assert np.array_equal(site_s, site[cols])
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, pi, case.tree_structure, sub, case.leaf_node_ids)
    staged = _mixture(lnf, pr)
    assert _rel(site_s, staged) <= SITE_RTOL
    assert abs(lnL_s - staged.sum()) < TOTAL_ATOL
    ref, ref_site = oracle.compute_site_class_log_likelihood(list(Q), pr, pi, case.tree_structure, bl, None, sub, case.leaf_node_ids,
                                                             return_sites=True)
    assert np.isfinite(ref_site).all() and ref_site.min() > LN_TINY + 1.0      # the unscaled reference has not underflowed here
    bound = eigen_route_bound(oracle, P_dev, pi, case.tree_structure, sub, case.leaf_node_ids, lambda x: _mixture(x, pr))
    assert np.all(np.abs(site_s - ref_site) <= SITE_RTOL * np.abs(ref_site) + bound)
    assert abs(lnL_s - ref) < TOTAL_ATOL


# ---------------------------------------------------------------------------------------------
# the tail of the per-site gate against 60-digit arithmetic (tests/golden/make_parity_tail.py)
# ---------------------------------------------------------------------------------------------
def _parity_tail_case(name):
    from crabml_amd import synth
    if name == "c2":
        c = synth.make_case("M8", 100, 10000, seed=2)
        return dict(kind="classes", Q=c.Q, props=np.asarray(c.proportions), pi=c.pi, parent=c.parent, bl=c.branch_lengths, codes=c.codes,
                    ts=c.tree_structure, leaf=c.leaf_node_ids)
    if name == "free_ratio_60":
        c = synth.make_case("free-ratio", 60, 1500, seed=22)
        return dict(kind="branch", Q=c.Q_per_node, props=np.array([1.0]), pi=c.pi, parent=c.parent, bl=c.branch_lengths, codes=c.codes,
                    ts=c.tree_structure, leaf=c.leaf_node_ids)
    c = load_case({"lysin": "lysin_m2a_gaps", "synthetic_edge": "synthetic_edge"}[name])
    return dict(kind="classes", Q=c["Q"], props=c["proportions"], pi=c["pi"], parent=c["parent"], bl=c["branch_lengths"], codes=c["sequences"],
                ts=tree_structure(c["parent"]), leaf=leaf_ids(c))


@pytest.mark.parametrize("name", ["c2", "lysin", "synthetic_edge", "free_ratio_60"])
def test_parity_tail_sites_against_60_digit_truth(gpu_engine_cls, oracle, name):
    """On the sites where eigen routes disagree most (and on control sites), the device's per-site log-likelihood against
    the value computed at 60 decimal digits: inside the 1e-10 gate wherever the reference's own route (numpy's LAPACK
    eigh + inverse, matrix.rs:41-67) is, and everywhere within the ONE mechanism that governs ill-conditioned sites in
    this suite: 1e-10 relative plus the per-site bound of util.eigen_route_bound evaluated at the device's own transition
    matrices (how far two evaluations can be apart whose P(t) entries agree to 2e-14; the device's entries are within
    2e-14 of the oracle's, asserted per fixture above, and within 1.1e-15 of the 60-digit ones, tests/golden/
    make_parity_tail.py).  Round 2 allowed the device ten times the LAPACK route's error instead; what separates two
    correct routes on such a site is rounding noise of 1e-16 absolute in entries of P that are themselves ~1e-8, and the
    last bits of either route decide the factor (measured, device / LAPACK route: 7.0e-13 / 8.2e-13 on configs[1], 2.4e-11 / 5.3e-11 on synthetic_edge, 2.7e-13 /
    9.6e-13 on the free-ratio case; on lysin's one such site 1.0e-9 ... 4.8e-9 across three builds that differ only in
    the summation order of the re-orthogonalisation, against 1.0e-9 -- and 1.7e-8 before the eigenvectors were
    re-orthogonalised).  The rigorous statement is the per-site bound of util.eigen_route_bound, asserted in
    tests/test_oracle.py for all three routes and the truth, and in this file for the device."""
    z = np.load(os.path.join(GOLDEN, "parity_tail.npz"), allow_pickle=False)
    c = _parity_tail_case(name)
    sites, truth, lapack = z[f"{name}__sites"], z[f"{name}__truth"], z[f"{name}__lnf_lapack"]
    with _engine(gpu_engine_cls, c["parent"], c["codes"]) as eng:
        if c["kind"] == "classes":
            _, site = eng.site_classes(c["Q"], c["props"], c["pi"], c["bl"], return_sites=True)
            P_dev = np.stack([eng.export_transition_matrices(k) for k in range(len(c["Q"]))])
            mix = lambda x: _mixture(x, c["props"])
        else:
            _, site = eng.branch(c["Q"], c["pi"], c["bl"], return_sites=True)
            P_dev = eng.export_transition_matrices(0)[None]
            mix = lambda x: np.maximum(x[:, 0], LN_TINY)
    den = np.maximum(np.abs(truth), 1e-3)            # an all-missing column has ln f = 0
    err_gpu = np.abs(site[sites] - truth) / den
    err_ref = np.abs(lapack - truth) / den
    bound = eigen_route_bound(oracle, P_dev, c["pi"], c["ts"], np.ascontiguousarray(np.asarray(c["codes"])[:, sites]), c["leaf"], mix)
    print(f"{name}: device max {err_gpu.max():.2e}, LAPACK route max {err_ref.max():.2e}, largest per-site bound {bound.max():.2e}")
    assert np.all(np.abs(site[sites] - truth) <= SITE_RTOL * den + bound)
    assert np.all(err_gpu[err_ref <= 0.1 * SITE_RTOL] <= SITE_RTOL)
