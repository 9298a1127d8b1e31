"""Clade tables (subtree site repeats) against whole walks: bit-identical results, fewer products.  Needs an MI355X: -m gpu."""
import numpy as np
import pytest

from util import SITE_RTOL, tree_structure

pytestmark = pytest.mark.gpu


def _engine(Engine, case, n_states=61, device=0):
    eng = Engine(device)
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=n_states)
    return eng


@pytest.mark.parametrize("model,taxa,sites,seed", [("M8", 100, 10000, 2), ("M8", 30, 2000, 11), ("M8", 14, 600, 5), ("M8", 257, 3000, 8)])
def test_site_classes_with_and_without_tables(gpu_engine_cls, model, taxa, sites, seed):
    from crabml_amd import synth
    case = synth.make_case(model, taxa, sites, seed=seed)
    with _engine(gpu_engine_cls, case) as eng:
        on, site_on = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        n_cl, u = eng.last_clade_tables()
        dense_on = eng.last_work()[0]
        by_on = eng.site_lnl_by_class(case.Q, case.pi, case.branch_lengths)
        par_on = eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
        again = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
        eng.set_clade_tables(False)
        off, site_off = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        assert eng.last_clade_tables() == (0, 0)
        dense_off = eng.last_work()[0]
        by_off = eng.site_lnl_by_class(case.Q, case.pi, case.branch_lengths)
        par_off = eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
        eng.set_clade_tables(True)
        back = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
    print(f"{taxa} x {sites}: {n_cl} clades, {u} sub-patterns in all; products {dense_off} -> {dense_on} ({dense_on / dense_off:.2f})")
    assert n_cl > 0 and dense_on < dense_off
    assert on == off == again == back and par_on == par_off
    assert np.array_equal(site_on, site_off) and np.array_equal(by_on, by_off)


def test_other_models_and_state_spaces(gpu_engine_cls, oracle):
    from crabml_amd import synth
    # free-ratio (one eigensystem per branch) and Model A (four classes, per-branch choice of eigensystem)
    case = synth.make_case("free-ratio", 60, 3000, seed=4)
    with _engine(gpu_engine_cls, case) as eng:
        on = eng.codon_branch(case.kappa, case.omega_per_node, case.pi, case.branch_lengths)
        assert eng.last_clade_tables()[0] > 0
        eng.set_clade_tables(False)
        assert eng.codon_branch(case.kappa, case.omega_per_node, case.pi, case.branch_lengths) == on
    case = synth.make_case("ModelA", 60, 3000, seed=3)
    with _engine(gpu_engine_cls, case) as eng:
        args = (case.kappa, case.omega0, case.omega2, case.p0, case.p1, case.pi, case.branch_lengths, case.branch_labels)
        on = eng.codon_branch_site(*args)
        assert eng.last_clade_tables()[0] > 0
        eng.set_clade_tables(False)
        assert eng.codon_branch_site(*args) == on
    # 20 states, a tree with a multifurcation and a unary node, against the oracle
    rng = np.random.default_rng(3)
    n = 20
    pi = rng.dirichlet(3.0 * np.ones(n))
    Sx = rng.gamma(1.0, 1.0, size=(n, n))
    Sx = np.triu(Sx, 1) + np.triu(Sx, 1).T
    Q = Sx * pi[None, :]
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q /= -(pi * np.diag(Q)).sum()
    #            leaves 0..8                      9        10       11      12 (unary)  13 root
    parent = np.array([9, 9, 9, 10, 10, 11, 11, 12, 13, 13, 11, 13, 13, -1], dtype=np.int32)
    bl = rng.uniform(0.01, 0.1, size=14)
    bl[13] = 0.0
    base = rng.integers(0, n, size=700)
    codes = np.stack([np.where(rng.random(700) < 0.07, rng.integers(0, n, size=700), base) for _ in range(9)]).astype(np.int32)
    eng = gpu_engine_cls(0)
    try:
        eng.set_tree(parent, np.arange(9))
        eng.set_alignment(codes, n_states=n)
        on, site_on = eng.site_classes(Q[None], [1.0], pi, bl, return_sites=True)
        used = eng.last_clade_tables()
        eng.set_clade_tables(False)
        off, site_off = eng.site_classes(Q[None], [1.0], pi, bl, return_sites=True)
    finally:
        eng.close()
    ref, ref_site = oracle.compute_site_class_log_likelihood([Q], [1.0], pi, tree_structure(parent), bl, None, codes, list(range(9)), return_sites=True)
    assert used[0] > 0 and on == off and np.array_equal(site_on, site_off)
    assert abs(on - ref) < 1e-6 and np.max(np.abs(site_on - ref_site) / np.abs(ref_site)) <= SITE_RTOL


def test_batches_and_shards(gpu_engine_cls):
    """A batched call keeps its partial updates (no tables), a single call after it uses tables again; every shard of a
    multi-device handle builds tables for its own patterns."""
    from crabml_amd import synth
    case = synth.make_case("M8", 40, 4000, seed=21)
    nn = case.n_nodes
    B = 8
    bl = np.repeat(case.branch_lengths[None], B, 0)
    bl[np.arange(1, B), np.arange(B - 1)] *= 1.0 + 1e-6
    ka, om, pr = np.full(B, case.kappa), np.repeat(case.class_omegas[None], B, 0), np.repeat(np.asarray(case.proportions)[None], B, 0)
    with _engine(gpu_engine_cls, case) as eng:
        batch = eng.codon_site_classes(ka, om, pr, case.pi, bl)
        assert eng.last_clade_tables() == (0, 0) and eng.last_partial_updates() == B - 1
        singles = np.array([eng.codon_site_classes(ka[b], om[b], pr[b], case.pi, bl[b]) for b in range(B)])
        assert eng.last_clade_tables()[0] > 0
        assert np.array_equal(batch, singles)
        one = singles[0]
        # a whole gradient stencil: branch rows take path programs on the base vector's tiles (whole walk), the rows that move a
        # model parameter walk the whole tree -- through clade tables of their own
        B2 = nn + 3
        bl2 = np.repeat(case.branch_lengths[None], B2, 0)
        bl2[np.arange(1, nn), np.arange(nn - 1)] *= 1.0 + 1e-6
        ka2, om2, pr2 = np.full(B2, case.kappa), np.repeat(case.class_omegas[None], B2, 0), np.repeat(np.asarray(case.proportions)[None], B2, 0)
        ka2[nn] *= 1.0 + 1e-6
        om2[nn + 1, 2] *= 1.0 + 1e-6
        om2[nn + 2, 10] *= 1.0 + 1e-6
        grad = eng.codon_site_classes(ka2, om2, pr2, case.pi, bl2)
        assert eng.last_partial_updates() == nn - 1 and eng.last_clade_tables()[0] > 0
        dense_mixed = eng.last_work()[0]
        eng.set_clade_tables(False)
        grad_off = eng.codon_site_classes(ka2, om2, pr2, case.pi, bl2)
        assert eng.last_work()[0] > dense_mixed and np.array_equal(grad, grad_off)
        eng.set_clade_tables(True)
        want = np.array([eng.codon_site_classes(ka2[b], om2[b], pr2[b], case.pi, bl2[b]) for b in range(B2)])
        assert np.array_equal(grad, want)
        # unrelated vectors (no single-branch structure): every one of them through tables
        rng = np.random.default_rng(4)
        B3 = 5
        ka3 = case.kappa * rng.uniform(0.8, 1.2, B3)
        bl3 = case.branch_lengths[None] * rng.uniform(0.7, 1.3, (B3, nn))
        got3 = eng.codon_site_classes(ka3, om2[:B3], pr2[:B3], case.pi, bl3)
        assert eng.last_partial_updates() == 0 and eng.last_clade_tables()[0] > 0
        want3 = np.array([eng.codon_site_classes(ka3[b], om2[b], pr2[b], case.pi, bl3[b]) for b in range(B3)])
        eng.set_clade_tables(False)
        off3 = eng.codon_site_classes(ka3, om2[:B3], pr2[:B3], case.pi, bl3)
        assert np.array_equal(got3, want3) and np.array_equal(got3, off3)
    eng = gpu_engine_cls([0, 0, 0])
    try:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes)
        multi = eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
        n_cl, _ = eng.last_clade_tables()
        eng.set_clade_tables(False)
        multi_off = eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
    finally:
        eng.close()
    assert n_cl > 0 and multi == multi_off and abs(multi - one) < 1e-7


@pytest.mark.parametrize("model,taxa,sites,seed", [("M8", 100, 10000, 2), ("ModelA", 120, 2500, 6), ("free-ratio", 60, 3000, 4)])
def test_one_clade_launch_against_one_per_level(gpu_engine_cls, monkeypatch, model, taxa, sites, seed):
    """All nesting levels in one launch (work items wait for the tables they gather from) give the bits of one launch per
    level (CRAB_CLADE_LEVEL_LAUNCHES=1, round 2's scheme), for single and batched evaluations, and of whole walks."""
    from crabml_amd import synth
    case = synth.make_case(model, taxa, sites, seed=seed)

    def evaluate(eng):
        if model == "ModelA":
            return eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi, case.branch_lengths,
                                   case.branch_labels, return_sites=True)
        if model == "free-ratio":
            return eng.branch(case.Q_per_node, case.pi, case.branch_lengths, return_sites=True)
        return eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)

    with _engine(gpu_engine_cls, case) as eng:
        merged, site_merged = evaluate(eng)
        n_merged = eng.last_prune_launches()
        n_cl, _ = eng.last_clade_tables()
        cut = eng.last_tree_parts() > 0      # an under-filled launch of main programs is cut into subtree programs + roots
        again, _ = evaluate(eng)
        if model == "M8":   # a batch: every vector gets tables of its own, all in the one launch
            t2 = np.stack([case.branch_lengths, case.branch_lengths * 1.01, case.branch_lengths * 0.99])
            batch_merged = eng.site_classes_batch(np.stack([case.Q] * 3), np.stack([case.proportions] * 3), case.pi, t2)
    monkeypatch.setenv("CRAB_CLADE_LEVEL_LAUNCHES", "1")
    with _engine(gpu_engine_cls, case) as eng:
        levels, site_levels = evaluate(eng)
        n_levels = eng.last_prune_launches()
        if model == "M8":
            batch_levels = eng.site_classes_batch(np.stack([case.Q] * 3), np.stack([case.proportions] * 3), case.pi, t2)
        eng.set_clade_tables(False)
        whole, site_whole = evaluate(eng)
    print(f"{model} {taxa} x {sites}: {n_cl} clades, launches {n_merged} merged / {n_levels} per level")
    assert n_cl > 0 and n_merged == 2 + cut and n_levels >= n_merged
    assert merged == levels == whole == again
    assert np.array_equal(site_merged, site_levels) and np.array_equal(site_merged, site_whole)
    if model == "M8":
        assert np.array_equal(batch_merged, batch_levels) and batch_merged[0] == merged


def test_missing_table_tiles_fall_back_to_level_launches(gpu_engine_cls, monkeypatch, capfd):
    """The launch that reads the tables of the merged clade launch checks that every tile was announced; if not (an XCD whose
    chunk found no workgroup, a wait that gave up) nothing of that evaluation is used: the engine says so once, goes back to one
    launch per nesting level and evaluates again.  CRAB_DEBUG_SYNC=2 makes the host expect one tile more than there is."""
    from crabml_amd import synth
    case = synth.make_case("M8", 100, 10000, seed=2)   # clades nest three levels deep
    with _engine(gpu_engine_cls, case) as eng:
        good, site_good = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        assert np.isfinite(good) and eng.last_prune_launches() == 2
        monkeypatch.setenv("CRAB_DEBUG_SYNC", "2")
        again, site_again = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        launches = eng.last_prune_launches()
        monkeypatch.delenv("CRAB_DEBUG_SYNC")
        later = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
        assert eng.last_prune_launches() == launches      # for good
    assert "one launch per nesting level" in capfd.readouterr().err
    assert launches > 2 and again == good == later and np.array_equal(site_again, site_good)
