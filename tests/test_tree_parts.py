"""Tree parts: a single evaluation with few work items per resident workgroup runs subtree programs first and a root
program second (crab_last_tree_parts).  Needs an MI355X: -m gpu.  The gate is bit equality with the uncut walk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(Engine, parent, codes, **kw):
    eng = Engine(0)
    eng.set_tree(parent, np.arange(codes.shape[0]))
    eng.set_alignment(codes, n_states=61, **kw)
    return eng


@pytest.fixture(autouse=True)
def _no_clade_tables(monkeypatch):
    monkeypatch.setenv("CRAB_NO_CLADE_TABLES", "1")   # they take precedence over tree parts (tests/test_clade_tables.py)


def _both(monkeypatch, fn):
    monkeypatch.setenv("CRAB_NO_TREE_PARTS", "1")
    off = fn()
    monkeypatch.setenv("CRAB_NO_TREE_PARTS", "0")
    on = fn()
    return off, on


@pytest.mark.parametrize("model,n_taxa,n_sites,seed", [("M8", 30, 400, 5), ("M8", 100, 1300, 2), ("ModelA", 500, 6250, 3),
                                                       ("free-ratio", 1000, 12500, 4), ("M0", 12, 64, 1)])
def test_cut_and_uncut_walks_agree_bit_for_bit(gpu_engine_cls, monkeypatch, model, n_taxa, n_sites, seed):
    """Gene size, a slice of configs[1], and one GPU's share of configs 3 and 4 over 8 GPUs."""
    from crabml_amd import synth
    case = synth.make_case(model, n_taxa, n_sites, seed=seed)

    def run():
        with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
            if model == "ModelA":
                out = eng.branch_site(case.Q3, case.qfactors[0], case.qfactors[1], case.omega0, case.omega2, case.site_class_freqs,
                                      case.pi, case.branch_lengths, case.branch_labels, return_sites=True)
            elif model == "free-ratio":
                out = eng.branch(case.Q_per_node, case.pi, case.branch_lengths, no_clamp=True, return_sites=True)
            else:
                out = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
            return out, eng.last_tree_parts(), eng.last_work(), eng.last_timing()[1]

    (off, n_off, w_off, t_off), (on, n_on, w_on, t_on) = _both(monkeypatch, run)
    print(f"{model} {n_taxa}x{n_sites}: {n_on} parts; pruning {t_off:.3f} ms uncut, {t_on:.3f} ms cut")
    assert n_off == 0 and n_on >= 2
    assert on[0] == off[0] and np.array_equal(on[1], off[1])
    assert w_on == w_off and np.isfinite(on[0])


def test_cutting_odd_trees_and_the_by_class_table(gpu_engine_cls, monkeypatch):
    """Polytomies, unary nodes, zero-length branches and missing data; the [S, K] table goes through the same programs."""
    from crabml_amd import synth
    base = synth.make_case("M8", 6, 10, seed=77)
    Q, props, pi = base.Q[[0, 6, 10]], np.array([0.5, 0.3, 0.2]), base.pi
    rng = np.random.default_rng(123)
    n_cut = 0
    for trial in range(8):
        n_leaves = int(rng.integers(12, 60))
        parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
        while len(open_nodes) > 1:
            k = int(min(len(open_nodes), rng.choice([1, 2, 2, 2, 3, 4])))
            pick = rng.choice(len(open_nodes), size=k, replace=False)
            for i in sorted(pick, reverse=True):
                parent[open_nodes.pop(int(i))] = nxt
            open_nodes.append(nxt)
            nxt += 1
        par = np.array([parent.get(v, -1) for v in range(nxt)], dtype=np.int32)
        bl = rng.exponential(0.08, size=nxt)
        bl[rng.random(nxt) < 0.1] = 0.0
        bl[par < 0] = 0.0
        S = int(rng.choice([1, 64, 65, 200]))
        anc = rng.integers(0, 61, size=S)
        codes = np.where(rng.random((n_leaves, S)) < 0.7, anc[None, :], rng.integers(0, 61, size=(n_leaves, S))).astype(np.int32)
        codes[rng.random(codes.shape) < 0.05] = 64

        def run():
            with _engine(gpu_engine_cls, par, codes, compress=False) as eng:
                a = eng.site_classes(Q, props, pi, bl, no_clamp=True, return_sites=True)
                n = eng.last_tree_parts()
                return a, eng.site_lnl_by_class(Q, pi, bl, no_clamp=True), n

        (a0, t0, n0), (a1, t1, n1) = _both(monkeypatch, run)
        assert n0 == 0
        n_cut += n1 > 0
        assert np.array_equal(a0[1], a1[1], equal_nan=True) and np.array_equal(t0, t1, equal_nan=True), f"trial {trial}"
        assert a0[0] == a1[0] or (np.isnan(a0[0]) and np.isnan(a1[0]))
    assert n_cut >= 4
