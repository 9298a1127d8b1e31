"""Partial updates (second half): in a batch, a vector that moves ONE branch re-walks only the path from that
branch to the root, on tiles the first vector's walk left in HBM.  Needs an MI355X: -m gpu.

The gate is bit equality: every element of a batch -- whichever way it was computed -- must equal the evaluation of its
vector alone (a whole walk), because the path programs execute the same products in the same order
(tests/test_host_logic.py checks the programs themselves on the CPU).
"""
import os

import numpy as np
import pytest

from util import load_case

pytestmark = pytest.mark.gpu


def _engine(Engine, parent, codes, device=0, **kw):
    eng = Engine(device)
    eng.set_tree(parent, np.arange(codes.shape[0]))
    eng.set_alignment(codes, n_states=61, **kw)
    return eng


def _random_tree(rng, n_leaves):
    parent, open_nodes, nxt = {}, list(range(n_leaves)), n_leaves
    while len(open_nodes) > 1:
        k = int(min(len(open_nodes), rng.choice([2, 2, 2, 3, 4, 1])))
        pick = rng.choice(len(open_nodes), size=k, replace=False)
        for i in sorted(pick, reverse=True):
            parent[open_nodes.pop(int(i))] = nxt
        open_nodes.append(nxt)
        nxt += 1
    return np.array([parent.get(v, -1) for v in range(nxt)], dtype=np.int32)


@pytest.mark.parametrize("n_taxa,n_sites,seed", [(14, 150, 8), (100, 700, 2), (37, 65, 5)])
def test_gradient_stencil_equals_single_evaluations_bit_for_bit(gpu_engine_cls, n_taxa, n_sites, seed):
    """The forward-difference stencil of an M8 fit: the centre, one row per branch length, rows for kappa, an omega and
    a proportion; site-class models by parameters and by rate matrices passed by value."""
    from crabml_amd import synth
    case = synth.make_case("M8", n_taxa, n_sites, seed=seed)
    nn, K = case.n_nodes, len(case.proportions)
    om, pr = case.class_omegas, np.asarray(case.proportions)
    B = 1 + (nn - 1) + 3
    bl = np.repeat(case.branch_lengths[None], B, 0)
    ka = np.full(B, case.kappa)
    oms, prs = np.repeat(om[None], B, 0), np.repeat(pr[None], B, 0)
    for v in range(nn - 1):
        bl[1 + v, v] += 1e-8 * max(1.0, bl[1 + v, v])
    ka[nn] *= 1.0 + 1e-8
    oms[nn + 1, 3] *= 1.0 + 1e-8
    prs[nn + 2] = pr * np.where(np.arange(K) == 2, 1.001, 1.0)
    prs[nn + 2] /= prs[nn + 2].sum()
    with _engine(gpu_engine_cls, case.parent, case.codes) as eng:
        got = eng.codon_site_classes(ka, oms, prs, case.pi, bl)
        n_short = eng.last_partial_updates()
        dense, gather = eng.last_work()
        want = np.array([eng.codon_site_classes(ka[b], oms[b], prs[b], case.pi, bl[b]) for b in range(B)])
        assert eng.last_partial_updates() == 0
        again = eng.codon_site_classes(ka, oms, prs, case.pi, bl)
        Q = np.repeat(case.Q[None], B, 0)
        got_q = eng.site_classes_batch(Q, np.repeat(pr[None], B, 0), case.pi, bl)
        n_short_q = eng.last_partial_updates()
        want_q = np.array([eng.site_classes(case.Q, pr, case.pi, bl[b]) for b in range(B)])
        n_pat = eng.n_patterns
    # the proportion row changes the shared scale of the mixture, hence every branch's effective length: a whole walk;
    # by value the last three rows repeat the centre's matrices on every branch (no branch moved): whole walks too
    assert n_short == nn - 1 and n_short_q == nn - 1
    assert np.array_equal(got, want) and np.array_equal(got, again)
    assert np.array_equal(got_q, want_q)
    assert len(set(got[: nn].tolist())) == nn                     # every branch perturbation is visible
    whole_walks = (B - n_short) * K * (n_taxa - 3)
    assert whole_walks * n_pat < dense < B * K * (n_taxa - 3) * n_pat / 2    # far fewer products than B whole walks


def test_partial_updates_on_odd_tree_shapes(gpu_engine_cls):
    """Polytomies, unary nodes, leaves hanging off the root, zero-length branches, missing data; pattern counts around
    the tile boundary; every branch moved in turn."""
    from crabml_amd import synth
    base = synth.make_case("M8", 6, 10, seed=77)
    Q, props, pi = base.Q[[0, 6, 10]], np.array([0.5, 0.3, 0.2]), base.pi
    rng = np.random.default_rng(99 + int(os.environ.get("CRAB_SWEEP_SEED", "0")))
    for trial in range(10):
        n_leaves = int(rng.integers(3, 30))
        par = _random_tree(rng, n_leaves)
        nn = len(par)
        t = rng.exponential(0.08, size=nn)
        t[rng.random(nn) < 0.1] = 0.0
        t[par < 0] = 0.0
        S = int(rng.choice([1, 63, 64, 65, 130]))
        anc = rng.integers(0, 61, size=S)
        codes = np.where(rng.random((n_leaves, S)) < 0.7, anc[None, :], rng.integers(0, 61, size=(n_leaves, S))).astype(np.int32)
        codes[rng.random(codes.shape) < 0.05] = 64
        movable = [v for v in range(nn) if par[v] >= 0]
        B = 1 + len(movable)
        bl = np.repeat(t[None], B, 0)
        for i, v in enumerate(movable):
            bl[1 + i, v] = bl[1 + i, v] * (1 + 1e-6) + 1e-7
        Qb, prb = np.repeat(Q[None], B, 0), np.repeat(props[None], B, 0)
        with _engine(gpu_engine_cls, par, codes, compress=False) as eng:
            got = eng.site_classes_batch(Qb, prb, pi, bl, no_clamp=True)
            n_short = eng.last_partial_updates()
            want = np.array([eng.site_classes(Q, props, pi, bl[b], no_clamp=True) for b in range(B)])
        assert n_short == (len(movable) if len(movable) >= 4 else 0), f"trial {trial}"
        assert np.array_equal(got, want, equal_nan=True), f"trial {trial}: {n_leaves} leaves, {nn} nodes, {S} sites"


def test_partial_updates_for_branch_site_and_free_ratio(gpu_engine_cls):
    from crabml_amd import synth
    a = synth.make_case("ModelA", 40, 500, seed=3)
    nn = a.n_nodes
    B = nn + 2
    bl = np.repeat(a.branch_lengths[None], B, 0)
    bl[np.arange(1, nn), np.arange(nn - 1)] *= 1.0 + 1e-7
    w2 = np.full(B, a.omega2)
    w2[nn] *= 1.0 + 1e-7
    p0 = np.full(B, a.p0)
    p0[nn + 1] *= 1.0 + 1e-7
    full = lambda x: np.full(B, x)     # noqa: E731
    with _engine(gpu_engine_cls, a.parent, a.codes) as eng:
        got = eng.codon_branch_site(full(a.kappa), full(a.omega0), w2, p0, full(a.p1), a.pi, bl, a.branch_labels)
        n_short = eng.last_partial_updates()
        want = np.array([eng.codon_branch_site(a.kappa, a.omega0, w2[b], p0[b], a.p1, a.pi, bl[b], a.branch_labels) for b in range(B)])
    assert n_short == nn - 1 and np.array_equal(got, want)
    # free-ratio: a branch's omega moves one matrix too (its own normalisation folds into that branch's length)
    f = synth.make_case("free-ratio", 30, 400, seed=4)
    nn = f.n_nodes
    B = 2 * (nn - 1) + 2
    om = np.repeat(f.omega_per_node[None], B, 0)
    bl = np.repeat(f.branch_lengths[None], B, 0)
    ka = np.full(B, f.kappa)
    for v in range(nn - 1):
        om[1 + v, v] *= 1.0 + 1e-7
        bl[nn + v, v] *= 1.0 + 1e-7
    ka[B - 1] *= 1.0 + 1e-7
    with _engine(gpu_engine_cls, f.parent, f.codes) as eng:
        got = eng.codon_branch(ka, om, f.pi, bl, no_clamp=True)
        n_short = eng.last_partial_updates()
        want = np.array([eng.codon_branch(ka[b], om[b], f.pi, bl[b], no_clamp=True) for b in range(B)])
    assert n_short == 2 * (nn - 1) and np.array_equal(got, want)
    assert len(set(got.tolist())) == B


def test_partial_updates_on_a_multi_device_handle_and_switched_off(gpu_engine_cls, monkeypatch):
    c = load_case("lysozymeLarge_modelA")
    from crabml_amd import synth
    case = synth.make_case("M8", 25, 300, seed=11)
    nn = case.n_nodes
    B = nn
    bl = np.repeat(case.branch_lengths[None], B, 0)
    bl[np.arange(1, nn), np.arange(nn - 1)] *= 1.0 + 1e-7
    Qb = np.repeat(case.Q[None], B, 0)
    prb = np.repeat(np.asarray(case.proportions)[None], B, 0)
    with _engine(gpu_engine_cls, case.parent, case.codes) as one, _engine(gpu_engine_cls, case.parent, case.codes, device=[0, 0]) as two:
        a = one.site_classes_batch(Qb, prb, case.pi, bl)
        assert one.last_partial_updates() == nn - 1
        b = two.site_classes_batch(Qb, prb, case.pi, bl)
        assert two.last_partial_updates() == nn - 1
        monkeypatch.setenv("CRAB_NO_PATH_UPDATES", "1")
        off = one.site_classes_batch(Qb, prb, case.pi, bl)
        assert one.last_partial_updates() == 0
        monkeypatch.setenv("CRAB_NO_PATH_UPDATES", "0")
        monkeypatch.setenv("CRAB_PATH_SAVE_GB", "0.000001")       # no room for the save area: whole walks, same bits
        small = one.site_classes_batch(Qb, prb, case.pi, bl)
        assert one.last_partial_updates() == 0
    assert np.array_equal(a, off) and np.array_equal(a, small) and np.abs(a - b).max() < 1e-7
