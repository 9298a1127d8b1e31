"""One handle over several GPUs of one process (crab_create_multi), through the C ABI.  Needs an MI355X: -m gpu.

The one-GPU test box stands in for a node with the device list [0, 0, 0]: three shards, each with its own stream and
host thread, on one card.  What must hold is device-count independent: per-site outputs are the single-engine bits
(each pattern is computed by exactly one shard, with the same arithmetic), totals differ from the single engine only
by the order in which pattern blocks are added (1e-7, the slice tolerance of the property tests) and are bit-identical
from run to run.
"""
import os

import numpy as np
import pytest

from util import SITE_RTOL, TOTAL_ATOL, load_case, tree_structure

pytestmark = pytest.mark.gpu

DEVICES = [0, 0, 0]


def _pair(Engine, parent, codes, **kw):
    one, many = Engine(0), Engine(DEVICES)
    for e in (one, many):
        e.set_tree(parent, np.arange(codes.shape[0]))
        e.set_alignment(codes, n_states=61, **kw)
    return one, many


def test_site_class_models_on_three_shards(gpu_engine_cls):
    from crabml_amd import synth
    case = synth.make_case("M8", 40, 3001, seed=12)
    Q, pr, pi, bl = case.Q, case.proportions, case.pi, case.branch_lengths
    one, many = _pair(gpu_engine_cls, case.parent, case.codes)
    with one, many:
        assert many.n_devices == 3 and one.n_devices == 1
        assert many.n_patterns == one.n_patterns == many.n_patterns_total
        a, sa = one.site_classes(Q, pr, pi, bl, return_sites=True)
        b, sb = many.site_classes(Q, pr, pi, bl, return_sites=True)
        b2, sb2 = many.site_classes(Q, pr, pi, bl, return_sites=True)
        assert np.array_equal(sa, sb) and abs(a - b) < 1e-7
        assert b2 == b and np.array_equal(sb, sb2)                       # fixed summation order: bit-identical repeats
        assert np.array_equal(one.site_lnl_by_class(Q, pi, bl), many.site_lnl_by_class(Q, pi, bl))
        # batched vectors (a finite-difference stencil): B sums, each in shard order
        B = 5
        Qb, prb = np.repeat(Q[None], B, 0), np.repeat(np.asarray(pr)[None], B, 0)
        blb = np.repeat(bl[None], B, 0).copy()
        blb[np.arange(1, B), np.arange(B - 1)] *= 1.0 + 1e-6
        got = many.site_classes_batch(Qb, prb, pi, blb)
        want = one.site_classes_batch(Qb, prb, pi, blb)
        assert np.abs(got - want).max() < 1e-7 and got[0] == b and len(set(got.tolist())) == B
        # from parameters; the class posteriors of the adaptive BEB grid (rows written by the owning shard)
        ka, om = np.full(B, case.kappa), np.repeat(case.class_omegas[None], B, 0)
        assert np.abs(many.codon_site_classes(ka, om, prb, pi, blb) - one.codon_site_classes(ka, om, prb, pi, blb)).max() < 1e-7
        w = np.array([0.5, 0.3, 0.2])
        p1 = one.codon_class_posteriors(ka[:3], om[:3], prb[:3], w, pi, bl)
        p3 = many.codon_class_posteriors(ka[:3], om[:3], prb[:3], w, pi, bl)
        assert np.array_equal(p1, p3)
        # introspection adds up (whole walks: with clade tables every shard counts the products of its own clades' sub-patterns)
        assert one.last_clade_tables()[0] > 0 and many.last_clade_tables()[0] > one.last_clade_tables()[0]
        for eng in (one, many):
            eng.set_clade_tables(False)
        assert np.array_equal(one.codon_class_posteriors(ka[:3], om[:3], prb[:3], w, pi, bl), p1)
        assert np.array_equal(many.codon_class_posteriors(ka[:3], om[:3], prb[:3], w, pi, bl), p1)
        d1, g1 = one.last_work()
        d3, g3 = many.last_work()
        assert (d1, g1) == (d3, g3)
        assert np.array_equal(one.export_transition_matrices(3), many.export_transition_matrices(3))
        dev_ms, prune_ms, eig_ms = many.last_timing()
        assert dev_ms > 0 and prune_ms > 0


def test_every_rate_matrix_is_decomposed_once_per_handle(gpu_engine_cls):
    """The shards share one host cache of decompositions: a new parameter vector costs K decompositions per handle,
    not K per device; each shard's device-resident pool still records its own K misses."""
    import time
    c = load_case("lysozyme_m8")
    with gpu_engine_cls(DEVICES) as many:
        many.set_tree(c["parent"], np.arange(c["sequences"].shape[0]))
        many.set_alignment(np.tile(c["sequences"], (1, 3)), n_states=61)   # 390 columns but 86 patterns: two shards, one empty
        lnL = many.site_classes(c["Q"], c["proportions"], c["pi"], c["branch_lengths"])
        hits, misses, resident = many.eig_pool_stats()
        assert misses == 3 * len(c["Q"]) and resident == misses
        assert abs(lnL - 3 * float(c["lnL_pyref"])) < 3 * TOTAL_ATOL
        many.site_classes(c["Q"], c["proportions"], c["pi"], c["branch_lengths"])
        assert many.eig_pool_stats()[1] == misses


def test_branch_site_and_branch_models_on_three_shards(gpu_engine_cls):
    from crabml_amd import synth
    case = synth.make_case("ModelA", 30, 1000, seed=13)
    args = (case.qfactors[0], case.qfactors[1], case.omega0, case.omega2, case.site_class_freqs)
    one, many = _pair(gpu_engine_cls, case.parent, case.codes)
    with one, many:
        a, sa = one.branch_site(case.Q3, *args, case.pi, case.branch_lengths, case.branch_labels, return_sites=True)
        b, sb = many.branch_site(case.Q3, *args, case.pi, case.branch_lengths, case.branch_labels, return_sites=True)
        assert np.array_equal(sa, sb) and abs(a - b) < 1e-7
        pa = one.codon_branch_site(case.kappa, case.omega0, case.omega2, case.p0, case.p1, case.pi, case.branch_lengths, case.branch_labels)
        pb = many.codon_branch_site(case.kappa, case.omega0, case.omega2, case.p0, case.p1, case.pi, case.branch_lengths, case.branch_labels)
        assert abs(pa - pb) < 1e-7
        with pytest.raises(ValueError, match="Branch labels length"):
            many.branch_site(case.Q3, *args, case.pi, case.branch_lengths, case.branch_labels[:-1])
    fr = synth.make_case("free-ratio", 25, 900, seed=14)
    one, many = _pair(gpu_engine_cls, fr.parent, fr.codes)
    with one, many:
        a, sa = one.branch(fr.Q_per_node, fr.pi, fr.branch_lengths, return_sites=True)
        b, sb = many.branch(fr.Q_per_node, fr.pi, fr.branch_lengths, return_sites=True)
        assert np.array_equal(sa, sb) and abs(a - b) < 1e-7
        assert abs(many.codon_branch(fr.kappa, fr.omega_per_node, fr.pi, fr.branch_lengths) -
                   one.codon_branch(fr.kappa, fr.omega_per_node, fr.pi, fr.branch_lengths)) < 1e-7


def test_beb_grid_on_three_shards(gpu_engine_cls):
    """f(X | grid point) is a product over all sites: the handle adds the shards' per-grid-point sums before the
    posteriors are integrated."""
    from crabml_amd import beb
    c = load_case("lysozyme_m2a_beb")
    cols, props = beb.paml_grid_m2a()
    one, many = _pair(gpu_engine_cls, c["parent"], c["sequences"])
    with one, many:
        p1, l1, f1 = one.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
        p3, l3, f3 = many.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
        again = many.beb_grid(c["Q"], c["pi"], c["branch_lengths"], cols, props)
    np.testing.assert_allclose(l3, l1, rtol=1e-12)
    assert abs(f3 - f1) < 1e-9 and np.abs(p3 - p1).max() <= 1e-12
    assert np.array_equal(again[0], p3) and again[2] == f3
    assert np.abs(p3 - c["posteriors_ref"]).max() <= 1e-11


def test_unchanged_shim_uses_the_listed_devices(gpu_engine_cls, monkeypatch):
    """CRAB_DEVICES=0,0: the seven crabml_rust functions, called as crabML calls them, run on a multi-device handle."""
    from crabml_amd import crabml_rust as shim
    c = load_case("lysozyme_m8")
    ts = tree_structure(c["parent"])
    names = [str(x) for x in c["leaf_names"]]
    leaf = list(range(c["sequences"].shape[0]))
    args = dict(q_matrices=list(c["Q"]), proportions=list(c["proportions"]), pi=c["pi"], tree_structure=ts,
                branch_lengths=c["branch_lengths"].tolist(), leaf_names=names, sequences=c["sequences"], leaf_node_ids=leaf)
    shim.clear_cache()
    monkeypatch.delenv("CRAB_DEVICES", raising=False)
    single = shim.compute_site_class_log_likelihood(**args)
    monkeypatch.setenv("CRAB_DEVICES", "0,0")
    multi = shim.compute_site_class_log_likelihood(**args)
    bc = shim.compute_site_log_likelihoods_by_class(q_matrices=list(c["Q"]), pi=c["pi"], tree_structure=ts,
                                                    branch_lengths=c["branch_lengths"].tolist(), leaf_names=names,
                                                    sequences=c["sequences"], leaf_node_ids=leaf)
    P = shim.matrix_exponential(c["Q"][0], c["pi"], 0.1)
    monkeypatch.delenv("CRAB_DEVICES")
    bc1 = shim.compute_site_log_likelihoods_by_class(q_matrices=list(c["Q"]), pi=c["pi"], tree_structure=ts,
                                                     branch_lengths=c["branch_lengths"].tolist(), leaf_names=names,
                                                     sequences=c["sequences"], leaf_node_ids=leaf)
    shim.clear_cache()
    assert abs(multi - single) < 1e-9 and abs(multi - float(c["lnL_pyref"])) < TOTAL_ATOL
    assert np.array_equal(bc, bc1) and P.shape == (61, 61)
    with pytest.raises(ValueError, match="leaf_node_ids has"):
        shim.compute_site_class_log_likelihood(**{**args, "leaf_node_ids": leaf[:-1]})


def test_multi_device_errors(gpu_engine_cls):
    from crabml_amd import _lib
    with pytest.raises(ValueError, match="out of range"):
        gpu_engine_cls([0, 99])
    with pytest.raises(ValueError):
        gpu_engine_cls([])
    c = load_case("lysozyme_m0")
    with gpu_engine_cls(DEVICES) as many:
        with pytest.raises(_lib.CrabError) as ei:
            many.n_states, many.n_nodes, many.n_sites = 61, len(c["parent"]), 1
            many.site_classes(c["Q"], [1.0], c["pi"], c["branch_lengths"])
        assert ei.value.status == _lib.CRAB_ERR_STATE and "shard" in str(ei.value)
