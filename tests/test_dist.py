"""The N > 1 path on CPUs: two gloo ranks shard the site patterns, evaluate their slice (here with the oracle
standing in for the per-rank GPU engine -- the checker, used as a checker) and exchange one double each."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from util import DATA, leaf_ids, load_case, tree_structure

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pattern_slices_cover_and_align():
    from crabml_amd.dist import pattern_slices
    for n, ws in [(9947, 8), (86, 2), (0, 4), (128, 3), (100000, 8), (1, 1)]:
        sl = pattern_slices(n, ws, 128)
        assert len(sl) == ws and sl[0][0] == 0 and sl[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(sl[:-1], sl[1:]))
        assert all(b % 128 == 0 or b == n for b, _ in sl)
    with pytest.raises(ValueError):
        pattern_slices(10, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crabml_amd.dist import ShardedLikelihood, pattern_slices
        from crabml_amd.engine import compress_patterns
        from oracle import oracle as orc
        c = load_case("hiv_m8a")
        seq = c["sequences"]
        s2p, w, first = compress_patterns(seq)
        b, e = pattern_slices(len(w), world, align=16)[rank]
        cols = first[b:e]

        def evaluate(Q, props, pi, bl):
            if len(cols) == 0:
                return 0.0
            _, site = orc.compute_site_class_log_likelihood(list(Q), props, pi, tree_structure(c["parent"]), bl, None,
                                                            np.ascontiguousarray(seq[:, cols]), leaf_ids(c), threads=1,
                                                            return_sites=True)
            return float((site * w[b:e]).sum())

        total = ShardedLikelihood(evaluate)(c["Q"], c["proportions"], c["pi"], c["branch_lengths"])
        out[rank] = total
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sum_equals_whole(oracle):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        totals = [out[r] for r in range(world)]
    c = load_case("hiv_m8a")
    whole = oracle.compute_site_class_log_likelihood(list(c["Q"]), c["proportions"], c["pi"], tree_structure(c["parent"]),
                                                     c["branch_lengths"], None, c["sequences"], leaf_ids(c))
    assert totals[0] == totals[1]                  # rank-ordered sum: identical on every rank
    assert abs(totals[0] - whole) < 1e-9


def test_gene_sharding_and_worker_pool():
    """Config 5 shards by gene: every gene lands on exactly one rank; results come back in order whatever the
    completion order of the workers."""
    import threading
    import time
    from crabml_amd.dist import gene_shard, map_genes
    for n, ws in [(10, 3), (7, 8), (0, 2), (64, 8)]:
        shards = [gene_shard(n, r, ws) for r in range(ws)]
        assert sorted(g for s in shards for g in s) == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    with pytest.raises(ValueError):
        gene_shard(4, 2, 2)

    made, closed, seen = [], [], set()
    lock = threading.Lock()

    class FakeEngine:
        def __init__(self):
            made.append(self)
            self.busy = False

        def close(self):
            closed.append(self)

    def fn(e, g):
        assert not e.busy                      # an engine is never shared by two genes at once
        e.busy = True
        time.sleep(0.002 * (5 - g % 5))        # later genes finish first
        with lock:
            seen.add(threading.get_ident())
        e.busy = False
        return g * g

    out = map_genes(fn, range(23), workers=4, make_engine=FakeEngine)
    assert out == [g * g for g in range(23)]
    assert len(made) == 4 and len(closed) == 4 and 1 < len(seen) <= 4
    assert map_genes(fn, [], workers=4, make_engine=FakeEngine) == []


def _table_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crabml_amd.dist import gather_site_table
        rng = np.random.default_rng(1)
        full = rng.normal(size=(50, 4))
        owner = rng.integers(0, world, size=50)           # which rank's pattern slice holds each site
        local = np.where((owner == rank)[:, None], full, np.nan)
        got = gather_site_table(local)
        out[rank] = bool(np.array_equal(got, full))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_of_the_site_by_class_table():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_table_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert all(out[r] for r in range(world))


def _beb_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crabml_amd import beb
        from crabml_amd.dist import sharded_beb_grid
        from oracle import beb as obeb
        c = load_case("lysozyme_m2a_beb")
        cols, props = beb.paml_grid_m2a()
        pick = np.arange(0, 10000, 53)
        cols, props = cols[pick], props[pick]
        table = c["site_log_liks"]
        mine = np.arange(table.shape[0]) % world == rank          # this rank's sites

        class SliceEngine:   # the two-call protocol of Engine.beb_grid, with the oracle standing in for the GPU
            def beb_grid(self, Q, pi, bl, cols_, props_, partial=False, lnfx=None):
                if partial:
                    return None, obeb.site_posteriors(table[mine], cols_, props_)[1], 0.0
                F = obeb.scale_table(table)
                fx = np.log(np.exp(lnfx - lnfx.max()).sum()) + lnfx.max()
                gw = np.exp(lnfx - fx)
                post = np.full((table.shape[0], cols_.shape[1]), np.nan)
                terms = props_[None, :, :] * F[:, cols_]                       # [S, G, n_mix]
                post[mine] = ((terms / terms.sum(axis=2, keepdims=True)) * gw[None, :, None]).sum(axis=1)[mine]
                return post, lnfx, fx

        post, ln_fx = sharded_beb_grid(SliceEngine(), None, None, None, cols, props)
        whole, _, whole_fx = obeb.site_posteriors(table, cols, props)
        out[rank] = (bool(np.allclose(post[mine], whole[mine], rtol=1e-11, atol=1e-300)), bool(np.isnan(post[~mine]).all()),
                     abs(ln_fx - whole_fx) < 1e-10, float(ln_fx))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_beb_grid_over_pattern_slices():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_beb_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert all(r[0] and r[1] and r[2] for r in res)
    assert res[0][3] == res[1][3]          # rank-ordered sum: the same ln f(X) bits on every rank


class _OracleEngine:
    """The calls the optimiser mirror makes on an Engine, answered by the CPU oracle (CPU tests only)."""

    def __init__(self, *_):
        self.closed = False

    def set_tree_structure(self, tree_structure, leaf_node_ids):
        self.ts, self.leaf = list(tree_structure), list(leaf_node_ids)

    def set_alignment(self, codes, pattern_begin=0, pattern_end=-1, **_):
        from crabml_amd.engine import compress_patterns
        codes = np.ascontiguousarray(codes, dtype=np.int32)
        _, w, first = compress_patterns(codes)                       # the engine's own pattern order
        end = len(w) if pattern_end < 0 else pattern_end
        self.codes, self.weights = np.ascontiguousarray(codes[:, first[pattern_begin:end]]), w[pattern_begin:end]

    def codon_site_classes(self, kappa, omegas, proportions, pi, branch_lengths, no_clamp=False):
        from crabml_amd import models
        from oracle import oracle as orc
        out = []
        for k, om, pr, bl in zip(kappa, omegas, proportions, branch_lengths):
            raw = [models.build_codon_Q_matrix(float(k), float(w), pi, 1.0) for w in om]
            scale = sum(p * models.mean_rate(Q, pi) for p, Q in zip(pr, raw))
            if self.codes.shape[1] == 0:
                out.append(0.0)
                continue
            site = orc.compute_site_class_log_likelihood([Q / scale for Q in raw], pr, pi, self.ts, bl, None, self.codes, self.leaf,
                                                         threads=1, return_sites=True)[1]
            out.append(float(self.weights @ site))
        return np.array(out)

    def close(self):
        self.closed = True


def _scan_genes():
    from crabml_amd.io import Alignment, Tree
    from test_optimize import LYSOZYME_START
    aln = Alignment.from_phylip(os.path.join(DATA, "lysozymeSmall.txt"), seqtype="codon")
    codes = np.asarray(aln.sequences)
    return [(Alignment.from_codes(aln.names, codes[:, lo:lo + 30]), Tree.from_newick(LYSOZYME_START)) for lo in (0, 30, 60, 90)]


def _scan_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crabml_amd import scan
        rows = scan.scan(_scan_genes(), null="M7", rank=rank, world_size=world, workers=2, make_engine=_OracleEngine, ncatG=2, maxiter=2, init_with_m0=False)
        out[rank] = [(r["lnL_null"], r["lnL_M8"], r["LRT"], r["pvalue"]) for r in rows]
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gene_scan_equals_the_single_rank_scan():
    """BASELINE config 5's shape in small: genes dealt round-robin to ranks, M7 and M8 fitted per gene through the optimiser
    mirror (two iterations, two beta classes: the fit itself is tested on the device), rows gathered on every rank."""
    from crabml_amd import scan
    import warnings
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_scan_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        alone = [(r["lnL_null"], r["lnL_M8"], r["LRT"], r["pvalue"])
                 for r in scan.scan(_scan_genes(), null="M7", workers=1, make_engine=_OracleEngine, ncatG=2, maxiter=2, init_with_m0=False)]
    assert res[0] == res[1] == alone and len(alone) == 4
    from crabml_amd.io import Alignment
    bad = _scan_genes()[:2]
    bad[1] = (Alignment.from_codes(["x%d" % i for i in range(7)], np.asarray(bad[1][0].sequences)), bad[1][1])   # names not in the tree
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rows = scan.scan(bad, workers=1, make_engine=_OracleEngine, ncatG=2, maxiter=1, init_with_m0=False)
    assert "LRT" in rows[0] and rows[1]["error"].startswith("ValueError: Alignment and tree have different species")
    for l7, l8, stat, p in alone:
        assert np.isfinite(l7) and np.isfinite(l8) and stat >= 0 and 0 <= p <= 1
        assert stat == max(0.0, 2 * (l8 - l7))


def test_lrt_statistics_follow_the_reference():
    """analysis/lrt.py:8-138: chi-square tail, the boundary mixture's halved tail, and the negative-statistic guard."""
    from scipy.stats import chi2
    from crabml_amd.scan import calculate_lrt, calculate_lrt_mixture
    assert calculate_lrt(-902.510018, -899.999237, 2) == (2 * (-899.999237 + 902.510018), chi2.sf(2 * (-899.999237 + 902.510018), 2))
    stat, p = calculate_lrt_mixture(-902.503869, -899.999237, 1)
    assert p == 0.5 * chi2.sf(stat, 1) and abs(stat - 5.009264) < 1e-6
    assert calculate_lrt(-10.0, -10.0, 2) == (0.0, 1.0) and calculate_lrt_mixture(-10.0, -10.0, 1) == (0.0, 1.0)
    with pytest.warns(UserWarning, match="Negative LRT"):
        assert calculate_lrt(-10.0, -10.5, 2) == (0.0, 1.0)
    with pytest.warns(UserWarning, match="Negative LRT"):
        assert calculate_lrt_mixture(-10.0, -10.5, 1) == (0.0, 1.0)


def _fit_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crabml_amd import synth
        from crabml_amd.optimize import M2aOptimizer
        aln, tree = synth.as_alignment_and_tree(synth.make_case("M8", 5, 400, seed=11))
        opt = M2aOptimizer(aln, tree, engine=_OracleEngine(), rank=rank, world_size=world)
        res = opt.optimize(maxiter=2)
        out[rank] = (res, opt.result.x.tolist(), opt.calc.engine.codes.shape[1], opt.n_calls)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_fit_over_pattern_slices_equals_the_single_rank_fit():
    """A fit with the site patterns sharded over ranks: every likelihood of every stencil is the rank-ordered sum of
    partial sums (one all-gather of B doubles per gradient, north_star's "lnL and finite-difference gradient components"),
    so both ranks hold identical iterates, and they are the single-rank fit's up to the summation order."""
    from crabml_amd import synth
    from crabml_amd.optimize import M2aOptimizer
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fit_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][3] == res[1][3]      # same bits on both ranks
    assert res[0][2] % 128 == 0 and res[0][2] > 0 and res[1][2] > 0                                               # both ranks held patterns
    aln, tree = synth.as_alignment_and_tree(synth.make_case("M8", 5, 400, seed=11))
    alone = M2aOptimizer(aln, tree, engine=_OracleEngine())
    want = alone.optimize(maxiter=2)
    assert alone.calc.engine.codes.shape[1] == res[0][2] + res[1][2]
    # a different summation order moves f by ~1e-13, the 1e-8-step gradient by ~1e-5, and the iterates with it
    np.testing.assert_allclose(res[0][0], want, rtol=1e-3)
    assert abs(res[0][0][-1] - want[-1]) < 1e-6 * abs(want[-1])


def _gather_worker(rank, world, port, out, backend):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":
        import torch
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from crabml_amd import dist as cd
        rows = []
        for n in (1, 5, 1, 203):        # sizes repeat: the second use of a size runs on the buffers of the first
            mine = np.arange(n, dtype=np.float64) * 0.1 + 1000.0 * (rank + 1) + n
            rows.append(cd.all_gather_doubles(mine))
        # one rank: the public helpers short-cut, so the collective is exercised through all_gather_doubles above
        s = cd.ordered_sum(0.1 * (rank + 1))
        v = cd.ordered_sum_vector(np.array([[1.0, 2.0], [3.0, 4.0]]) * (rank + 1))
        out[rank] = ([r.tolist() for r in rows], s, v.tolist(), len(cd._GATHER_BUFFERS))
    finally:
        dist.destroy_process_group()


def _check_gather(out, world):
    for rank in range(world):
        rows, s, v, n_buf = out[rank]
        for n, table in zip((1, 5, 1, 203), rows):
            want = [(np.arange(n) * 0.1 + 1000.0 * (r + 1) + n).tolist() for r in range(world)]
            assert table == want
        acc = 0.0
        for r in range(world):
            acc += 0.1 * (r + 1)
        assert s == acc
        assert v == (np.array([[1.0, 2.0], [3.0, 4.0]]) * sum(range(1, world + 1))).tolist()
        assert n_buf <= 4                   # one set of buffers per payload size (1, 5, 203, and the 2 x 2 vector if world > 1)


def test_two_rank_gloo_all_gather_of_doubles_reuses_its_buffers():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gather_worker, args=(world, _free_port(), out, "gloo"), nprocs=world, join=True)
        _check_gather(dict(out), world)


@pytest.mark.gpu
def test_rccl_all_gather_of_doubles_on_one_rank():
    """The RCCL leg of all_gather_doubles (pinned mirrors, device buffers, all_gather_into_tensor) on the box's one
    GPU: a one-rank communicator is the most this box can hold (two ranks on one card are refused by RCCL)."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gather_worker, args=(1, _free_port(), out, "nccl"), nprocs=1, join=True)
        _check_gather(dict(out), 1)
