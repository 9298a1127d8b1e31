"""The batched eigensolver on the device (kernel 0, crab_set_eigensolver(e, 1)) against the host solver, the oracle and
the 60-digit fixture.  Needs an MI355X: -m gpu."""
import os
import time

import numpy as np
import pytest

from util import (GOLDEN, SITE_MODEL_CASES, SITE_RTOL, TOTAL_ATOL, eigen_route_bound, leaf_ids, load_case, tree_structure)

pytestmark = pytest.mark.gpu
LN_TINY = np.log(1e-300)


def _engine(Engine, parent, codes, where, n_states=61):
    eng = Engine(0)
    eng.set_eigensolver(where)
    eng.set_tree(parent, np.arange(codes.shape[0]))
    eng.set_alignment(codes, n_states=n_states)
    return eng


def _mixture(lnf, props):
    x = np.maximum(lnf, LN_TINY)
    with np.errstate(divide="ignore"):
        x = x + np.log(props)[None, :]
    mx = x.max(axis=1)
    return mx + np.log(np.exp(x - mx[:, None]).sum(axis=1))


@pytest.mark.parametrize("name", SITE_MODEL_CASES)
def test_fixtures_with_the_device_eigensolver(gpu_engine_cls, oracle, name):
    c = load_case(name)
    ts, leaf = tree_structure(c["parent"]), leaf_ids(c)
    Q, props, pi, bl, seq = c["Q"], c["proportions"], c["pi"], c["branch_lengths"], c["sequences"]
    with _engine(gpu_engine_cls, c["parent"], seq, "device") as eng:
        lnL, site = eng.site_classes(Q, props, pi, bl, return_sites=True)
        P_dev = np.stack([eng.export_transition_matrices(k) for k in range(len(Q))])
        again = eng.site_classes(Q, props, pi, bl)
        from_params = eng.codon_site_classes(float(c["kappa"]), c["omegas"], props, pi, bl)
    with _engine(gpu_engine_cls, c["parent"], seq, "host") as eng:
        lnL_h, site_h = eng.site_classes(Q, props, pi, bl, return_sites=True)
    ref, ref_site = oracle.compute_site_class_log_likelihood(list(Q), props, pi, ts, bl, None, seq, leaf, return_sites=True)
    assert again == lnL
    assert abs(lnL - ref) < TOTAL_ATOL and abs(lnL - lnL_h) < TOTAL_ATOL and abs(lnL - float(c["lnL_pyref"])) < TOTAL_ATOL
    assert abs(from_params - lnL) < TOTAL_ATOL
    for k in range(len(Q)):
        for v in range(len(bl)):
            if c["parent"][v] >= 0:
                np.testing.assert_allclose(P_dev[k, v], oracle.matrix_exponential(Q[k], pi, bl[v]), rtol=0, atol=2e-14)
    bound = eigen_route_bound(oracle, P_dev, pi, ts, seq, leaf, lambda x: _mixture(x, props))
    for other in (ref_site, site_h):
        assert np.all(np.abs(site - other) <= SITE_RTOL * np.maximum(np.abs(other), 1e-3) + bound)
    lnf = oracle.site_lnf_from_transition_matrices(P_dev, pi, ts, seq, leaf)       # staged: strict on every site
    assert np.max(np.abs(site - _mixture(lnf, props)) / np.maximum(np.abs(site), 1e-3)) <= SITE_RTOL


@pytest.mark.parametrize("name", ["c2", "lysin", "synthetic_edge", "free_ratio_60"])
def test_device_eigensolver_against_60_digit_truth(gpu_engine_cls, name):
    from test_gpu_parity import _parity_tail_case
    z = np.load(os.path.join(GOLDEN, "parity_tail.npz"), allow_pickle=False)
    c = _parity_tail_case(name)
    sites, truth, lapack = z[f"{name}__sites"], z[f"{name}__truth"], z[f"{name}__lnf_lapack"]
    with _engine(gpu_engine_cls, c["parent"], c["codes"], "device") as eng:
        if c["kind"] == "classes":
            _, site = eng.site_classes(c["Q"], c["props"], c["pi"], c["bl"], return_sites=True)
        else:
            _, site = eng.branch(c["Q"], c["pi"], c["bl"], return_sites=True)
    den = np.maximum(np.abs(truth), 1e-3)
    err_gpu, err_ref = np.abs(site[sites] - truth) / den, np.abs(lapack - truth) / den
    print(f"{name}: device solver max {err_gpu.max():.2e}, LAPACK route max {err_ref.max():.2e}")
    assert err_gpu.max() <= max(10.0 * err_ref.max(), SITE_RTOL)
    assert np.all(err_gpu[err_ref <= 0.1 * SITE_RTOL] <= SITE_RTOL)


def test_other_state_spaces_batches_and_partial_updates(gpu_engine_cls, oracle):
    rng = np.random.default_rng(5)
    for n in (4, 20, 63):
        pi = rng.dirichlet(3.0 * np.ones(n))
        Sx = rng.gamma(1.0, 1.0, size=(n, n))
        Sx = np.triu(Sx, 1) + np.triu(Sx, 1).T
        Q = Sx * pi[None, :]
        np.fill_diagonal(Q, -Q.sum(axis=1))
        Q /= -(pi * np.diag(Q)).sum()
        parent = np.array([4, 4, 5, 5, 6, 6, -1], dtype=np.int32)
        bl = np.array([0.1, 0.2, 0.05, 0.3, 0.15, 0.25, 0.0])
        codes = rng.integers(0, n, size=(4, 70)).astype(np.int32)
        with _engine(gpu_engine_cls, parent, codes, "device", n_states=n) as eng:
            lnL, site = eng.site_classes(Q[None], [1.0], pi, bl, return_sites=True)
        ref, ref_site = oracle.compute_site_class_log_likelihood([Q], [1.0], pi, tree_structure(parent), bl, None, codes, list(range(4)),
                                                                 return_sites=True)
        assert abs(lnL - ref) < TOTAL_ATOL and np.max(np.abs(site - ref_site) / np.abs(ref_site)) < 1e-9, n
    # a gradient stencil in one call equals the single evaluations bit for bit, path programs included
    from crabml_amd import synth
    case = synth.make_case("M8", 20, 300, seed=9)
    nn = case.n_nodes
    B = nn + 2
    bl = np.repeat(case.branch_lengths[None], B, 0)
    bl[np.arange(1, nn), np.arange(nn - 1)] *= 1.0 + 1e-7
    ka = np.full(B, case.kappa)
    ka[nn] *= 1.0 + 1e-7
    om = np.repeat(case.class_omegas[None], B, 0)
    om[nn + 1, 2] *= 1.0 + 1e-7
    pr = np.repeat(np.asarray(case.proportions)[None], B, 0)
    with _engine(gpu_engine_cls, case.parent, case.codes, "device") as eng:
        got = eng.codon_site_classes(ka, om, pr, case.pi, bl)
        assert eng.last_partial_updates() == nn - 1
        want = np.array([eng.codon_site_classes(ka[b], om[b], pr[b], case.pi, bl[b]) for b in range(B)])
    with _engine(gpu_engine_cls, case.parent, case.codes, "device") as fresh:       # no dependence on what the pool has seen
        assert fresh.codon_site_classes(ka[nn], om[nn], pr[nn], case.pi, bl[nn]) == got[nn]
    assert np.array_equal(got, want)
    with _engine(gpu_engine_cls, case.parent, case.codes, "host") as eng:
        host = eng.codon_site_classes(ka, om, pr, case.pi, bl)
    assert np.abs(got - host).max() < TOTAL_ATOL


def test_multi_device_handle_with_the_device_eigensolver(gpu_engine_cls):
    """Every shard decomposes on its own device: same bits from each (one kernel, one input), so the shard-ordered sum
    equals the single engine's to the slice tolerance and repeats bit for bit."""
    from crabml_amd import synth
    case = synth.make_case("M8", 24, 900, seed=12)
    with _engine(gpu_engine_cls, case.parent, case.codes, "device") as one:
        want = one.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
        _, want_sites = one.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
    eng = gpu_engine_cls([0, 0, 0])
    try:
        eng.set_eigensolver("device")
        eng.set_tree(case.parent, np.arange(case.codes.shape[0]))
        eng.set_alignment(case.codes)
        got = eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
        _, sites = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
        again = eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
    finally:
        eng.close()
    assert again == got and abs(got - want) < 1e-7 and np.array_equal(sites, want_sites)


def test_many_matrices_per_call(gpu_engine_cls):
    """Free-ratio at 1000 taxa: a kappa change decomposes 1 998 matrices.  Same likelihood from both solvers; timing printed."""
    from crabml_amd import synth
    case = synth.make_case("free-ratio", 1000, 2000, seed=4)
    out = {}
    for where in ("host", "device"):
        with _engine(gpu_engine_cls, case.parent, case.codes, where) as eng:
            eng.codon_branch(case.kappa, case.omega_per_node, case.pi, case.branch_lengths, no_clamp=True)
            t0 = time.perf_counter()
            vals = [eng.codon_branch(case.kappa * (1 + 1e-6 * (i + 1)), case.omega_per_node, case.pi, case.branch_lengths, no_clamp=True)
                    for i in range(5)]
            out[where] = (vals, (time.perf_counter() - t0) / 5, eng.eig_pool_stats()[1])
    print(f"1 998 new rate matrices per call: host solver {out['host'][1] * 1e3:.1f} ms per evaluation, device solver {out['device'][1] * 1e3:.1f} ms")
    assert out["host"][2] == out["device"][2] >= 5 * 1998
    assert np.abs(np.array(out["host"][0]) - np.array(out["device"][0])).max() < 1e-6 * 10


def test_a_matrix_that_does_not_converge_fails_the_call():
    """Kernel 0 reports a matrix still rotating after its last sweep; the engine fails the call with the eigen status and
    forgets the slots (checked in a child process with the sweep limit lowered to 2: CRAB_EIGEN_MAX_SWEEPS)."""
    import subprocess
    import sys
    code = """
import numpy as np
from crabml_amd import synth
from crabml_amd._lib import CrabError, CRAB_ERR_EIGEN
from crabml_amd.engine import Engine
case = synth.make_case("M8", 12, 200, seed=3)
with Engine(0) as eng:
    eng.set_eigensolver("device")
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes)
    for attempt in range(2):
        try:
            eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths)
        except CrabError as e:
            assert e.status == CRAB_ERR_EIGEN and "did not converge" in str(e), (e.status, str(e))
        else:
            raise SystemExit("no error")
    assert eng.eig_pool_stats()[2] == 0, eng.eig_pool_stats()      # nothing was kept
    eng.set_eigensolver("host")
    print("host", eng.codon_site_classes(case.kappa, case.class_omegas, case.proportions, case.pi, case.branch_lengths))
"""
    env = dict(os.environ, CRAB_EIGEN_MAX_SWEEPS="2", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("host -"), out.stdout + out.stderr
