"""Time the BEB grid integration on the device (crab_eval_beb_grid) against the CPU restatement of the reference's
loops (vectorised numpy; the reference itself runs them in pure Python: ~22 s for 130 sites in the build container)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from crabml_amd import beb, models, synth
from crabml_amd.engine import Engine
from oracle import beb as obeb

cols, props = beb.paml_grid_m2a()
for n_taxa, n_sites in ((7, 130), (100, 10000)):
    case = synth.make_case("M8", n_taxa, n_sites, seed=2)
    m = models.M2aCodonModel(kappa=2.5, omega0=0.2, omega2=2.5, p0=0.7, p1=0.2, pi=case.pi)
    pr, om = m.get_site_classes()
    Q = beb.beb_rate_matrices(2.5, case.pi, pr, om)
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        eng.beb_grid(Q, case.pi, case.branch_lengths, cols, props)
        t = time.perf_counter(); n = 5
        for _ in range(n):
            post, lnfx, ln_fx = eng.beb_grid(Q, case.pi, case.branch_lengths, cols, props)
        dev = (time.perf_counter() - t) / n
        t = time.perf_counter()
        table = eng.site_lnl_by_class(Q, case.pi, case.branch_lengths)
        tbl = time.perf_counter() - t
        n_pat = eng.n_patterns
    t = time.perf_counter()
    o_post, _, o_ln_fx = obeb.site_posteriors(table, cols, props)
    cpu = time.perf_counter() - t
    print(f"{n_taxa} taxa x {n_sites} codons ({n_pat} patterns), 21 classes, 10 000 grid points: device {dev*1e3:.2f} ms per BEB "
          f"(21-class table alone {tbl*1e3:.2f} ms); numpy restatement of the integration {cpu:.2f} s; "
          f"max |posterior difference| {np.abs(post - o_post).max():.1e}, ln f(X) {ln_fx:.6f} vs {o_ln_fx:.6f}", flush=True)

# ---- the adaptive-grid path (compute_site_posteriors, beb.py:663-731): 5^4 = 625 weighted M2a parameter vectors ----
from crabml_amd.beb import site_posteriors_over_grid
from crabml_amd.engine import build_site_class_q_matrices
case = synth.make_case("M8", 100, 10000, seed=2)
rng = np.random.default_rng(0)
grid = []
for _ in range(625):
    p0, p1 = sorted(rng.uniform(0.05, 0.45, size=2))
    w0, w2, ka = rng.uniform(0.05, 0.9), rng.uniform(1.2, 6.0), 2.5 * np.exp(rng.normal(0, 0.1))
    grid.append((ka, [w0, 1.0, w2], [p0, p1, 1.0 - p0 - p1]))
wts = rng.dirichlet(np.ones(len(grid)))
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    site_posteriors_over_grid(eng, grid[:64], wts[:64], case.pi, case.branch_lengths)
    t = time.perf_counter()
    post = site_posteriors_over_grid(eng, grid, wts, case.pi, case.branch_lengths, batch=64)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    for ka, om, pr in grid[:20]:
        eng.site_lnl_by_class(build_site_class_q_matrices(ka, om, pr, case.pi), case.pi, case.branch_lengths)
    one = (time.perf_counter() - t) / 20
print(f"adaptive-grid BEB, 100 taxa x 10000 codons, 625 grid points x 3 classes: {dt:.3f} s on the device in batches of 64 "
      f"({dt / 625 * 1e3:.2f} ms per grid point); one [S, 3] table per grid point through crab_eval_site_lnl_by_class: {one * 1e3:.2f} ms each; "
      f"row sums {post.sum(axis=1).min():.12f}..{post.sum(axis=1).max():.12f}", flush=True)
