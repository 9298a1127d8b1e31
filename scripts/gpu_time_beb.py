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
