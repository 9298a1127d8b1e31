"""Clade tables against whole walks on random shapes: trees of 4-300 taxa, 50-6000 codons, divergences from near-identical
sequences (every clade repeats) to saturated ones (nothing repeats), three models; the total and every site must agree bit for bit.
    python scripts/gpu_clade_fuzz.py [cases = 120] [first seed = 1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(first)
used = 0
for i in range(n_cases):
    model = ["M8", "M0", "M7", "ModelA", "free-ratio"][int(rng.integers(0, 5))]
    taxa = int(rng.choice([4, 5, 7, 9, 16, 33, 64, 65, 100, 150, 300]))
    sites = int(rng.choice([50, 64, 65, 300, 1000, 3000, 6000]))
    mb = float(rng.choice([0.0005, 0.005, 0.03, 0.1, 0.5]))
    case = synth.make_case(model, taxa, sites, seed=first * 1000 + i, mean_branch=mb, gap_fraction=float(rng.choice([0.0, 0.01, 0.2])))
    res = []
    for on in (True, False):
        with Engine(0) as eng:
            eng.set_tree(case.parent, case.leaf_node_ids)
            eng.set_alignment(case.codes, n_states=61)
            eng.set_clade_tables(on)
            if model == "ModelA":
                v, s = eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi,
                                       case.branch_lengths, case.branch_labels, return_sites=True)
            elif model == "free-ratio":
                v, s = eng.branch(case.Q_per_node, case.pi, case.branch_lengths, return_sites=True)
            else:
                v, s = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
            res.append((v, s, eng.last_clade_tables()[0] if on else 0, eng.last_work()[0]))
    (v1, s1, ncl, w1), (v0, s0, _, w0) = res
    same = (np.float64(v1).tobytes() == np.float64(v0).tobytes()) and np.array_equal(s1.view(np.uint64), s0.view(np.uint64))
    used += ncl > 0
    print(f"{i:3d} {model:10s} {taxa:3d} x {sites:4d} branch {mb:<6g}: clades {ncl:3d}, products {w1}/{w0}, lnL {v1:.6f} {'same bits' if same else 'DIFFERENT'}", flush=True)
    if not same or not np.isfinite(v1):
        sys.exit(f"case {i}: clade tables and whole walks differ (lnL {v1!r} vs {v0!r})")
print(f"{n_cases} cases, clade tables used in {used}: all bit-identical")
