"""Random trees and alignments: the merged clade launch against whole walks, bit for bit, several evaluations per case
(a stale table line or a missed wait would show as a difference that comes and goes).   usage: gpu_fuzz_clades.py [cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for i in range(n_cases):
    model = ["M8", "ModelA", "free-ratio"][i % 3]
    taxa = int(rng.integers(8, 400))
    sites = int(rng.integers(300, 20000))
    seed = int(rng.integers(1, 1 << 30))
    case = synth.make_case(model, taxa, sites, seed=seed, mean_branch=float(rng.choice([0.02, 0.05, 0.1, 0.3])))
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)

        def ev():
            if model == "ModelA":
                return eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi, case.branch_lengths,
                                       case.branch_labels, return_sites=True)
            if model == "free-ratio":
                return eng.branch(case.Q_per_node, case.pi, case.branch_lengths, return_sites=True)
            return eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)

        runs = [ev() for _ in range(4)]
        n_cl, u = eng.last_clade_tables()
        launches = eng.last_prune_launches()
        eng.set_clade_tables(False)
        whole, site_whole = ev()
    same = all(r[0] == whole and np.array_equal(r[1], site_whole) for r in runs)
    bad += not same
    print(f"{i:3d} {model:10s} {taxa:4d} x {sites:6d} seed {seed:10d}: {n_cl:3d} clades, {launches} launches, lnL {whole!r} {'ok' if same else 'DIFFERENT'}", flush=True)
print(f"{n_cases} cases, {bad} with differences, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
