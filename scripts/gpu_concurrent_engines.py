"""Several engines of one process evaluating at once (one host thread and one stream each, as crabml_amd.dist.map_genes runs them):
do the merged clade launches of concurrent streams still find all their XCDs served?  Prints how many engines fell back to level
launches and checks every value against a single-threaded evaluation.   usage: gpu_concurrent_engines.py [threads] [rounds]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
shapes = [("M8", 40, 1500), ("M8", 100, 4000), ("M8", 24, 600), ("M8", 60, 9000)]
cases = [synth.make_case(m, t, s, seed=5 + i) for i, (m, t, s) in enumerate(shapes)]
ref = []
for case in cases:
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        ref.append(eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths))
        print(f"{case.codes.shape}: {eng.last_clade_tables()[0]} clades, {eng.last_prune_launches()} launches", flush=True)
bad = [0] * n_threads
launches = [0] * n_threads


def worker(w):
    case = cases[w % len(cases)]
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        for _ in range(rounds):
            v = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
            bad[w] += v != ref[w % len(cases)]
        launches[w] = eng.last_prune_launches()


t0 = time.time()
th = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"{n_threads} engines x {rounds} evaluations in {time.time() - t0:.2f} s; wrong values per engine {bad}; launches per evaluation at the end {launches}")
sys.exit(1 if any(bad) else 0)
