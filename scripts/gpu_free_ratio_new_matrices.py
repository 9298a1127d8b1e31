"""Free-ratio evaluations in which every one of the 1 998 rate matrices is new (an optimiser step on the omegas): where the
time goes with the host and with the device eigensolver.   python scripts/gpu_free_ratio_new_matrices.py [taxa] [codons]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
taxa = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
codons = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
case = synth.make_case("free-ratio", taxa, codons, seed=2)
Q = np.asarray(case.Q_per_node)
for where in ("host", "device"):
    with Engine(0) as eng:
        eng.set_eigensolver(where)
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        rows = []
        for i in range(8):
            Qi = Q * (1.0 + 1e-6 * (i + 1))
            t = time.perf_counter()
            eng.branch(Qi, case.pi, case.branch_lengths)
            w = (time.perf_counter() - t) * 1e3
            d, p, e = eng.last_timing()
            rows.append((w, d, p, e))
        w, d, p, e = np.median(np.array(rows[2:]), axis=0)
        print(f"{where:6s} eigensolver: wall {w:.2f} ms = host phase before the device {e:.2f} + device {d:.2f} (pruning {p:.2f}) + rest {w - d - e:.2f}", flush=True)
