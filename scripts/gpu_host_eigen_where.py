"""Host eigendecompositions of one configs[1] evaluation: a single decomposition on the calling thread against the eleven of
an evaluation on the worker pool (what bench.py's breakdown_ms.host_eigen holds), and the step around them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine, spectral_decompose

case = synth.make_case("M8", 100, 10000, seed=2)
Q = np.asarray(case.Q)
for k in (0, 5, 10):
    ts = []
    for r in range(200):
        t = time.perf_counter(); spectral_decompose(Q[k], case.pi); ts.append(time.perf_counter() - t)
    print(f"one decomposition (class {k}, ctypes call included): median {np.median(ts)*1e6:.1f} us, min {min(ts)*1e6:.1f}")
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    for gap_ms in (0.0, 5.0):
        wall, dev, eig = [], [], []
        for i in range(60):
            Qi = Q * (1.0 + 1e-6 * (i + 1))     # new matrices every call: eleven cache misses
            if gap_ms: time.sleep(gap_ms * 1e-3)
            t = time.perf_counter()
            eng.site_classes(Qi, case.proportions, case.pi, case.branch_lengths)
            wall.append(time.perf_counter() - t)
            a, b, c = eng.last_timing(); dev.append(a); eig.append(c)
        w, d, e = np.median(wall[5:]) * 1e3, np.median(dev[5:]), np.median(eig[5:])
        print(f"evaluation with eleven new matrices, {gap_ms} ms idle between calls: wall {w:.3f} ms = device {d:.3f} + host eigen {e:.3f} + other {w-d-e:.3f}")
    wall, dev, eig = [], [], []
    for i in range(60):
        t = time.perf_counter()
        eng.site_classes(Q, case.proportions, case.pi, case.branch_lengths)
        wall.append(time.perf_counter() - t)
        a, b, c = eng.last_timing(); dev.append(a); eig.append(c)
    w, d, e = np.median(wall[5:]) * 1e3, np.median(dev[5:]), np.median(eig[5:])
    print(f"evaluation with cached eigensystems: wall {w:.3f} ms = device {d:.3f} + hashing and look-up {e:.3f} + other {w-d-e:.3f}")
