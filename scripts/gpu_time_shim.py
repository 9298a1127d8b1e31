"""The stateless seven-function shim (crabml_rust drop-in) against the handle API at C2: what the per-call alignment
hash and the list -> array conversions of the reference-shaped boundary cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth, crabml_rust as shim
from crabml_amd.engine import Engine
case = synth.make_case("M8", 100, 10000, seed=2)
Ql = [np.ascontiguousarray(q) for q in case.Q]
pr = list(case.proportions)
bl = list(case.branch_lengths)
names = [f"t{i}" for i in range(case.n_taxa)]
leaf = list(range(case.n_taxa))
seq = np.ascontiguousarray(case.codes, dtype=np.int32)
def call(i):
    Ql[0] = Ql[0] * (1 + 1e-9)        # a new rate matrix every call, like an optimiser
    return shim.compute_site_class_log_likelihood(Ql, pr, case.pi, case.tree_structure, bl, names, seq, leaf)
for i in range(3): a = call(i)
t = time.perf_counter(); n = 50
for i in range(n): a = call(i)
dt = (time.perf_counter() - t) / n
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids); eng.set_alignment(case.codes)
    Qs = np.stack(Ql)
    for i in range(3): b = eng.site_classes(Qs, case.proportions, case.pi, case.branch_lengths)
    t = time.perf_counter()
    for i in range(n):
        Qs[0] *= 1 + 1e-9
        b = eng.site_classes(Qs, case.proportions, case.pi, case.branch_lengths)
    dh = (time.perf_counter() - t) / n
print(f"C2 M8: shim compute_site_class_log_likelihood {dt*1e3:.3f} ms per call ({1/dt:.0f}/s), handle API {dh*1e3:.3f} ms ({1/dh:.0f}/s); lnL {a:.6f} / {b:.6f}")
