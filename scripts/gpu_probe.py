"""Quick GPU probe: correctness against the oracle on a small case, then timing of the C2 workload."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
from oracle import oracle as orc

def check(model, n_taxa, n_sites, seed):
    case = synth.make_case(model, n_taxa, n_sites, seed=seed)
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        lnL, site = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
    ref, ref_site = orc.compute_site_class_log_likelihood(list(case.Q), case.proportions, case.pi, case.tree_structure,
                                                          case.branch_lengths, None, case.codes, case.leaf_node_ids, return_sites=True)
    rel = np.max(np.abs(site - ref_site) / np.abs(ref_site))
    print(f"{model} {n_taxa}x{n_sites}: gpu={lnL:.9f} oracle={ref:.9f} |d|={abs(lnL-ref):.3e} max site rel={rel:.3e}", flush=True)

check("M0", 5, 100, 1)
check("M8", 12, 300, 3)
check("M8", 40, 1000, 4)

case = synth.make_case("M8", 100, 10000, seed=2)
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    print("patterns", eng.n_patterns, flush=True)
    for i in range(3):
        lnL = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
    t = time.time(); n = 10
    dev = []; pr = []; eg = []
    for i in range(n):
        lnL = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
        a, b, c = eng.last_timing(); dev.append(a); pr.append(b); eg.append(c)
    dt = (time.time() - t) / n
    dense, gather = eng.last_work()
    print(f"C2 M8 100x10k: lnL={lnL:.6f} wall/eval={dt*1e3:.3f} ms device={np.median(dev):.3f} ms prune={np.median(pr):.3f} ms eigen={np.median(eg):.3f} ms")
    print(f"dense updates={dense} gather={gather} MFMA TFLOP/s (61-based) = {dense*7442/np.median(pr)/1e9:.2f}")
