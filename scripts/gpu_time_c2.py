"""Time the C2 workload's kernels (env CRAB_NT / CRAB_BLOCKS_PER_CU select kernel variants)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
model = sys.argv[1] if len(sys.argv) > 1 else "M8"
n_taxa = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_sites = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
case = synth.make_case(model, n_taxa, n_sites, seed=2)
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    def ev():
        if model == "ModelA":
            return eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi, case.branch_lengths, case.branch_labels)
        if model == "free-ratio":
            return eng.branch(case.Q_per_node, case.pi, case.branch_lengths)
        return eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
    for i in range(3): lnL = ev()
    dev = []; pr = []; eg = []
    t = time.time(); n = 10
    for i in range(n):
        lnL = ev(); a, b, c = eng.last_timing(); dev.append(a); pr.append(b); eg.append(c)
    dt = (time.time() - t) / n
    dense, gather = eng.last_work()
    print(f"{model} {n_taxa}x{n_sites} NT={os.environ.get('CRAB_NT','-')} BPC={os.environ.get('CRAB_BLOCKS_PER_CU','-')}: lnL={lnL!r} wall={dt*1e3:.3f} ms device={np.median(dev):.3f} prune={np.median(pr):.3f} eigen={np.median(eg):.3f} ms; {dense*7442/np.median(pr)/1e9:.2f} TFLOP/s", flush=True)
