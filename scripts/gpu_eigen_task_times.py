"""Start and end of every host decomposition of a configs[1] evaluation inside the worker pool (CRAB_DEBUG_EIGEN_TIMES=1)."""
import sys, os
os.environ["CRAB_DEBUG_EIGEN_TIMES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
case = synth.make_case("M8", 100, 10000, seed=2)
Q = np.asarray(case.Q)
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    for i in range(12):
        eng.site_classes(Q * (1.0 + 1e-6 * (i + 1)), case.proportions, case.pi, case.branch_lengths)
