"""lnL and per-pattern log-likelihoods as bit patterns for a few cases: run with two builds of the library (CRAB_LIB_ROOT) and diff."""
import sys, os, hashlib
root = os.environ.get("CRAB_LIB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
for model, taxa, codons in (("M8", 100, 10000), ("M8", 30, 400), ("M7", 14, 600), ("ModelA", 60, 1000), ("M0", 9, 300)):
    case = synth.make_case(model, taxa, codons, seed=5)
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        if model == "ModelA":
            v, site = eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi, case.branch_lengths, case.branch_labels, return_sites=True)
        else:
            v, site = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, return_sites=True)
            v2 = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths, no_clamp=True)
            print("   no_clamp", float(v2).hex())
        print(model, taxa, codons, float(v).hex(), hashlib.sha1(site.tobytes()).hexdigest()[:16])
