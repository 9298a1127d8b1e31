#!/usr/bin/env python3
"""One evaluation with the device eigensolver on a small case (for instrumented builds: CRAB_AMD_LIB=<variant>)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crabml_amd import synth   # noqa: E402
from crabml_amd.engine import Engine   # noqa: E402

case = synth.make_case("M8", 30, 400, seed=2000)
with Engine(0) as eng:
    eng.set_eigensolver("device")
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes)
    for i in range(2):
        print(eng.codon_site_classes(case.kappa * (1 + 0.01 * i), case.class_omegas, case.proportions, case.pi, case.branch_lengths), flush=True)
