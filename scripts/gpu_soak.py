"""Soak: engine life cycles and long runs must not leak device memory or drift.
    python scripts/gpu_soak.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crabml_amd import synth, crabml_rust as shim
from crabml_amd.engine import Engine

case = synth.make_case("M8", 30, 400, seed=1)
with Engine(0) as eng:                     # first use: the HIP runtime loads code objects and sets up its own pools
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes)
    eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
free0 = torch.cuda.mem_get_info()[0]
ref = None
for i in range(200):                       # 200 engine life cycles
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes)
        v = eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
        ref = v if ref is None else ref
        assert v == ref
free1 = torch.cuda.mem_get_info()[0]
big = synth.make_case("M8", 100, 10000, seed=2)
with Engine(0) as eng:
    eng.set_tree(big.parent, big.leaf_node_ids)
    eng.set_alignment(big.codes)
    first = eng.site_classes(big.Q, big.proportions, big.pi, big.branch_lengths)
    t = time.perf_counter()
    for i in range(3000):                  # a long optimisation's worth of evaluations, new matrices every call
        q = big.Q * (1.0 + 1e-9 * (i % 5000))
        v = eng.site_classes(q, big.proportions, big.pi, big.branch_lengths)
    dt = time.perf_counter() - t
    again = eng.site_classes(big.Q, big.proportions, big.pi, big.branch_lengths)
    free_mid = torch.cuda.mem_get_info()[0]
free2 = torch.cuda.mem_get_info()[0]
# the stateless shim over its engine cache, alternating two alignments
other = synth.make_case("M8", 30, 400, seed=2)
for i in range(100):
    for c in (case, other):
        shim.compute_site_class_log_likelihood(list(c.Q), list(c.proportions), c.pi, c.tree_structure, list(c.branch_lengths),
                                               None, np.ascontiguousarray(c.codes, dtype=np.int32), list(range(30)))
shim_free = torch.cuda.mem_get_info()[0]
print(f"free HBM: after first use {free0>>20} MiB, after 200 engine life cycles {free1>>20} MiB, during the long run {free_mid>>20} MiB, "
      f"after it {free2>>20} MiB, with the shim's two cached engines {shim_free>>20} MiB")
print(f"3000 evaluations with fresh matrices: {3000/dt:.0f}/s; first == last on the same inputs: {first == again}")
assert first == again and abs(free1 - free0) < (64 << 20) and abs(free2 - free0) < (64 << 20)
