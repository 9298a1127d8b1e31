"""Whole finite-difference gradients in one call at the config 3 / config 4 shapes (one GPU's share of the sites):
B = 1 + number of parameters, each vector differing from the base point in one parameter."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine

def run(label, eng, call, B):
    out = call()
    t = time.perf_counter()
    out = call()
    dt = time.perf_counter() - t
    dev, pr, eg = eng.last_timing()
    print(f"{label}: {B} vectors in one call: {dt:.3f} s ({B / dt:.0f} evaluations/s; device {dev:.0f} ms, host eigen {eg:.1f} ms); "
          f"distinct lnL values {len(set(np.round(out, 9).tolist()))}", flush=True)

case = synth.make_case("ModelA", 500, 6250, seed=3)
nn = case.n_nodes
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids); eng.set_alignment(case.codes)
    B = 1 + (nn - 1) + 5                      # 997 branch lengths + kappa, omega0, omega2, p0, p1
    P = np.repeat(np.array([case.kappa, case.omega0, case.omega2, case.p0, case.p1])[None], B, 0)
    bl = np.repeat(case.branch_lengths[None], B, 0)
    bl[np.arange(1, nn), np.arange(nn - 1)] *= 1 + 1e-6
    P[np.arange(nn, nn + 5), np.arange(5)] *= 1 + 1e-6
    run("Model A 500 taxa x 6250 codons", eng,
        lambda: eng.codon_branch_site(P[:, 0], P[:, 1], P[:, 2], P[:, 3], P[:, 4], case.pi, bl, case.branch_labels), B)
case = synth.make_case("free-ratio", 1000, 12500, seed=4)
nn = case.n_nodes
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids); eng.set_alignment(case.codes)
    B = 1 + 600                               # a 600-parameter slice of the 3 995-parameter gradient: 300 branch lengths, 300 omegas
    om = np.repeat(case.omega_per_node[None], B, 0)
    bl = np.repeat(case.branch_lengths[None], B, 0)
    bl[np.arange(1, 301), np.arange(300)] *= 1 + 1e-6
    om[np.arange(301, 601), np.arange(300)] *= 1 + 1e-6
    run("free-ratio 1000 taxa x 12500 codons", eng, lambda: eng.codon_branch(np.full(B, case.kappa), om, case.pi, bl), B)
