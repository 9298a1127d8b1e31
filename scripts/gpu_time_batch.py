"""Throughput of batched parameter-vector evaluation (crab_eval_site_classes_batch), as a finite-difference gradient
uses it: B = 1 + number of perturbed parameters, branch lengths perturbed one at a time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
model, n_taxa, n_sites = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
batches = [int(x) for x in sys.argv[4:]] or [1, 4, 16]
case = synth.make_case(model, n_taxa, n_sites, seed=2)
with Engine(0) as eng:
    eng.set_tree(case.parent, case.leaf_node_ids)
    eng.set_alignment(case.codes, n_states=61)
    K = len(case.proportions)
    for B in batches:
        Q = np.repeat(case.Q[None], B, axis=0)
        pr = np.repeat(np.asarray(case.proportions)[None], B, axis=0)
        bl = np.repeat(case.branch_lengths[None], B, axis=0).copy()
        for b in range(1, B):
            bl[b, b % (case.n_nodes - 1)] *= 1.0 + 1e-6
        for _ in range(2):
            out = eng.site_classes_batch(Q, pr, case.pi, bl)
        n = max(3, 40 // B)
        t = time.time()
        for _ in range(n):
            out = eng.site_classes_batch(Q, pr, case.pi, bl)
        dt = (time.time() - t) / n
        dev, pr_ms, eg = eng.last_timing()
        dense, _ = eng.last_work()
        print(f"{model} {n_taxa}x{n_sites} B={B}: {B / dt:.1f} evals/s wall ({dt * 1e3:.2f} ms per batch; device {dev:.2f} ms, prune {pr_ms:.2f} ms, "
              f"eigen {eg:.2f} ms); prune {dense * 7442 / pr_ms / 1e9:.2f} TFLOP/s", flush=True)
