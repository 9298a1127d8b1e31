"""BASELINE config 5 shape: many small genes (30 taxa x 400 codons, M8) on one GPU, `workers` engines at once.
    python scripts/gpu_time_genes.py [n_genes] [evals_per_gene]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.dist import map_genes
from crabml_amd.engine import discretise_omega

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_eval = int(sys.argv[2]) if len(sys.argv) > 2 else 40
genes = [synth.make_case("M8", 30, 400, seed=1000 + g) for g in range(n_genes)]
om = np.append(discretise_omega("beta", 0.7, 1.5, 10), 2.5)
pr = np.append(np.full(10, 0.09), 0.1)


def optimise_like(batch):
    def fn(eng, case):   # what one gene's optimisation does to the engine: re-target, then a run of objective calls
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61)
        out = 0.0
        for i in range(n_eval):
            if batch == 1:
                out = eng.codon_site_classes(case.kappa * (1 + 1e-3 * i), om, pr, case.pi, case.branch_lengths)
            else:
                # objective + forward-difference gradient of M8 in one call, as L-BFGS-B asks for it (optimizer.py:1197-1203):
                # the point itself, one vector per branch length, and kappa, p0, p, q, omega_s perturbed in turn
                nb = len(case.branch_lengths) - 1
                B = 1 + nb + 5
                kappa = case.kappa * (1 + 1e-3 * i)
                ka = np.full(B, kappa)
                bl = np.repeat(case.branch_lengths[None], B, 0)
                bl[np.arange(1, 1 + nb), np.arange(nb)] *= 1 + 1e-6
                oms, prs = np.repeat(om[None], B, 0), np.repeat(pr[None], B, 0)
                ka[1 + nb] *= 1 + 1e-6
                prs[2 + nb] = np.append(np.full(10, 0.0900001), 1 - 0.900001)
                oms[3 + nb] = np.append(discretise_omega("beta", 0.7 * (1 + 1e-6), 1.5, 10), 2.5)
                oms[4 + nb] = np.append(discretise_omega("beta", 0.7, 1.5 * (1 + 1e-6), 10), 2.5)
                oms[5 + nb, 10] *= 1 + 1e-6
                out = eng.codon_site_classes(ka, oms, prs, case.pi, bl)[0]
        return out
    return fn


ref = None
for batch in (1, 63):
    for workers in (1, 2, 4, 8, 16):
        map_genes(optimise_like(batch), genes[:workers], workers=workers)   # warm-up: library, allocations
        t = time.perf_counter()
        out = map_genes(optimise_like(batch), genes, workers=workers)
        dt = time.perf_counter() - t
        if ref is None:
            ref = out
        same = all(a == b for a, b in zip(out, ref)) 
        print(f"{n_genes} genes x {n_eval} objective calls, {batch:2d} vector(s) per call, {workers:2d} engines: "
              f"{n_genes * n_eval * batch / dt:9.0f} gene-evaluations/s   (results {'identical' if same else 'DIFFER'})", flush=True)
