#!/usr/bin/env python3
"""A fit at BASELINE configs[1] size (M8, 100 taxa x 10 000 codons) through the optimiser mirror: a few L-BFGS-B
iterations, each gradient (1 + 5 + 197 vectors) one batched call.
    python scripts/gpu_time_fit_c2.py [maxiter]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crabml_amd import synth   # noqa: E402
from crabml_amd.optimize import M8Optimizer   # noqa: E402

maxiter = int(sys.argv[1]) if len(sys.argv) > 1 else 3
aln, tree = synth.as_alignment_and_tree(synth.make_case("M8", 100, 10000, seed=1))
opt = M8Optimizer(aln, tree, init_with_m0=False)
opt.compute_log_likelihood(opt.result.x if hasattr(opt, "result") else
                           [0.7, 2.2, -0.36, 0.4, 0.9] + [-2.3] * opt.n_branches)   # first use of the device
opt.n_calls = opt.n_vectors = 0
t0 = time.perf_counter()
out = opt.optimize(maxiter=maxiter)
dt = time.perf_counter() - t0
print(f"M8 fit, 100 taxa x 10000 codons, {opt.n_branches} branches: {opt.result.nit} iterations, {opt.n_calls} gradient calls, "
      f"{opt.n_vectors} parameter vectors in {dt:.2f} s: {dt / opt.n_calls * 1e3:.0f} ms per gradient, {opt.n_vectors / dt:.0f} lnL evaluations/s")
print("lnL", [round(h["log_likelihood"], 3) for h in opt.history])
