#!/usr/bin/env python3
"""Whole fits through the optimiser mirror (crabml_amd/optimize.py) on lysozyme: the batched-gradient form against the
reference's form (scipy differentiating through n + 1 single calls), same iterates either way.
    python scripts/gpu_time_fit.py > gpurun_out/fit_timing.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_optimize import _lysozyme   # noqa: E402

from crabml_amd.optimize import M0Optimizer, M1aOptimizer, M2aOptimizer, M7Optimizer, M8aOptimizer, M8Optimizer   # noqa: E402

for cls in (M0Optimizer, M1aOptimizer, M2aOptimizer, M7Optimizer, M8Optimizer, M8aOptimizer):
    for batched in (True, False, True, False):
        aln, tree = _lysozyme()
        opt = cls(aln, tree, batched_gradient=batched)
        t0 = time.perf_counter()
        out = opt.optimize()
        dt = time.perf_counter() - t0
        print(f"{cls.__name__:13s} {'batched   ' if batched else 'sequential'}  lnL {out[-1]:.6f}  {opt.result.nit:3d} iterations  "
              f"{opt.n_calls:5d} calls  {opt.n_vectors:5d} vectors  {dt * 1e3:8.1f} ms", flush=True)
