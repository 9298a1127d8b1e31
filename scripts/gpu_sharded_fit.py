#!/usr/bin/env python3
"""Rehearsal of a pattern-sharded fit on ONE card: N ranks (gloo) share the GPU, each holds a slice of the patterns.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 scripts/gpu_sharded_fit.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch.distributed as dist   # noqa: E402

from crabml_amd import synth   # noqa: E402
from crabml_amd.optimize import M8Optimizer   # noqa: E402

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
aln, tree = synth.as_alignment_and_tree(synth.make_case("M8", 20, 2000, seed=5))
opt = M8Optimizer(aln, tree, init_with_m0=False, rank=rank, world_size=world)       # every rank on device 0 here
t0 = time.perf_counter()
out = opt.optimize(maxiter=10)
dt = time.perf_counter() - t0
x = opt.result.x
if world > 1:
    xs = [None] * world
    dist.all_gather_object(xs, x.tolist())
    same = all(v == xs[0] for v in xs)
else:
    same = True
if rank == 0:
    print(f"world {world}: patterns on rank 0 {opt.calc.engine.n_patterns} of {opt.calc.engine.n_patterns_total}, "
          f"{opt.result.nit} iterations, {opt.n_calls} calls, {opt.n_vectors} vectors, {dt:.2f} s, lnL {out[-1]:.9f}, "
          f"identical iterates on all ranks: {same}", flush=True)
    print("x[:5]", np.round(x[:5], 9).tolist(), flush=True)
if world > 1:
    dist.destroy_process_group()
