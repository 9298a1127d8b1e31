#!/usr/bin/env python3
"""Where one gene's M7 + M8 fits spend their time (config 5, one engine): cProfile of scan.lrt_gene, top entries by own time.
    python scripts/gpu_prof_scan.py [host|device]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crabml_amd import scan, synth   # noqa: E402
from crabml_amd.engine import Engine   # noqa: E402

where = sys.argv[1] if len(sys.argv) > 1 else "host"
genes = [synth.as_alignment_and_tree(synth.make_case("M8", 30, 400, seed=2000 + g)) for g in range(3)]
with Engine(0) as eng:
    eng.set_eigensolver(where)
    scan.lrt_gene(eng, *genes[0])
    t0 = time.perf_counter()
    scan.lrt_gene(eng, *genes[1])
    print(f"{where}: one gene, unprofiled: {time.perf_counter() - t0:.3f} s")
    pr = cProfile.Profile()
    pr.enable()
    out = scan.lrt_gene(eng, *genes[2])
    pr.disable()
print({k: (v["vectors"], v["calls"]) for k, v in out["fits"].items()})
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
