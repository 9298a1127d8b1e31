#!/usr/bin/env python3
"""BASELINE config 5 end to end on one GPU: M7 and M8 fitted to synthetic genes (30 taxa x 400 codons, simulated under
M8) through the optimiser mirror, `workers` engines sharing the card; likelihood-ratio test per gene.
    python scripts/gpu_time_scan.py [n_genes] [workers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402

from crabml_amd import scan, synth   # noqa: E402

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8


genes = [synth.as_alignment_and_tree(synth.make_case("M8", 30, 400, seed=2000 + g)) for g in range(n_genes)]
scan.scan(genes[:2], workers=2)   # first use of the device and of scipy's code paths
for w in (1, workers):
    t0 = time.perf_counter()
    rows = scan.scan(genes, null="M7", workers=w)
    dt = time.perf_counter() - t0
    vec = sum(f["vectors"] for r in rows for f in r["fits"].values())
    print(f"{n_genes} genes, {w} worker(s): {dt:.2f} s  ({dt / n_genes * 1e3:.0f} ms per gene, M7 + M8 fits; {vec} parameter vectors, "
          f"{vec / dt:.0f} vectors/s)", flush=True)
print("LRT statistics:", np.round([r["LRT"] for r in rows], 2))
print("p < 0.05 in", sum(r["pvalue"] < 0.05 for r in rows), "of", n_genes, "genes (simulated with 10% of sites at omega = 2.5)")
