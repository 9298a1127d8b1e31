#!/usr/bin/env python3
"""Config 5 scan throughput against the number of engines sharing the GPU.   python scripts/gpu_scan_workers.py [n_genes] [w1,w2,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crabml_amd import scan, synth   # noqa: E402

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ws = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
genes = [synth.as_alignment_and_tree(synth.make_case("M8", 30, 400, seed=2000 + g)) for g in range(n_genes)]
scan.scan(genes[:2], workers=2)
for w in ws:
    t0 = time.perf_counter()
    rows = scan.scan(genes, null="M7", workers=w)
    dt = time.perf_counter() - t0
    calls = sum(f["calls"] for r in rows for f in r["fits"].values())
    print(f"{w:2d} worker(s): {dt / n_genes * 1e3:.0f} ms per gene; {calls / n_genes:.0f} calls per gene, {dt * w / calls * 1e3:.2f} ms per call per worker", flush=True)
