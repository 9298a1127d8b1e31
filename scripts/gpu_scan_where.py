#!/usr/bin/env python3
"""Where a scan's wall time goes per engine call, against the number of engines sharing the GPU: the device section (HIP
events on the engine's stream), the rest of the C call (host work, waiting for the stream), and everything outside the
C call (Python, waiting for the interpreter lock).   python scripts/gpu_scan_where.py [n_genes] [w1,w2,...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crabml_amd import scan, synth   # noqa: E402
from crabml_amd.engine import Engine   # noqa: E402

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ws = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
genes = [synth.as_alignment_and_tree(synth.make_case("M8", 30, 400, seed=2000 + g)) for g in range(n_genes)]
acc = {"calls": 0, "c_wall": 0.0, "device": 0.0}
lock = threading.Lock()
orig = Engine.codon_site_classes


def timed(self, *a, **k):
    t0 = time.perf_counter()
    out = orig(self, *a, **k)
    dt = time.perf_counter() - t0
    dev = self.last_timing()[0]
    with lock:
        acc["calls"] += 1
        acc["c_wall"] += dt
        acc["device"] += dev * 1e-3
    return out


Engine.codon_site_classes = timed
scan.scan(genes[:2], workers=2)
for w in ws:
    acc.update(calls=0, c_wall=0.0, device=0.0)
    t0 = time.perf_counter()
    scan.scan(genes, null="M7", workers=w)
    dt = time.perf_counter() - t0
    n = acc["calls"]
    per_call = dt * w / n
    print(f"{w:2d} engine(s): {dt / n_genes * 1e3:.0f} ms per gene; per call and engine {per_call * 1e3:.2f} ms = device section {acc['device'] / n * 1e3:.2f}"
          f" + rest of the C call {(acc['c_wall'] - acc['device']) / n * 1e3:.2f} + outside the call {(per_call - acc['c_wall'] / n) * 1e3:.2f}", flush=True)
