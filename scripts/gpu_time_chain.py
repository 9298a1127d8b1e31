"""Pruning kernel on degenerate trees that isolate one phase: a chain of unary internal nodes above a cherry is a pure
stream of products (LDS-DMA + barrier + MFMA, no leaf edges, no stack); a star is a pure stream of leaf edges."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.engine import Engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 97
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
model = sys.argv[3] if len(sys.argv) > 3 else "M8"
case = synth.make_case(model, 100, min(n_sites, 10000), seed=2)
codes = np.tile(case.codes[:2], (1, -(-n_sites // case.codes.shape[1])))[:, :n_sites]
parent = np.full(3 + L, -1, dtype=np.int32)
parent[0] = parent[1] = 2
for v in range(2, 2 + L):
    parent[v] = v + 1
bl = np.full(3 + L, float(os.environ.get('CHAIN_T', '0.02')))
with Engine(0) as eng:
    eng.set_tree(parent, np.arange(2, dtype=np.int32))
    eng.set_alignment(codes, n_states=61, compress=False)
    for _ in range(3):
        eng.site_classes(case.Q, case.proportions, case.pi, bl)
    pr = []
    for _ in range(10):
        eng.site_classes(case.Q, case.proportions, case.pi, bl)
        pr.append(eng.last_timing()[1])
    dense, _ = eng.last_work()
    print(f"chain of {L} products, {n_sites} patterns x {len(case.proportions)} classes: prune {np.median(pr):.3f} ms, {dense * 7442 / np.median(pr) / 1e9:.2f} TFLOP/s "
          f"({dense * 2 * 64 * 64 / np.median(pr) / 1e9:.2f} padded)", flush=True)
