"""Share-versus-whole timing on ONE GPU (VERDICT r2 #3c): the pruning time of every 1/W slice of an alignment's patterns
(the slices an N = W run gives its ranks, in the engine's shard order) against the whole alignment's.
The projected W-way speed-up of the pruning step is whole / slowest slice.   usage: gpu_time_shares.py MODEL TAXA CODONS [W...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crabml_amd import synth
from crabml_amd.dist import pattern_slices
from crabml_amd.engine import Engine, compress_patterns
model, n_taxa, n_sites = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
Ws = [int(x) for x in sys.argv[4:]] or [2, 4, 8]
case = synth.make_case(model, n_taxa, n_sites, seed=6 if model == "M8" else 3)
n_pat = len(compress_patterns(case.codes)[1])


def prune_ms(begin, end):
    with Engine(0) as eng:
        eng.set_tree(case.parent, case.leaf_node_ids)
        eng.set_alignment(case.codes, n_states=61, pattern_begin=begin, pattern_end=end)

        def ev():
            if model == "ModelA":
                return eng.branch_site(case.Q3, *case.qfactors, case.omega0, case.omega2, case.site_class_freqs, case.pi, case.branch_lengths, case.branch_labels)
            if model == "free-ratio":
                return eng.branch(case.Q_per_node, case.pi, case.branch_lengths)
            return eng.site_classes(case.Q, case.proportions, case.pi, case.branch_lengths)
        for _ in range(3):
            ev()
        pr = []
        for _ in range(8):
            ev()
            pr.append(eng.last_timing()[1])
        return float(np.median(pr)), eng.last_work()[0]


whole_ms, whole_dense = prune_ms(0, -1)
print(f"{model} {n_taxa}x{n_sites}: {n_pat} patterns; whole alignment: prune {whole_ms:.3f} ms, {whole_dense} products", flush=True)
for W in Ws:
    res = [prune_ms(b, e) for b, e in pattern_slices(n_pat, W, 128)]
    worst = max(r[0] for r in res)
    print(f"  {W} shares: prune ms {' '.join('%.3f' % r[0] for r in res)}; slowest {worst:.3f} -> projected speed-up {whole_ms / worst:.2f} of {W}; "
          f"products of the largest share {max(r[1] for r in res) / whole_dense:.3f} of the whole", flush=True)
