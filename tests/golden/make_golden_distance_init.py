#!/usr/bin/env python3
"""Golden vector for the least-squares starting branch lengths, from the reference's own code.

Runs ONLY in the build container (reference checkout at /root/reference): imports the reference's
``crabml.optimize.distance_init`` with its own Alignment and Tree classes and calls ``compute_pairwise_distances`` and
``initialize_branch_lengths_ls`` on the lysin example exactly as tests/test_paml_examples/test_lysin.py prepares it.
Output: tests/golden/lysin_distance_init.npz (arrays only).

    python tests/golden/make_golden_distance_init.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    pkg = types.ModuleType("crabml")
    pkg.__path__ = [f"{REF}/src/crabml"]
    sys.modules["crabml"] = pkg
    opt = types.ModuleType("crabml.optimize")           # skip optimize/__init__.py (it imports the Rust extension)
    opt.__path__ = [f"{REF}/src/crabml/optimize"]
    sys.modules["crabml.optimize"] = opt
    from crabml.io.sequences import Alignment
    from crabml.io.trees import Tree
    from crabml.optimize.distance_init import compute_pairwise_distances, initialize_branch_lengths_ls

    aln = Alignment.from_fasta(os.path.join(HERE, "data", "lysin.fasta"), seqtype="codon")
    lines = [ln.strip() for ln in open(os.path.join(HERE, "data", "lysin.trees")) if ln.strip()]
    text = "".join(ln for ln in lines[1:] if not ln.startswith("#")).replace(": ", ":").replace(" #1", "")
    tree = Tree.from_newick(text)
    branch_nodes = [n for n in tree.postorder() if n.parent is not None]
    d = compute_pairwise_distances(aln)
    b = initialize_branch_lengths_ls(aln, tree, branch_nodes)
    np.savez_compressed(os.path.join(HERE, "lysin_distance_init.npz"), pairwise=d, branch_lengths=b,
                        branch_node_names=np.array([n.name or "" for n in branch_nodes]), newick=np.array(text))
    print("pairwise", d.shape, "branches", b.shape, "mean", b.mean())


if __name__ == "__main__":
    main()
