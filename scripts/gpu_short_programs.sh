#!/bin/bash
# Short programs (few leaves, one to four products per item) under the 64-pattern and the 128-pattern kernel:
# what a clade launch's items look like, timed as whole walks of tiny trees.
cd "$(dirname "$0")/.."
for shape in "M8 4 300000" "M8 5 300000" "M8 6 300000" "M8 8 200000" "M8 100 10000"; do
    for nt in 1 2; do
        CRAB_NT=$nt CRAB_NO_CLADE_TABLES=1 CRAB_NO_TREE_PARTS=1 timeout -k 10 300 python3 scripts/gpu_time_c2.py $shape || exit 1
    done
done
