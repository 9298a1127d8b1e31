#!/bin/bash
# The BASELINE configurations' shapes on one GPU, with clade tables (default) and as whole walks / tree parts (CRAB_NO_CLADE_TABLES=1):
# per-evaluation pruning time and executed TFLOP/s.
cd "$(dirname "$0")/.."
for off in 0 1; do
    echo "== CRAB_NO_CLADE_TABLES=$off"
    for shape in "M8 100 10000" "ModelA 500 6250" "free-ratio 1000 12500" "ModelA 500 50000" "free-ratio 1000 100000" "M8 30 400"; do
        CRAB_NO_CLADE_TABLES=$off timeout -k 10 400 python3 scripts/gpu_time_c2.py $shape || exit 1
    done
done
