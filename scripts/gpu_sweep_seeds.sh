#!/bin/bash
# Runs the three randomised GPU sweeps of tests/test_gpu_parity.py under further seeds (CRAB_SWEEP_SEED offsets the
# generators).  usage: gpu_sweep_seeds.sh FIRST LAST
cd "$(dirname "$0")/.."
fails=0
for sd in $(seq ${1:-1} ${2:-20}); do
    out=$(CRAB_SWEEP_SEED=$sd timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 --timeout-method thread -k "sweep or random_batches" 2>&1 | grep -E "^E  .*trial|passed|failed" | head -3)
    case "$out" in *failed*) echo "seed $sd: $out"; fails=$((fails+1));; esac
done
echo "seeds ${1:-1}..${2:-20}: $fails failing"
[ $fails -eq 0 ]
