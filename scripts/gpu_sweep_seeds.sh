#!/bin/bash
# Runs the randomised GPU sweeps of tests/test_gpu_parity.py and the clade-table comparisons under further seeds
# (CRAB_SWEEP_SEED offsets the generators), one pytest process after the other; stops at the first run that ends in a time
# limit or a crash (nothing more goes to the GPU after that).  usage: gpu_sweep_seeds.sh FIRST LAST
cd "$(dirname "$0")/.."
fails=0
for sd in $(seq ${1:-1} ${2:-20}); do
    CRAB_SWEEP_SEED=$sd timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_clade_tables.py -m gpu -q --timeout 300 --timeout-method thread -k "sweep or random_batches or clade" > /tmp/sweep_$sd.log 2>&1
    rc=$?
    if [ $rc -ge 124 ]; then echo "seed $sd: rc $rc, stopping"; tail -5 /tmp/sweep_$sd.log; exit 1; fi
    out=$(grep -E "^E  .*trial|passed|failed" /tmp/sweep_$sd.log | head -3)
    echo "seed $sd: $(tail -1 /tmp/sweep_$sd.log)"
    case "$out" in *failed*) echo "seed $sd: $out"; fails=$((fails+1));; esac
done
echo "seeds ${1:-1}..${2:-20}: $fails failing"
[ $fails -eq 0 ]
