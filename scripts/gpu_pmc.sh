#!/bin/bash
# PMC passes over the C2 timing run (one rocprofv3 run per counter group; see MI355X_MICROARCH.md slot limits)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${1:-run}
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 scripts/gpu_time_c2.py M8 100 10000 > $OUT/$name.log 2>&1; }
run l2 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
for n in l2 fetch write sq; do echo "== $n"; python3 scripts/pmc_summary.py $OUT/$n; done
