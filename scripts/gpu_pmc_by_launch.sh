#!/bin/bash
# Counter passes per pruning launch of an evaluation (clade levels, main programs): HBM traffic, L2, issue counters.
#   scripts/gpu_pmc_by_launch.sh <tag> <model> <taxa> <codons>
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmcl_${1:-run}
shift || true
SHAPE="${*:-M8 100 10000}"
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 280 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 scripts/gpu_time_c2.py $SHAPE > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
tail -1 $OUT/sq.log
for n in fetch write l2 sq; do echo "== $n"; python3 scripts/pmc_by_launch.py $OUT/$n; done
