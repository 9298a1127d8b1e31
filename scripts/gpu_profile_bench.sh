#!/bin/bash
# Round profile of the bench workload: kernel-trace stats, then PMC passes (each in its own rocprofv3 run).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-target > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-target > $OUT/$name.log 2>&1; }
run l2 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
for n in l2 fetch write sq; do echo "== $n"; python3 scripts/pmc_summary.py $OUT/$n; done > $OUT/pmc_summary.txt
cat $OUT/bench.json; find $OUT/trace -name "*kernel_stats*" | head; cat $OUT/pmc_summary.txt
