#!/bin/bash
# Per-launch durations of the pruning kernel inside an evaluation (rocprofv3 kernel trace of gpu_time_c2.py).
#   scripts/gpu_launch_times.sh <launches per evaluation> <model> <taxa> <codons>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PER=$1; shift
out=gpurun_out/launch_times
rm -rf $out; mkdir -p $out
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 scripts/gpu_time_c2.py "$@" > $out/log.txt 2>&1 || { tail -3 $out/log.txt; exit 1; }
grep -v rocprof $out/log.txt | tail -1
python3 scripts/prune_launch_times.py $(find $out -name '*kernel_trace.csv' | head -1) $PER 4 | grep -v gap
