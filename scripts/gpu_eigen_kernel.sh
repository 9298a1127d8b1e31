#!/bin/bash
# Device eigensolver: its tests, then kernel durations over one gene's M7 + M8 fits (rocprofv3 kernel stats).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-eig}
mkdir -p $OUT
timeout -k 10 600 python3 -u -m pytest tests/test_device_eigen.py -q --timeout 300 --timeout-method thread -s > $OUT/tests.log 2>&1 || { tail -n 30 $OUT/tests.log; exit 1; }
grep -E "solver|passed|failed" $OUT/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_scan -o scan --output-format csv -- python3 scripts/gpu_prof_scan.py device > $OUT/prof_scan.log 2>&1
find $OUT/prof_scan -name "*kernel_trace.csv" -delete
find $OUT/prof_scan -name "*kernel_stats*" -exec cat {} +
grep unprofiled $OUT/prof_scan.log
