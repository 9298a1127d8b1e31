#!/bin/bash
# Card-busy share of the config 5 scan: kernel trace of scripts/gpu_scan_workers.py, summarised by scripts/card_busy.py.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-busy}; W=${2:-4}
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/trace -o scan --output-format csv -- python3 scripts/gpu_scan_workers.py 32 $W > $OUT/scan.log 2>&1
cat $OUT/scan.log | grep worker
T=$(find $OUT/trace -name "*kernel_trace.csv")
python3 scripts/card_busy.py $T 3.0 | tee $OUT/card_busy.txt
rm -f $T
