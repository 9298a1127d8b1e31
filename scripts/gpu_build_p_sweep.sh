#!/bin/bash
# build_p_kernel's duration against matrices per workgroup (rocprofv3 kernel stats of the bench workload)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pw in 0 1 2 3 5 9; do
    out=gpurun_out/bp_sweep/$pw; rm -rf $out; mkdir -p $out
    CRAB_BUILD_P_PER_WG=$pw timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-target > $out/bench.json 2>/dev/null || exit 1
    echo "per_wg $pw: $(grep build_p $out/*kernel_stats.csv | cut -d, -f9-13)"
done
