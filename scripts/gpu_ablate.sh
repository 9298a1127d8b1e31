#!/bin/bash
# Per-launch pruning times of every library variant under build_variants/ (rocprofv3 kernel trace of gpu_time_c2.py).
#   scripts/gpu_ablate.sh [launches per evaluation = 4] [gpu_time_c2.py args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PER=${1:-4}; shift
ROOT=$PWD
for d in build_variants/*/; do
    name=$(basename $d)
    sed "s#sys.path.insert(0, .*#sys.path.insert(0, \"$ROOT/$d\")#" scripts/gpu_time_c2.py > /tmp/t_$name.py
    out=gpurun_out/ablate/$name
    rm -rf $out; mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 /tmp/t_$name.py "$@" > $out/log.txt 2>&1
    rc=$?
    if [ $rc -ge 124 ]; then echo "$name: rc $rc, stopping"; tail -3 $out/log.txt; exit 1; fi   # time limit or a crash: nothing more on the GPU
    if [ $rc -ne 0 ]; then echo "== $name: rc $rc: $(tail -1 $out/log.txt)"; fi
    echo "== $name: $(grep -v rocprof $out/log.txt | tail -1)"
    python3 scripts/prune_launch_times.py $(find $out -name '*kernel_trace.csv' | head -1) $PER 4 | grep -v gap
done
