#!/bin/bash
# Same-box A/B: what the presence of the save / load ops costs a plain configs[1] walk (library built with and without
# -DCRAB_NO_PATH_OPS; three alternations).
cd "$(dirname "$0")/.."
export CRAB_NO_PATH_UPDATES=1 CRAB_NO_TREE_PARTS=1
for i in 1 2 3; do
  echo -n "without ops: "; CRAB_AMD_LIB=$PWD/gpurun_ab/libcrabml_nopath.so timeout -k 10 200 python3 scripts/gpu_time_c2.py M8 100 10000 || exit 1
  echo -n "with ops:    "; timeout -k 10 200 python3 scripts/gpu_time_c2.py M8 100 10000 || exit 1
done
