#!/bin/bash
# Clade nesting depth and planner costs with all levels in one launch (round 3): pruning time per setting.
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/r3m/levels.txt}
: > $OUT
for shape in "M8 100 10000" "M8 500 50000" "ModelA 500 6250"; do
  for lv in 3 4 5 7; do
    for ic in 1.0 0.5 0.25; do
      echo "levels $lv item_cost $ic" >> $OUT
      CRAB_CLADE_MAX_LEVELS=$lv CRAB_CLADE_ITEM_COST=$ic timeout -k 5 200 python scripts/gpu_time_c2.py $shape >> $OUT 2>&1 || exit 1
    done
  done
done
