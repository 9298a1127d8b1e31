#!/bin/bash
# Partial updates on/off, same box: a whole finite-difference gradient at configs[1] size (198 vectors) and at gene size,
# the M8 fit at configs[1] size, the per-gene scan (config 5), gradients at the config 3/4 per-GPU shares.
cd "$(dirname "$0")/.."
for off in 0 1; do
  echo "== CRAB_NO_PATH_UPDATES=$off"
  CRAB_NO_PATH_UPDATES=$off timeout -k 10 300 python3 scripts/gpu_time_batch.py M8 100 10000 1 16 198 || exit 1
  CRAB_NO_PATH_UPDATES=$off timeout -k 10 300 python3 scripts/gpu_time_batch.py M8 30 400 1 58 || exit 1
  CRAB_NO_PATH_UPDATES=$off timeout -k 10 300 python3 scripts/gpu_time_fit_c2.py 3 || exit 1
  CRAB_NO_PATH_UPDATES=$off timeout -k 10 400 python3 scripts/gpu_time_scan.py 16 8 || exit 1
  CRAB_NO_PATH_UPDATES=$off timeout -k 10 400 python3 scripts/gpu_time_gradients.py || exit 1
done
