#!/bin/bash
# Timing ablations of the pruning kernel on the C2 workload (CRAB_DEBUG_SKIP bits: 1 leaf gather, 2 stack, 4 MFMA, 8 LDS-DMA+barriers)
set -e
for dbg in 0 1 2 4 8 3 5 12 15; do
  CRAB_DEBUG_SKIP=$dbg timeout -k 10 120 python scripts/gpu_time_c2.py M8 100 10000 | sed "s/^/dbg=$dbg /"
done
for bpc in 1 2 3; do
  CRAB_BLOCKS_PER_CU=$bpc timeout -k 10 120 python scripts/gpu_time_c2.py M8 100 10000
done
