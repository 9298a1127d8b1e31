#!/bin/bash
# Per-GPU shapes of BASELINE configs 3 and 4 (one GPU's share of the 8-GPU pattern sharding) and config 2
set -e
cd "$(dirname "$0")/.."
timeout -k 10 200 python scripts/gpu_time_c2.py M8 100 10000
timeout -k 10 400 python scripts/gpu_time_c2.py ModelA 500 6250
timeout -k 10 600 python scripts/gpu_time_c2.py free-ratio 1000 12500
