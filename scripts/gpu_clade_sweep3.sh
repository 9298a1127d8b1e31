#!/bin/bash
# Cost-model knobs below 1: a clade's own walk priced cheaper than the main programs' (whose launch is cut into whole rounds)
cd "$(dirname "$0")/.."
for pp in 1.0 0.85 0.7 0.5; do
  for gc in 0.15 0.05; do
    echo "prepass $pp gather $gc: $(CRAB_CLADE_PREPASS_COST=$pp CRAB_CLADE_GATHER_COST=$gc timeout -k 10 200 python3 scripts/gpu_time_c2.py M8 100 10000 | tail -1 | sed 's/.*wall/wall/') | target $(CRAB_CLADE_PREPASS_COST=$pp CRAB_CLADE_GATHER_COST=$gc timeout -k 10 300 python3 scripts/gpu_time_c2.py M8 500 50000 | tail -1 | sed 's/.*wall/wall/')"
  done
done
