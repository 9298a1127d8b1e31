#!/bin/bash
# Clade-table cost model sweep at configs[1] (bench.py, 30 steps): evaluations/s and pruning time per setting.
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
run() {
    local tag="$1"; shift
    env "$@" timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-target > $OUT/b.json 2>/dev/null || { echo "$tag: failed"; return; }
    python3 - "$tag" $OUT/b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:46s} {d['lnl_evals_per_s']:7.1f} evals/s  prune {d['breakdown_ms']['prune_kernel']:.3f} ms  executed products {d['roofline']['dense_updates_per_launch']}")
PY
}
run "defaults" X=1
run "off" CRAB_NO_CLADE_TABLES=1
for pc in 1.5 2.0 3.0; do run "prepass $pc" CRAB_CLADE_PREPASS_COST=$pc; done
for gc in 0.05 0.3 0.6; do run "gather $gc" CRAB_CLADE_GATHER_COST=$gc; done
for ic in 0.5 1.0 2.0; do run "item $ic" CRAB_CLADE_ITEM_COST=$ic; done
for ml in 4 8 12; do run "max leaves $ml" CRAB_CLADE_MAX_LEAVES=$ml; done
run "prepass 2 item 1" CRAB_CLADE_PREPASS_COST=2.0 CRAB_CLADE_ITEM_COST=1.0
run "prepass 2 gather 0.3" CRAB_CLADE_PREPASS_COST=2.0 CRAB_CLADE_GATHER_COST=0.3
