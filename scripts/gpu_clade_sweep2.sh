#!/bin/bash
# Clade tables with nesting: cost-model knobs at configs[1] and the target workload (bench.py, 30 steps).
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-sweep2}; mkdir -p $OUT
run() {
    local tag="$1"; shift
    env "$@" timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b.json 2>/dev/null || { echo "$tag: failed"; return; }
    python3 - "$tag" $OUT/b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); r = d["roofline"]; t = d["target_workload"]
print(f"{sys.argv[1]:34s} {d['lnl_evals_per_s']:7.1f} evals/s prune {d['breakdown_ms']['prune_kernel']:.3f} ms products {r['executed_dense_updates_per_step']} launches {r['launches_per_step']} | target {t['lnl_evals_per_s']:.1f} evals/s prune {t['prune_kernel_ms']:.2f} ms products {t['executed_dense_updates_per_eval']}")
PY
}
run "levels 3" CRAB_CLADE_MAX_LEVELS=3
run "levels 3 prepass 1.3" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_PREPASS_COST=1.3
run "levels 3 prepass 1.6" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_PREPASS_COST=1.6
run "levels 3 item 1" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_ITEM_COST=1.0
run "levels 3 item 3" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_ITEM_COST=3.0
run "levels 3 gather 0.4" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_GATHER_COST=0.4
run "levels 3 gather 0.8" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_GATHER_COST=0.8
run "levels 2 prepass 1.3" CRAB_CLADE_MAX_LEVELS=2 CRAB_CLADE_PREPASS_COST=1.3
run "levels 4 item 2" CRAB_CLADE_MAX_LEVELS=4 CRAB_CLADE_ITEM_COST=2.0
run "levels 3 max leaves 16" CRAB_CLADE_MAX_LEVELS=3 CRAB_CLADE_MAX_LEAVES=16
