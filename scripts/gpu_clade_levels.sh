#!/bin/bash
# Clade tables: nesting depth against time at configs[1] and at the target workload (bench.py, 30 steps).
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-levels}; mkdir -p $OUT
for lv in 1 2 3 4 8; do
    CRAB_CLADE_MAX_LEVELS=$lv timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline ${2:---no-target} > $OUT/b.json 2>/dev/null || { echo "levels $lv: failed"; continue; }
    python3 - $lv $OUT/b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d["roofline"]
line = f"max levels {sys.argv[1]}: {d['lnl_evals_per_s']:7.1f} evals/s  prune {d['breakdown_ms']['prune_kernel']:.3f} ms  executed products {r['executed_dense_updates_per_step']}  clades {r['clade_tables']['clades']}  launches {r['launches_per_step']}"
t = d.get("target_workload")
if t:
    line += f" | target: {t['lnl_evals_per_s']:.1f} evals/s, prune {t['prune_kernel_ms']:.2f} ms, products {t['executed_dense_updates_per_eval']}, clades {t['clades']}"
print(line)
PY
done
