#!/bin/bash
# Tree parts on/off at the per-GPU shapes of configs 2-4 and at full size, same box; then a kernel trace of the config-3 share.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for off in 1 0; do
  echo "== CRAB_NO_TREE_PARTS=$off"
  for shape in "M8 100 10000" "ModelA 500 6250" "free-ratio 1000 12500" "M8 500 6250" "ModelA 500 50000" "free-ratio 1000 100000" "M8 30 400"; do
    CRAB_NO_TREE_PARTS=$off timeout -k 10 300 python3 scripts/gpu_time_c2.py $shape || exit 1
  done
done
rm -rf gpurun_out/parts_trace
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/parts_trace -o t --output-format csv -- python3 scripts/gpu_time_c2.py ModelA 500 6250 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/parts_trace/**/t_kernel_trace.csv", recursive=True)[0])))
pr = [r for r in rows if "prune" in r["Kernel_Name"]]
pr.sort(key=lambda r: int(r["Start_Timestamp"]))
for a, b in list(zip(pr[::2], pr[1::2]))[-4:]:
    d1 = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    d2 = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3
    print(f"parts launch {d1:.1f} us (grid {a['Grid_Size_X'] if 'Grid_Size_X' in a else a.get('Grid_Size')}), gap {gap:.1f} us, root launch {d2:.1f} us")
PY
