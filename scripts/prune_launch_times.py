#!/usr/bin/env python3
"""Mean duration of the pruning kernel's launches by position inside an evaluation (launch 1: clade programs / tree parts /
whole walks; launch 2: main / root / path programs), from a rocprofv3 kernel trace (csv).
    python scripts/prune_launch_times.py <kernel_trace.csv> [launches per evaluation = 2] [skip first N evaluations = 6]"""
import csv
import sys

path = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if "prune_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
rows = rows[skip * per:]
n = len(rows) // per
for k in range(per):
    d = [rows[i * per + k][1] - rows[i * per + k][0] for i in range(n)]
    print(f"launch {k + 1}: mean {sum(d) / len(d) / 1e3:.1f} us over {len(d)} evaluations (min {min(d) / 1e3:.1f}, max {max(d) / 1e3:.1f})")
gaps = [rows[i * per + 1][0] - rows[i * per][1] for i in range(n)] if per > 1 else [0]
print(f"gap between launch 1 and 2: mean {sum(gaps) / len(gaps) / 1e3:.1f} us")
