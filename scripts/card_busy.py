#!/usr/bin/env python3
"""Share of the wall time in which at least one kernel was running, from a rocprofv3 kernel trace (csv), over the span
from the first kernel after `skip` seconds to the last one.   python scripts/card_busy.py <kernel_trace.csv> [skip_seconds]"""
import csv
import sys
from collections import defaultdict

path, skip = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
t_first = rows[0][0] + int(skip * 1e9)
rows = [r for r in rows if r[0] >= t_first]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
per = defaultdict(lambda: [0, 0])
for s, e, name in rows:
    per[name][0] += 1
    per[name][1] += e - s
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(r[1] for r in rows) - rows[0][0]
print(f"span {span / 1e9:.3f} s, at least one kernel running {busy / 1e9:.3f} s = {100 * busy / span:.1f} % card-busy; "
      f"sum of kernel durations {sum(v[1] for v in per.values()) / 1e9:.3f} s (overlap factor {sum(v[1] for v in per.values()) / busy:.2f})")
for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:50s} {n:8d} launches  {t / 1e6:10.1f} ms  avg {t / n / 1e3:8.1f} us")
