"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch, per kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "prune" not in k and "--all" not in sys.argv:
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        # rocprofv3 emits one row per dimension instance (XCD, SE...): sum them per dispatch
        print(f"  {c:32s} total={sum(v):.6g}  rows={len(v)}")
