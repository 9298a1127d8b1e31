"""rocprofv3 --pmc CSVs, per launch position: an evaluation with clade tables is a run of consecutive pruning launches
(clade levels, then the main programs); counters are summed per dispatch and averaged per position in the run."""
import csv, glob, sys, collections
root = sys.argv[1]
disp = {}                                    # dispatch id -> [kernel, {counter: sum}]
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            d = disp.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"].split("(")[0], collections.defaultdict(float)])
            d[1][row["Counter_Name"]] += float(row["Counter_Value"])
runs, cur = [], []
for i in sorted(disp):
    if "prune" in disp[i][0]:
        cur.append(disp[i][1])
    elif cur:
        runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
if not runs:
    sys.exit("no pruning launches in " + root)
n_launch = collections.Counter(len(r) for r in runs).most_common(1)[0][0]
runs = [r for r in runs if len(r) == n_launch][2:]   # the first evaluations warm caches
print(f"{len(runs)} evaluations of {n_launch} pruning launches")
for pos in range(n_launch):
    names = sorted(runs[0][pos])
    print(f"  launch {pos + 1}: " + "  ".join(f"{c}={sum(r[pos][c] for r in runs) / len(runs):.6g}" for c in names))
