#!/usr/bin/env python3
"""Copy the judged summaries of one `scripts/gpu_profile_bench.sh <tag>` run from gpurun_out/prof_<tag>/ into profiles/ and
derive profiles/<round>_pmc_c2.json (what bench.py reports as roofline.traffic).

    python scripts/save_profile.py r2v0 r2_v0
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{name}_bench.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{name}_bench_under_rocprof.json"))
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{name}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, f"{name}_bench_pmc_summary.txt"))
txt = open(os.path.join(src, "pmc_summary.txt")).read()


def counter(c):
    m = re.search(rf"prune_kernel_nt1\n(?:  .*\n)*?  {c}\s+total=([0-9.e+]+)\s+rows=(\d+)", txt)
    return float(m.group(1)), int(m.group(2))


fetch, n = counter("FETCH_SIZE")
write, _ = counter("WRITE_SIZE")
hit, _ = counter("TCC_HIT_sum")
miss, _ = counter("TCC_MISS_sum")
rnd = name.split("_")[0]
out = {
    "workload": "bench.py default (C2: M8, 100 taxa x 10000 codons, 11 classes), kernel prune_kernel_nt1",
    "source": f"profiles/{name}_bench_pmc_summary.txt (rocprofv3 --pmc, separate passes, {n} launches each)",
    "FETCH_SIZE_KB_per_launch": fetch / n, "WRITE_SIZE_KB_per_launch": write / n,
    "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read",
    "hbm_bytes_per_launch": (2 * fetch + write) / n * 1024,
    "TCC_HIT_per_launch": hit / n, "TCC_MISS_per_launch": miss / n, "l2_hit_rate": hit / (hit + miss),
}
with open(os.path.join(dst, f"{rnd}_pmc_c2.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
