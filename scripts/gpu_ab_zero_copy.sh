for i in 1 2 3; do
CRAB_NO_ZERO_COPY=1 timeout -k 10 100 python bench.py --no-cpu-baseline --no-target --steps 300 > gpurun_out/ab_nzc_$i.json 2>/dev/null || exit 1
timeout -k 10 100 python bench.py --no-cpu-baseline --no-target --steps 300 > gpurun_out/ab_zc_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for k in ("nzc","zc"):
    for i in (1,2,3):
        d=json.load(open(f"gpurun_out/ab_{k}_{i}.json"))
        print(k,i,round(d["lnl_evals_per_s"],1),round(d["ms_per_step"],4),{a:round(b,4) for a,b in d["breakdown_ms"].items()})
PY
