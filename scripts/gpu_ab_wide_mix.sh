#!/bin/bash
# The log-sum-exp mixture with one (pattern, class) per lane (default) against one pattern per lane (CRAB_NO_WIDE_MIX=1).
cd "$(dirname "$0")/.."
python - <<'PY' || exit 1
import json, os, subprocess, sys
def run(**env):
    r = subprocess.run([sys.executable, "tests/test_fused_mix.py"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])
a = run(CRAB_NO_FUSED_MIX="1", CRAB_NO_WIDE_MIX="1")
b = run(CRAB_NO_FUSED_MIX="1")
worst = 0.0
for ra, rb in zip(a, b):
    ha, hb = ra["hex"], rb["hex"]
    if len(ha) > 8:                       # site-class rows: lnL, lnL (no clamp), 5 batch values, sum of sites, 16 sites
        assert ha[-16:] == hb[-16:], ("per-site bits differ", ra["n_patterns"])
    for x, y in zip(ha, hb):
        x, y = float.fromhex(x), float.fromhex(y)
        worst = max(worst, abs(x - y) / max(1.0, abs(x)))
print("per-site values: same bits; totals: largest relative difference %.2e" % worst)
PY
for i in 1 2; do for sh in "M8 30 400" "M8 100 10000" "M8 500 50000"; do CRAB_NO_WIDE_MIX=1 timeout -k 10 200 python3 scripts/gpu_time_c2.py $sh && timeout -k 10 200 python3 scripts/gpu_time_c2.py $sh || exit 1; done; done
