#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4m; mkdir -p "$O"; cd "$R"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$O/smoke.log" 2>&1 || { tail -20 "$O/smoke.log"; exit 1; }
tail -2 "$O/smoke.log"
timeout -k 10 600 python bench.py > "$O/bench.json" 2> "$O/bench.err" || { tail -20 "$O/bench.err"; exit 1; }
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4m/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["ema_rmse_vs_ref"]["abs_diff"], d["config"]["collective_plan"]["collectives_per_step"], d["config"]["collective_plan"]["ring_us_total"], d["config"]["collective_plan"]["direct_us_total"])
PY
