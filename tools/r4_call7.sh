#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4g; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python -m pytest tests/test_gpu_env_knobs.py tests/test_gpu_graphed.py tests/test_gpu_gemm.py -m gpu -q -x -s > "$O/pytest.log" 2>&1 || { tail -60 "$O/pytest.log"; exit 1; }
grep -E "bands|passed|failed" "$O/pytest.log" | tail -8
timeout -k 10 600 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4g/force.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("index_exact"), json.dumps(d["config"]["collective_plan"])[:300])
PY
