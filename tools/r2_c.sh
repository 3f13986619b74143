#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2c
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests/test_gpu_force.py tests/test_gpu_ctc_pr.py tests/test_gpu_parity2.py -m gpu -q -x -s > "$O/pytest.log" 2>&1 || { grep -E "margin\]|FAILED|Error|assert" "$O/pytest.log" | head -60; exit 1; }
grep -E "passed|failed" "$O/pytest.log" | tail -5
timeout -k 10 300 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
cut -c1-300 "$O/force.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/force_stats" -- python3 "$R/bench.py" --workload force --steps 5 --warmup 2 --no-cpu-baseline > "$O/force_stats.log" 2>&1
echo "[r2c] force stats done"
