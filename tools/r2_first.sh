#!/bin/bash
# round-2 first GPU pass: GPU tests, force / pr bench lines, rocprof kernel statistics of the force step
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2a
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -40 "$O/pytest.log"; exit 1; }
tail -3 "$O/pytest.log"
timeout -k 10 300 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
cut -c1-400 "$O/force.json"
timeout -k 10 300 python bench.py --workload pr --steps 10 --warmup 3 --no-cpu-baseline > "$O/pr.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
cut -c1-400 "$O/pr.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/force_stats" -- python3 "$R/bench.py" --workload force --steps 5 --warmup 2 --no-cpu-baseline > "$O/force_stats.log" 2>&1
echo "[r2a] force stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/pr_stats" -- python3 "$R/bench.py" --workload pr --steps 5 --warmup 2 --no-cpu-baseline > "$O/pr_stats.log" 2>&1
echo "[r2a] pr stats done"
