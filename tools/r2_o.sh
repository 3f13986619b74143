#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2o
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --workload force --steps 6 --warmup 3 --no-cpu-baseline > "$O/trace.log" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$O/trace.log"
