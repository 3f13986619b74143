#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2k
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$O/stats.log" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$O/stats.log"
