#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4k; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/stats.log" 2>&1 || { tail -20 "$O/stats.log"; exit 1; }
tail -1 "$O/stats.log" | cut -c1-200
cd "$R" && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || { tail -20 "$O/bench.err"; exit 1; }
cut -c1-250 "$O/bench.json"
