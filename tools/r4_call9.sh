#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4i; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/exact_stats" -- python3 "$R/bench.py" --workload force --encoder-precision f32x3 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-line > "$O/exact.log" 2>&1 || { tail -20 "$O/exact.log"; exit 1; }
find "$O/exact_stats" -name "*kernel_stats.csv" | head -1
tail -1 "$O/exact.log" | cut -c1-200
