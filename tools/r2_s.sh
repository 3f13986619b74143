#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2s
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 400 python bench.py --model large --steps 10 --warmup 3 --no-cpu-baseline > "$O/large.json" 2> "$O/large.err" || { tail -30 "$O/large.err"; exit 1; }
cut -c1-400 "$O/large.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --model large --steps 4 --warmup 2 --no-cpu-baseline > "$O/stats.log" 2>&1
echo done
