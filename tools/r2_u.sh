#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r02
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/pr_stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/pr_stats" -- python3 "$R/bench.py" --workload pr --steps 5 --warmup 2 --no-cpu-baseline > "$O/pr_stats.log" 2>&1
cd "$R"
python3 bench.py --workload pr > "$O/bench_pr.json" 2> "$O/bench_pr.log"
tail -1 "$O/bench_pr.json" | cut -c1-200
