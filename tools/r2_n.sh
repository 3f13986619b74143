#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2n
mkdir -p "$O"
cd "$R"
for flag in "--no-pipeline" ""; do
  timeout -k 10 300 python bench.py --workload force $flag --steps 10 --warmup 3 --no-cpu-baseline > "$O/force$flag.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
  echo "force $flag $(cut -c1-200 "$O/force$flag.json")"
done
timeout -k 10 600 python -m pytest tests/test_gpu_force.py -m gpu -q -x -k "lstm" 2>&1 | tail -2
